#!/bin/bash
# round 4: the resident-grid form of the workgroup entry mode (16 x 256 entries per trip) on the cache-resident BASELINE configs; whole GPU suite first
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r4resident; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
timeout -k 10 300 python tests/gpu_fuzz.py 30 8000 > $out/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -1 $out/fuzz.log
timeout -k 10 300 python scripts/exp_bench.py webbase TILESPMV_ENTRY_TRIP=6 TILESPMV_ENTRY_TRIP=16 TILESPMV_ENTRY_TRIP=6,TILESPMV_COO_ORDERED=1 TILESPMV_ENTRY_TRIP=16,TILESPMV_COO_ORDERED=1 \
   TILESPMV_ENTRY_TRIP=16,TILESPMV_STRIP_COST=1200 TILESPMV_ENTRY_TRIP=16,TILESPMV_STRIP_COST=2000 TILESPMV_ENTRY_TRIP=6,Q=2 TILESPMV_ENTRY_TRIP=16,Q=2 > $out/exp_webbase.txt 2>&1; grep -v amdgpu.ids $out/exp_webbase.txt | cut -c1-200
TILESPMV_CREATE_HYB=1 timeout -k 10 300 python scripts/exp_bench.py scircuit Q=1 TILESPMV_WAVE_COO=2,TILESPMV_ENTRY_TRIP=16 TILESPMV_WAVE_COO=2,TILESPMV_ENTRY_TRIP=6 TILESPMV_WAVE_COO=2,TILESPMV_ENTRY_TRIP=16,TILESPMV_STRIP_COST=800 Q=2 > $out/exp_scircuit.txt 2>&1; grep -v amdgpu.ids $out/exp_scircuit.txt | cut -c1-200
for wl in powerlaw2000000 circuit1000000; do timeout -k 10 300 python scripts/exp_bench.py $wl TILESPMV_ENTRY_TRIP=6 TILESPMV_ENTRY_TRIP=16 > $out/exp_$wl.txt 2>&1; grep -v amdgpu.ids $out/exp_$wl.txt | cut -c1-200; done
timeout -k 10 400 python scripts/stamps_probe.py webbase TILESPMV_ENTRY_TRIP=6 > $out/stamps_webbase_trip6.txt 2>&1; grep -v amdgpu.ids $out/stamps_webbase_trip6.txt
timeout -k 10 400 python scripts/stamps_probe.py webbase TILESPMV_ENTRY_TRIP=16 > $out/stamps_webbase_trip16.txt 2>&1; grep -v amdgpu.ids $out/stamps_webbase_trip16.txt
