#!/bin/bash
# ThreadSanitizer over the multi-threaded host code (CPU build): scripts/tsan_host.sh
set -e
root=$(cd "$(dirname "$0")/.." && pwd); out=/tmp/tilespmv_tsan; mkdir -p $out
cd $root/tilespmv_amd/csrc
g++ -O1 -g -std=c++17 -pthread -fsanitize=thread -I../../include -DMAT_VAL_TYPE=double host_tile_create.cpp host_tilespmv_cpu.cpp host_mmio.cpp host_matrix_io.cpp $root/scripts/tsan_host.cpp -o $out/tsan_host
TILESPMV_NUM_THREADS=8 $out/tsan_host > $out/out.txt 2> $out/err.txt || true
grep -v "number of tile\|^$" $out/out.txt | tail -4
echo "tsan reports: $(grep -c 'WARNING: ThreadSanitizer' $out/err.txt || true)"
