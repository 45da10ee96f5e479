#!/bin/bash
# ThreadSanitizer over the multi-threaded host code (CPU build): scripts/tsan_host.sh
set -e
root=$(cd "$(dirname "$0")/.." && pwd); out=/tmp/tilespmv_tsan; mkdir -p $out
cd $root/tilespmv_amd/csrc
# hip_plan.hip is compiled host-only (plain g++, HIP headers for the types; libamdhip64 only satisfies the linker: the layout-digest build makes no HIP call)
g++ -O1 -g -std=c++17 -pthread -fsanitize=thread -I../../include -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -DMAT_VAL_TYPE=double -Wno-unused-result \
    host_tile_create.cpp host_tilespmv_cpu.cpp host_mmio.cpp host_matrix_io.cpp -x c++ hip_plan.hip hip_plan_stream.hip $root/scripts/host_stubs.cpp $root/scripts/tsan_host.cpp -o $out/tsan_host -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib
TILESPMV_NUM_THREADS=8 $out/tsan_host > $out/out.txt 2> $out/err.txt || true
grep -v "number of tile\|^$" $out/out.txt | tail -6
echo "tsan reports: $(grep -c 'WARNING: ThreadSanitizer' $out/err.txt || true)"
