#!/bin/bash
# AddressSanitizer + UBSan over the host-side C++ (Tile_create, tilespmv_cpu, .mtx reader, matrix cache) on the CPU build
# (GPU ASan is not available on the pool).  Runs here, no GPU needed:  scripts/asan_host.sh
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=/tmp/tilespmv_asan; mkdir -p $out
cd $root/tilespmv_amd/csrc
for f in host_tile_create host_tilespmv_cpu host_mmio host_matrix_io host_reorder; do
  g++ -O1 -g -fPIC -std=c++17 -pthread -fsanitize=address,undefined -fno-omit-frame-pointer -I../../include -DMAT_VAL_TYPE=double -c $f.cpp -o $out/$f.o
done
# the plan layout builder, host-only (plain g++; libamdhip64 only satisfies the linker: tilespmv_plan_layout_digest makes no HIP call)
for f in hip_plan hip_plan_stream; do
  g++ -O1 -g -fPIC -std=c++17 -pthread -fsanitize=address,undefined -fno-omit-frame-pointer -I../../include -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -DMAT_VAL_TYPE=double -Wno-unused-result -x c++ -c $f.hip -o $out/$f.o
done
g++ -O1 -g -fPIC -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer -I../../include -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -DMAT_VAL_TYPE=double -c $root/scripts/host_stubs.cpp -o $out/host_stubs.o
g++ -shared -fsanitize=address,undefined -pthread $out/*.o -o $out/libhost_asan.so -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib
cd $out
LD_PRELOAD=$(g++ -print-file-name=libasan.so):$(g++ -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1 \
  TILESPMV_NUM_THREADS=4 python $root/scripts/asan_host_drive.py > $out/out.txt 2> $out/err.txt
grep -v "errcount\|number of tile\|^$" $out/out.txt | tail -5
echo "sanitizer reports: $(grep -c -E 'ERROR: AddressSanitizer|runtime error' $out/err.txt || true)"
