#!/bin/bash
# End-to-end run of the reference's command line on a large generated .mtx (5-pt Laplacian 2048^2: 4.2 M rows, 21 M nnz,
# ~400 MB of text): parallel reader -> Tile_create -> tilespmv_cpu check -> HIP SpMV -> PASS line.  scripts/cli_demo.sh [N]
N=${1:-2048}
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out
python - <<PY
import sys, time
sys.path.insert(0, "$GRAFT_REPO_ROOT")
from tilespmv_amd import generators as G
m, n, rp, ci = G.laplacian5pt($N)
t = time.time(); G.write_mtx("/tmp/lap$N.mtx", m, n, rp, ci, G.compat_values(len(ci))); print("wrote /tmp/lap$N.mtx in %.1f s" % (time.time() - t))
PY
ls -la /tmp/lap$N.mtx
cd /tmp && TILESPMV_WARMUP=50 TILESPMV_BENCH_REPEAT=200 $GRAFT_REPO_ROOT/tilespmv_amd/bin/test_f64 -d 0 /tmp/lap$N.mtx 2>&1 | tee $out/cli_demo_lap$N.txt
