#!/bin/bash
# f2 at the size that motivated it: the nlpkkt160 stand-in (8,345,600 rows, 229,518,112 nonzeros) written as
# Matrix Market text (~4 GB), then through the CLI twice with --cache: first run parses the text with host_mmio.cpp and writes
# the CSR + Tile_matrix caches, second run reads them.  Keeps: sizes, seconds, peak RSS (wait4), the CLI's own lines.
#   scripts/mtx_scale.sh gpurun_out/r3_mtx_scale.txt      (needs ~12 GB of /tmp and of RAM; removes its files afterwards)
out=${1:-gpurun_out/r3_mtx_scale.txt}; work=${TMPDIR:-/tmp}/tilespmv_mtx_scale; mkdir -p $work $(dirname $out)
root=$(cd "$(dirname "$0")/.." && pwd)
{
echo "== write: nlpkkt160 stand-in as text"
python - "$work/nlpkkt160_like.mtx" <<'PY'
import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
from tilespmv_amd import api, generators as G
t = time.time(); m, n, rp, ci = G.nlpkkt_like(160); print("generate %.1f s: %d x %d, %d nonzeros" % (time.time() - t, m, n, len(ci)), flush=True)
t = time.time(); api.mtx_write(sys.argv[1], m, n, rp, ci, (np.arange(len(ci)) % 7 + 1).astype(np.float32), dtype=np.float32)
print("tilespmv_mtx_write %.1f s, %.2f GB of text" % (time.time() - t, os.path.getsize(sys.argv[1]) / 1e9), flush=True)
PY
for pass in first second; do
  echo "== CLI $pass run: test_f32 -d 0 nlpkkt160_like.mtx --cache"
  ( cd $work && TILESPMV_WARMUP=50 TILESPMV_BENCH_REPEAT=200 python - $root/tilespmv_amd/bin/test_f32 -d 0 $work/nlpkkt160_like.mtx --cache <<'PY'
import os, sys, time
t0 = time.time()
pid = os.fork()
if pid == 0:
    os.execv(sys.argv[1], sys.argv[1:])        # (this child has not touched the GPU: a plain exec)
_, status, ru = os.wait4(pid, 0)
print("exit status %d   wall %.1f s   peak RSS %.2f GB   user %.1f s   sys %.1f s" % (os.waitstatus_to_exitcode(status), time.time() - t0, ru.ru_maxrss / 1048576.0, ru.ru_utime, ru.ru_stime))
PY
  ) | grep -v "^$"
  ls -la $work | grep -E "csr_f32|tile_f32" | awk '{print "   cache file", $NF, $5, "bytes"}'
done
} > $out 2>&1
rm -rf $work
tail -40 $out
