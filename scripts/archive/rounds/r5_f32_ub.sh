#!/bin/bash
# Round 5: fp32 build, units per batch of the unit loop (a batch = one 16-byte value load per lane per 4 units in fp32: UB 4 keeps 1 KB per wavefront in flight, UB 8 two)
for rep in 1 2; do
for v in "" _ub8 _ub8w6; do
  echo "== variant '${v}' (rep $rep)"
  TILESPMV_LIB_VARIANT=$v python - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys, numpy as np, torch, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import bench
from tilespmv_amd import api, generators as G
st = torch.cuda.current_stream().cuda_stream
for wl in ("nlpkkt160", "laplacian4096", "fem3_68", "lap3d256", "band40_2000000", "fem6_46"):
    m, n, rp, ci, _ = bench.build_matrix(wl); rows = (m // 16) * 16; nnz = int(rp[rows])
    dt = np.float32
    v, x = G.compat_values(len(ci), dt), G.compat_x(n, dt)
    tm = api.Tile_create(rows, n, nnz, rp, ci, v, dtype=dt)
    p = api.Plan(tm, rows, n, nnz, placement_tries=1)
    xd = torch.from_numpy(x).cuda(); yd = torch.zeros(rows + 16, dtype=torch.float32, device="cuda")
    ms = min(p.time(xd.data_ptr(), yd.data_ptr(), st, warmup=10, reps=50) for _ in range(2))
    i = p.info()
    print("%-16s f32 %.4f ms  frac %.3f  entry_mode %d csr_form %d" % (wl, ms, api.algorithmic_bytes(nnz, rows, n, 4) / ms * 1e-6 / 8000, i["entry_mode"], i["csr_form"]), flush=True)
    p.close(); api.Tile_destroy(tm)
PY
done
done
