#!/bin/bash
# Round 5: what the LDS adds of the pooled units cost (timing-only diagnostic builds; TILESPMV_LIB_VARIANT picks the library)
set -e
for v in "" _pabl1 _pabl2 _pabl3; do
  echo "== variant '${v}'"
  TILESPMV_LIB_VARIANT=$v python - <<'PY'
import numpy as np, torch, os
from tilespmv_amd import api, generators as G
st = torch.cuda.current_stream().cuda_stream
for wl, gen in (("fem3_68", lambda: G.fem_hex(68, 68, 68, 3)),):
    m, n, rp, ci = gen(); rows = (m // 16) * 16; nnz = int(rp[rows])
    for dt in (np.float64, np.float32):
        v, x = G.compat_values(len(ci), dt), G.compat_x(n, dt)
        tm = api.Tile_create(rows, n, nnz, rp, ci, v, dtype=dt)
        for kw in (dict(csr_split=2, entry_mode=0),):
            p = api.Plan(tm, rows, n, nnz, **kw)
            xd = torch.from_numpy(x).cuda(); yd = torch.zeros(rows + 16, dtype=torch.float64 if dt == np.float64 else torch.float32, device="cuda")
            ms = p.time(xd.data_ptr(), yd.data_ptr(), st, warmup=10, reps=50)
            print(wl, dt.__name__, kw, "%.4f ms" % ms, flush=True)
            p.close()
PY
done
