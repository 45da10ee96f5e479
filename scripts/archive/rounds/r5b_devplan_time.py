"""Round 5 (second session): preparation end to end — host Tile_create + tilespmv_plan_create against tilespmv_plan_create_from_csr (everything on the device) — per workload,
with the stream digests and the first SpMV compared.  TILESPMV_PLAN_VERBOSE / TILESPMV_CREATE_VERBOSE print the stages."""
import os, sys, time
if os.environ.get("R5B_VERBOSE"): os.environ["TILESPMV_PLAN_VERBOSE"] = "1"; os.environ["TILESPMV_CREATE_VERBOSE"] = "1"
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from tilespmv_amd import api, generators as G
for wl in sys.argv[1].split(","):
    m, n, rp, ci, _ = bench.build_matrix(wl); rows = (m // 16) * 16; nnz = int(rp[rows])
    dt = np.float32 if wl == "nlpkkt160" else np.float64
    v = G.real_values(len(ci), dt); x = G.real_x(n, nnz, dt)
    xd = torch.from_numpy(x).cuda(); yh = torch.zeros(rows + 16, dtype=xd.dtype, device="cuda"); yd = torch.zeros_like(yh)
    for rep in range(2):
        t0 = time.time(); tm = api.Tile_create(rows, n, nnz, rp, ci, v, dtype=dt); t_tc = time.time() - t0
        t0 = time.time(); ph = api.Plan(tm, rows, n, nnz, placement_tries=1); t_pc = time.time() - t0
        t0 = time.time(); pd = api.Plan.from_csr(rows, n, nnz, rp, ci, v, dtype=dt, placement_tries=1); t_dev = time.time() - t0
        # ... and from a CSR that already lives on the device (nothing to upload)
        d_rp = torch.from_numpy(np.ascontiguousarray(rp[:rows + 1], dtype=np.int32)).cuda(); d_ci = torch.from_numpy(np.ascontiguousarray(ci[:nnz], dtype=np.int32)).cuda(); d_v = torch.from_numpy(np.ascontiguousarray(v[:nnz])).cuda()
        torch.cuda.synchronize()
        t0 = time.time(); pdd = api.Plan.from_device_csr(rows, n, nnz, d_rp.data_ptr(), d_ci.data_ptr(), d_v.data_ptr(), dt, placement_tries=1); t_dd = time.time() - t0
        pdd.close(); del d_rp, d_ci, d_v
        ph.spmv(xd.data_ptr(), yh.data_ptr()); pd.spmv(xd.data_ptr(), yd.data_ptr()); torch.cuda.synchronize()
        hs, ds = ph.stream_digests(), pd.stream_digests()
        same = sorted(hs) == sorted(ds) and all(hs[k] == ds[k] for k in hs)
        ms_h = ph.time(xd.data_ptr(), yh.data_ptr(), 0, 5, 20); ms_d = pd.time(xd.data_ptr(), yd.data_ptr(), 0, 5, 20)
        i = pd.info()
        hi = ph.info()
        print("%s rep %d: host Tile_create %.3f s + plan create %.3f s = %.3f s (timed choices %.0f ms) | from_csr %.3f s (device Tile_create %.3f s incl. CSR upload; timed choices %.0f ms), from a device-resident CSR %.3f s | streams identical: %s, y identical: %s (plan-fixed summation order: %s, same timed launch form: %s) | SpMV %.4f / %.4f ms"
              % (wl, rep, t_tc, t_pc, t_tc + t_pc, hi["timed_choices_us"] * 1e-3, t_dev, i["tile_create_us"] * 1e-6, i["timed_choices_us"] * 1e-3, t_dd, same, bool(torch.equal(yh, yd)), bool(hi["entry_ordered"] and i["entry_ordered"]),
                 all(hi[k] == i[k] for k in ("x_panels", "x_panel_merge", "x_slice_passes")), ms_h, ms_d), flush=True)
        ph.close(); pd.close(); api.Tile_destroy(tm)
