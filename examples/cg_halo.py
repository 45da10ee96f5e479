"""Conjugate gradients on a 5-point Laplacian with x sharded across the GPUs of one node.

    python examples/cg_halo.py --grid 2048                       # one GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 \
        examples/cg_halo.py --grid 4096                          # one rank per GPU, RCCL halo exchange

Every rank tiles only its own rows (Tile_create on the renumbered row block); per iteration one SpMV, one
halo exchange of two grid lines per rank, two all-reduced dot products.  Rank 0 prints one JSON line.
`--backend gloo` lets several ranks share one GPU (rehearsal only: the halo is staged through the host).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=1024)
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--tol", type=float, default=1e-8)
    ap.add_argument("--maxiter", type=int, default=500)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--mesh", default="laplacian", choices=["laplacian", "tri-shuffled"], help="5-point Laplacian on a grid x grid mesh, or a triangulation of grid^2 nodes numbered in shuffled windows of 4096")
    ap.add_argument("--reorder", action="store_true", help="every rank renumbers its own block by reverse Cuthill-McKee (HaloSpMV(reorder=True)): vectors live in that numbering, cg permutes b at entry and x at exit")
    a = ap.parse_args()
    import torch
    import torch.distributed as dist
    from tilespmv_amd import generators as G
    from tilespmv_amd.halo import HaloSpMV, cg
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if not torch.cuda.is_available():
        raise SystemExit("cg_halo.py needs a HIP device (there is no CPU path in the product)")
    torch.cuda.set_device(local % torch.cuda.device_count())
    if world > 1:
        dist.init_process_group(a.backend, rank=rank, world_size=world)
    dtype = np.float64 if a.dtype == "f64" else np.float32
    if a.mesh == "laplacian":
        m, n, rp, ci = G.laplacian5pt(a.grid)
        rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(rp))
        vals = np.where(ci == rows, 4.0 + 1e-3, -1.0).astype(dtype)   # shifted Laplacian: SPD, modest condition number
    else:
        m, n, rp, ci = G.tri_mesh(a.grid, a.grid, shuffle=4096)
        n = (n // 16) * 16; rp = rp[:n + 1]; keep = ci[:int(rp[n])] < n   # whole tile-rows, square
        cnt = np.add.reduceat(keep.astype(np.int64), rp[:-1].astype(np.int64)) if n else np.zeros(0, dtype=np.int64)
        ci = ci[:int(rp[n])][keep]; rp = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
        rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(rp))
        deg = np.bincount(rows, weights=(ci != rows).astype(np.float64), minlength=n)
        vals = np.where(ci == rows, deg[rows] + 1.0, -1.0).astype(dtype)   # graph Laplacian + I: SPD
    del rows
    t0 = time.time()
    A = HaloSpMV(rank, world, n, rp, ci, vals, dtype, reorder=a.reorder)
    setup_s = time.time() - t0
    rng = np.random.default_rng(11)
    xs = rng.uniform(-1, 1, n).astype(dtype)                      # manufactured solution
    # b = A xs, computed with the distributed operator itself
    xs_d = A.new_vector(); xs_d[:A.nloc] = torch.from_numpy(xs[A.r0:A.r1].copy()).cuda()
    b = A.from_plan_order(A.matvec(A.to_plan_order(xs_d), A.new_vector()))    # (identity copies when the operator is not reordered)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.time()
    x, it, rel = cg(A, b, tol=a.tol, maxiter=a.maxiter)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    solve_s = time.time() - t0
    err = torch.tensor([float(torch.sum((x[:A.nloc].cpu() - torch.from_numpy(xs[A.r0:A.r1])) ** 2)), float(np.sum(xs[A.r0:A.r1].astype(np.float64) ** 2))], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(err)
    if rank == 0:
        print(json.dumps({"example": "cg_halo", "grid": a.grid, "rows": n, "nnz": int(len(ci)), "dtype": a.dtype, "ranks": world,
                          "backend": a.backend if world > 1 else None, "iterations": it, "relative_residual": rel,
                          "relative_error": float((err[0] / err[1]) ** 0.5), "setup_s": round(setup_s, 3),
                          "solve_s": round(solve_s, 4), "ms_per_iteration": round(1e3 * solve_s / max(it, 1), 4),
                          "halo_bytes_per_rank": A.halo_bytes(), "row_blocks": len(A.blocks),
                          "mesh": a.mesh, "reordered": bool(a.reorder), "bandwidth_before_after": list(A.bandwidth) if A.bandwidth else None}))
    A.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
