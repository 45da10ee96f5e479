"""Solver-shaped multi-GPU SpMV: x is sharded like y and only the halo travels (new —
the reference is single-GPU, src/main.cu:74).

``dist.ShardedSpMV`` replicates x and leaves y sharded, which is the right shape for one SpMV but not
for an iteration y -> x' -> y' ...: the full-y combine then costs 5-70x the sharded compute (SURVEY.md
§8e).  Krylov solvers only ever need, on rank k, the entries of x that rank k's rows reference.  For
banded / stencil matrices that is the rank's own slice plus a thin halo (5-pt Laplacian on a 4096² grid,
8 ranks: 2 x 4096 values against 2.1 M owned), so per iteration each rank sends a few tens of KB to its
neighbours instead of receiving 7/8 of the vector.

Layout on rank k (square n x n matrix, rows and x cut at the same nnz-balanced multiples of 16):

    x_ext = [ own slice (nloc) | halo values, grouped by owner rank, ascending column (nhalo) | 16 pad ]

The rank's rows are renumbered into that index space once, at setup, and tiled by ``Tile_create`` like
any other matrix.  One exchange = pack (index_select) -> ``all_to_all_single`` (RCCL over xGMI; most
splits are zero for a stencil) straight into the halo part of ``x_ext``.  Rows are split into up to three
contiguous blocks — leading rows that touch the halo, the interior run that does not, trailing rows that
do — so the interior block multiplies while the halo is in flight.

``reorder=True`` (round 6): the rank's own block is renumbered by reverse Cuthill-McKee on its symmetrised pattern (``tilespmv_reorder_rcm``) before it is tiled — rows and own
columns alike, halo columns stay — and every vector of the operator lives in that numbering (``x_own``, ``new_vector``, ``matvec``, ``dot``): a solver permutes its right-hand side
once at entry (``to_plan_order``) and its solution once at exit (``from_plan_order``); ``cg`` does both.  On meshes numbered in shuffled windows the kernel runs 16-25 % faster in
the better numbering (profiles/r06_reorder.txt); around ONE product the two permutations would take that back, which is why this lives here and not in the drop-in path.
"""
import numpy as np

from .dist import partition_rows, shard_csr


class HaloSpMV:
    """One rank's share of y = A x with x distributed like y.

    ``make_local(rows, cols, rowptr, colidx, vals)`` builds the local multiplier of one row block (an
    object with ``spmv(x_ptr, y_ptr, stream)``); default: ``Tile_create`` + a resident HIP ``Plan``.
    The CPU (gloo) tests inject a stand-in, as for ``ShardedSpMV``.
    """

    def __init__(self, rank, world, n, rowptr, colidx, vals, dtype=np.float64, make_local=None, device="cuda",
                 overlap=True, group=None, bounds=None, reorder=False, **plan_kw):
        import torch
        import torch.distributed as dist
        from . import api
        if n % 16:
            raise ValueError("HaloSpMV needs n to be a multiple of 16 (whole tile-rows); got %d" % n)
        self.rank, self.world, self.n, self.group = rank, world, n, group
        self.dtype = np.dtype(dtype)
        self.tdtype = torch.float64 if self.dtype == np.float64 else torch.float32
        self.device = torch.device(device)
        if bounds is None:
            self.bounds = partition_rows(rowptr, n, world)
            self.r0, self.r1 = int(self.bounds[rank]), int(self.bounds[rank + 1])
            rp, ci, v = shard_csr(rowptr, colidx, vals, self.r0, self.r1)
        else:   # rowptr / colidx / vals are this rank's row block only (rebased row pointer, global columns): see ShardedSpMV
            self.bounds = np.asarray(bounds, dtype=np.int64)
            self.r0, self.r1 = int(self.bounds[rank]), int(self.bounds[rank + 1])
            rp, ci, v = rowptr, colidx, vals
        self.nloc = self.r1 - self.r0
        ci = np.asarray(ci, dtype=np.int64)

        # ---- columns -> [own | halo] index space
        own = (ci >= self.r0) & (ci < self.r1)
        halo_cols = np.unique(ci[~own])                        # ascending => grouped by owner rank
        self.nhalo = int(halo_cols.size)
        new_ci = np.empty(ci.size, dtype=np.int32)
        new_ci[own] = (ci[own] - self.r0).astype(np.int32)
        new_ci[~own] = (self.nloc + np.searchsorted(halo_cols, ci[~own])).astype(np.int32)
        owner = np.searchsorted(self.bounds, halo_cols, side="right") - 1
        self.recv_splits = [int(np.count_nonzero(owner == k)) for k in range(world)]

        # ---- who needs what from me (setup only; object collectives are fine here)
        need = [halo_cols[owner == k] for k in range(world)]
        if world > 1:
            everyone = [None] * world
            dist.all_gather_object(everyone, need, group=group)
            give = [np.asarray(everyone[k][rank], dtype=np.int64) - self.r0 for k in range(world)]
        else:
            give = [np.zeros(0, dtype=np.int64)]
        self.send_splits = [int(g.size) for g in give]
        give_all = np.concatenate(give) if give else np.zeros(0, dtype=np.int64)
        assert give_all.size == 0 or (give_all.min() >= 0 and give_all.max() < self.nloc)
        # ---- permuted numbering of the own block (perm[new] = old): rows, own columns and what the neighbours fetch from me
        self.perm = None
        self.perm_dev = None
        self.bandwidth = None
        if reorder and self.nloc > 0:
            rp32, ci32 = np.ascontiguousarray(rp, dtype=np.int32), np.ascontiguousarray(new_ci, dtype=np.int32)
            perm = api.reorder_rcm(self.nloc, rp32, ci32, dtype=self.dtype)
            before = api.csr_bandwidth(self.nloc, rp32, ci32, dtype=self.dtype)
            rp, new_ci, v = api.csr_permute(self.nloc, rp32, ci32, v, perm, dtype=self.dtype)
            self.bandwidth = (before, api.csr_bandwidth(self.nloc, rp, new_ci, dtype=self.dtype))
            inv = np.empty(self.nloc, dtype=np.int64); inv[perm] = np.arange(self.nloc, dtype=np.int64)
            give_all = inv[give_all]
            own = new_ci < self.nloc
            self.perm = perm
            self.perm_dev = torch.from_numpy(np.ascontiguousarray(perm, dtype=np.int32)).to(self.device)
        self.give_idx = torch.from_numpy(give_all).to(self.device)
        self.send_buf = torch.zeros(max(1, give_all.size), dtype=self.tdtype, device=self.device)
        self.x_ext = torch.zeros(self.nloc + self.nhalo + 16, dtype=self.tdtype, device=self.device)
        self.cols_ext = self.nloc + self.nhalo

        # ---- row blocks: [0,a) touches halo, [a,b) interior, [b,nloc) touches halo (16-row granularity)
        nblk = self.nloc // 16
        touches = np.zeros(nblk, dtype=bool)
        if self.nhalo and nblk:
            rows_of = np.repeat(np.arange(self.nloc, dtype=np.int64), np.diff(rp))
            touches[np.unique(rows_of[~own] // 16)] = True
        plan_blocks = [(0, self.nloc)]
        if overlap and world > 1 and touches.any():
            best, run0 = (0, 0), None                            # longest run of untouched 16-row blocks
            for i in range(nblk + 1):
                t = bool(touches[i]) if i < nblk else True
                if not t and run0 is None:
                    run0 = i
                if t and run0 is not None:
                    if i - run0 > best[1] - best[0]:
                        best = (run0, i)
                    run0 = None
            a, b = best
            if rp[b * 16] - rp[a * 16] >= 0.25 * max(1, rp[-1]):  # else not worth three launches
                plan_blocks = [(0, a * 16), (a * 16, b * 16), (b * 16, self.nloc)]
        self.blocks = []   # (row_begin, row_end, local multiplier, needs_halo)
        self._tms = []
        for q0, q1 in plan_blocks:
            if q1 <= q0:
                continue
            brp, bci, bv = shard_csr(rp, new_ci, v, q0, q1)
            if make_local is not None:
                loc = make_local(q1 - q0, self.cols_ext, brp, bci, bv)
            else:
                tm = api.Tile_create(q1 - q0, self.cols_ext, int(brp[-1]), brp, bci, bv, dtype=self.dtype)
                loc = api.Plan(tm, q1 - q0, self.cols_ext, int(brp[-1]), **plan_kw)
                api.Tile_destroy(tm)   # the plan owns its device copy; the host Tile_matrix is not needed afterwards
            self.blocks.append((q0, q1, loc, bool(touches[q0 // 16:q1 // 16].any())))
        self.local_nnz = int(rp[-1])

    # ------------------------------------------------------------------------------------------
    @property
    def x_own(self):
        """The rank's slice of x inside ``x_ext`` — write here to avoid a copy in ``matvec``."""
        return self.x_ext[:self.nloc]

    def new_vector(self, fill=0.0):
        """A local vector with the 16-element tail the kernels may write."""
        import torch
        return torch.full((self.nloc + 16,), fill, dtype=self.tdtype, device=self.device)

    def _permuted(self, v, scatter):
        import torch
        from . import api
        out = self.new_vector()
        if self.perm is None:
            out[:self.nloc].copy_(v[:self.nloc])
        elif self.device.type == "cuda":
            api.permute_vector(v.data_ptr(), out.data_ptr(), self.perm_dev.data_ptr(), self.nloc, scatter=scatter, stream=self._stream(), dtype=self.dtype)
        elif scatter:
            out[:self.nloc].index_copy_(0, self.perm_dev.long(), v[:self.nloc])
        else:
            torch.index_select(v[:self.nloc], 0, self.perm_dev.long(), out=out[:self.nloc])
        return out

    def to_plan_order(self, v):
        """The rank's slice of a vector in the caller's numbering -> the operator's (a new vector; a copy when the operator is not reordered)."""
        return self._permuted(v, scatter=False)

    def from_plan_order(self, v):
        """... and back."""
        return self._permuted(v, scatter=True)

    def _stream(self):
        import torch
        return torch.cuda.current_stream().cuda_stream if self.device.type == "cuda" else 0

    def exchange_start(self):
        """Pack and post the halo exchange; returns a handle for ``exchange_finish`` (None when nothing travels)."""
        import torch
        import torch.distributed as dist
        if self.world == 1:
            return None
        nsend = int(self.give_idx.numel())
        if nsend:
            torch.index_select(self.x_ext[:self.nloc], 0, self.give_idx, out=self.send_buf[:nsend])
        recv = self.x_ext[self.nloc:self.nloc + self.nhalo]
        send = self.send_buf[:nsend]
        if self.device.type == "cuda" and dist.get_backend(self.group) == "gloo":
            # rehearsal path (several ranks sharing one GPU): stage through the host
            r = torch.empty(self.nhalo, dtype=self.tdtype)
            dist.all_to_all_single(r, send.cpu(), self.recv_splits, self.send_splits, group=self.group)
            recv.copy_(r)
            return None
        return dist.all_to_all_single(recv, send, self.recv_splits, self.send_splits, group=self.group, async_op=True)

    @staticmethod
    def exchange_finish(work):
        if work is not None:
            work.wait()

    def matvec(self, x_own, y_own):
        """y_own[:nloc] = (A x)[r0:r1] with x_own = x[r0:r1] (torch tensors on the compute device).
        ``y_own`` needs nloc + 16 elements (``new_vector``)."""
        if x_own.data_ptr() != self.x_ext.data_ptr():
            self.x_ext[:self.nloc].copy_(x_own[:self.nloc])
        work = self.exchange_start()
        es, st = y_own.element_size(), self._stream()
        for q0, q1, loc, needs in self.blocks:          # interior first: overlaps the exchange
            if not needs:
                loc.spmv(self.x_ext.data_ptr(), y_own.data_ptr() + q0 * es, st)
        self.exchange_finish(work)
        for q0, q1, loc, needs in self.blocks:
            if needs:
                loc.spmv(self.x_ext.data_ptr(), y_own.data_ptr() + q0 * es, st)
        return y_own

    def dot(self, u, v):
        """Global dot product of two distributed vectors, as a 0-dim tensor on the compute device."""
        import torch
        import torch.distributed as dist
        d = torch.dot(u[:self.nloc], v[:self.nloc]).reshape(1)
        if self.world > 1:
            if self.device.type == "cuda" and dist.get_backend(self.group) == "gloo":
                h = d.cpu(); dist.all_reduce(h, group=self.group); d = h.to(self.device)
            else:
                dist.all_reduce(d, group=self.group)
        return d[0]

    def halo_bytes(self):
        """Bytes this rank sends + receives per exchange."""
        return (sum(self.send_splits) + self.nhalo) * self.dtype.itemsize

    def close(self):
        for _, _, loc, _ in self.blocks:
            if hasattr(loc, "close"):
                loc.close()
        self.blocks = []


def cg(A, b, x0=None, tol=1e-10, maxiter=1000, check_every=8, plan_order=False):
    """Conjugate gradients on a ``HaloSpMV`` operator (symmetric positive definite A).

    All vectors are the rank's slices (``A.new_vector()`` shaped); scalars stay on the device, so the host
    synchronises only every ``check_every`` iterations for the convergence test.  Returns (x, iterations,
    relative residual).  On a reordered operator (``HaloSpMV(reorder=True)``) ``b`` / ``x0`` are permuted into the
    operator's numbering once here and the solution back once at the end, unless ``plan_order`` says they already
    are (and the result shall stay) in it."""
    import torch
    permute = getattr(A, "perm", None) is not None and not plan_order
    if permute:
        b = A.to_plan_order(b)
        x0 = None if x0 is None else A.to_plan_order(x0)
    x = A.new_vector() if x0 is None else x0.clone()
    r = A.new_vector(); Ap = A.new_vector()
    n = A.nloc
    if x0 is None:
        r[:n].copy_(b[:n])
    else:
        A.matvec(x, Ap)
        torch.sub(b[:n], Ap[:n], out=r[:n])
    p = A.x_own                      # p lives in x_ext: matvec(p) needs no copy
    p.copy_(r[:n])
    rr = A.dot(r, r)
    bb = A.dot(b, b)
    bnorm2 = float(bb) if float(bb) > 0 else 1.0
    it = 0
    rel = (float(rr) / bnorm2) ** 0.5
    while it < maxiter and rel > tol:
        A.matvec(p, Ap)
        alpha = rr / A.dot(p, Ap)
        x[:n].addcmul_(p, alpha)
        r[:n].addcmul_(Ap[:n], -alpha)
        rr_new = A.dot(r, r)
        p.mul_(rr_new / rr).add_(r[:n])
        rr = rr_new
        it += 1
        if it % check_every == 0 or it == maxiter:
            rel = (float(rr) / bnorm2) ** 0.5
    rel = (float(rr) / bnorm2) ** 0.5
    if permute:
        x = A.from_plan_order(x)
    return x, it, rel
