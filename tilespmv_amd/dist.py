"""Row-block sharding of y = A*x across the GPUs of one node (new; the reference is single-GPU,
src/main.cu:74).  One process per GPU, ``torch.distributed`` for the optional y combine
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

Each output row depends on exactly one tile-row of A plus x, so contiguous blocks of whole
tile-rows are independent units: rank k preprocesses and holds only its own
row block, x is replicated, and the SpMV itself needs NO collective.  A combine of y is offered
for callers that need the full vector on every rank:

* ``allgather`` — every rank receives the other slices (7/8 of |y| spread over 7 xGMI links);
* ``allreduce`` — the "RCCL reduce on y" of the north star: every rank holds a zero-filled
  full-length y with its own slice filled in; the sum is bit-identical to the gather because
  the other contributions are exact zeros.
"""
import numpy as np


def partition_rows(rowptr, rows, nparts, align=16):
    """nnz-balanced contiguous row blocks whose boundaries are multiples of ``align`` rows
    (= whole tile-rows).  Returns nparts+1 row boundaries; the last one is ``rows``."""
    rowptr = np.asarray(rowptr)
    nblk = (rows + align - 1) // align
    edges = np.minimum(np.arange(nblk + 1, dtype=np.int64) * align, rows)
    work = rowptr[edges].astype(np.int64) + 2 * edges  # nonzeros + a little per-row cost
    want = work[-1] * np.arange(1, nparts, dtype=np.float64) / nparts
    cut = np.searchsorted(work, want, side="left")
    b = np.concatenate([[0], np.minimum(cut, nblk), [nblk]]).astype(np.int64)
    b = np.maximum.accumulate(b)
    out = np.minimum(b * align, rows)
    out[-1] = rows
    return out


def shard_csr(rowptr, colidx, vals, r0, r1):
    """Rows [r0, r1) of a CSR matrix as a CSR matrix with a rebased row pointer (views, no copy of payload)."""
    lo, hi = int(rowptr[r0]), int(rowptr[r1])
    rp = (np.asarray(rowptr[r0:r1 + 1], dtype=np.int64) - lo).astype(np.int32)
    return rp, colidx[lo:hi], vals[lo:hi]


def _block_key(rp, ci, v, dtype, hyb):
    """Digest of one rank's input block + the create flags (xxh3 where available: ~10 GB/s; blake2b otherwise)."""
    try:
        import xxhash
        h = xxhash.xxh3_128()
    except ImportError:   # pragma: no cover
        import hashlib
        h = hashlib.blake2b(digest_size=16)
    for a in (rp, ci, v):
        h.update(memoryview(np.ascontiguousarray(a)).cast("B"))
    h.update(("|%s|hyb=%d|v1" % (np.dtype(dtype).name, 1 if hyb else 0)).encode())
    return h.hexdigest()


class ShardedSpMV:
    """One rank's share of a row-partitioned SpMV.

    ``local`` is the object that multiplies the local row block: by default a HIP ``Plan``;
    the CPU (gloo) tests inject a stand-in so that the partition / combine logic can be
    exercised without a GPU.
    """

    def __init__(self, rank, world, rows, cols, rowptr, colidx, vals, dtype=np.float64, make_local=None, bounds=None, tile_cache=None, hyb=False, device_build=False, **plan_kw):
        """``bounds`` given: ``rowptr / colidx / vals`` are THIS RANK'S row block only (row pointer rebased to 0, global
        column ids) and ``bounds`` the row partition everybody agreed on — a rank then never holds the whole matrix.
        ``tile_cache``: path of a Tile_matrix cache for this rank's block (read when present and of the right shape, written
        otherwise: api.matrix_load / matrix_save).  ``device_build``: the tiled matrix and the plan are built on the device from the CSR block
        (``Plan.from_csr``: nothing but the CSR arrays crosses the bus; no host Tile_matrix, no tile cache, no HYB tiles)."""
        from . import api
        self.rank, self.world, self.rows, self.cols = rank, world, rows, cols
        self.dtype = np.dtype(dtype)
        if bounds is None:
            self.bounds = partition_rows(rowptr, rows, world)
            self.r0, self.r1 = int(self.bounds[rank]), int(self.bounds[rank + 1])
            rp, ci, v = shard_csr(rowptr, colidx, vals, self.r0, self.r1)
        else:
            self.bounds = np.asarray(bounds, dtype=np.int64)
            self.r0, self.r1 = int(self.bounds[rank]), int(self.bounds[rank + 1])
            rp, ci, v = rowptr, colidx, vals
            assert len(rp) == self.r1 - self.r0 + 1 and int(rp[0]) == 0
        self.local_rows, self.local_nnz = self.r1 - self.r0, int(rp[-1])
        self.seconds = {}
        if make_local is not None:
            self.local = make_local(self.local_rows, cols, rp, ci, v)
            self.tm = None
        elif device_build:
            import time
            t0 = time.perf_counter()
            self.tm, self.tile_cache = None, None
            self.tile_cache = "unused (device build)" if tile_cache is not None else None
            self.local = api.Plan.from_csr(self.local_rows, cols, self.local_nnz, rp, ci, v, dtype=self.dtype, hyb=hyb, **plan_kw)
            t1 = time.perf_counter()
            info = self.local.info()
            self.tiles = int(info["tiles"])
            self.seconds = {"tile_create": info["tile_create_us"] * 1e-6, "plan_build": info["build_us"] * 1e-6, "plan_upload": info["upload_us"] * 1e-6,
                            "plan_create_total": t1 - t0 - info["tile_create_us"] * 1e-6, "built_on": "device"}
        else:
            import time
            t0 = time.perf_counter()
            self.tm, self.tile_cache = None, None
            key = None
            if tile_cache is not None:
                import os
                # the cache belongs to THIS input: a digest of the block (row pointer, columns, values) and of the create flags sits beside it and must match —
                # same shape with other values, another HYB setting or a changed generator is a miss, not a silent hit
                key = _block_key(rp, ci, v, self.dtype, hyb)
                self.tile_cache = "miss"
                if os.path.exists(tile_cache) and os.path.exists(tile_cache + ".key") and open(tile_cache + ".key").read().strip() == key:
                    try:
                        tm, r2, c2, z2 = api.matrix_load(tile_cache, self.dtype)
                        if (r2, c2, z2) == (self.local_rows, cols, self.local_nnz):
                            self.tm, self.tile_cache = tm, "hit"
                        else:
                            api.Tile_destroy(tm)
                    except OSError:
                        pass
            if self.tm is None:
                self.tm = api.Tile_create(self.local_rows, cols, self.local_nnz, rp, ci, v, dtype=self.dtype, hyb=hyb)
                if tile_cache is not None:
                    try:
                        api.matrix_save(self.tm, self.local_rows, cols, self.local_nnz, tile_cache)
                        with open(tile_cache + ".key", "w") as f:
                            f.write(key + "\n")
                    except OSError:
                        self.tile_cache = "miss, not writable"
            t1 = time.perf_counter()
            self.local = api.Plan(self.tm, self.local_rows, cols, self.local_nnz, **plan_kw)
            t2 = time.perf_counter()
            self.tiles = int(self.tm.tilenum)
            # the plan holds its own device copy: the host Tile_matrix (about 1 GB for config 4) is not needed any more
            api.Tile_destroy(self.tm)
            self.tm = None
            info = self.local.info()
            self.seconds = {"tile_create": t1 - t0, "plan_build": info["build_us"] * 1e-6, "plan_upload": info["upload_us"] * 1e-6,
                            "plan_create_total": t2 - t1}
        self.slice_max = int(np.diff(self.bounds).max())
        self.equal_slices = bool((np.diff(self.bounds) == self.slice_max).all())
        self._gather = None

    # y_full: torch tensor with >= rows elements on the compute device; x: torch tensor with cols elements
    def spmv(self, x, y_full, stream=0, count=1):
        """y_full[r0:r1] = A[r0:r1, :] @ x — no communication.  ``count`` > 1 repeats it back to back."""
        yp = y_full.data_ptr() + self.r0 * y_full.element_size()
        if count == 1:
            self.local.spmv(x.data_ptr(), yp, stream)
        else:
            self.local.spmv_n(x.data_ptr(), yp, stream, count)

    def combine(self, y_full, mode, force=False):
        """``force``: run the collective even in a 1-rank group (a no-op numerically) — lets a single-GPU box push the real y
        through RCCL's all_reduce / all_gather_into_tensor (tests/test_gpu_parity.py)."""
        import torch
        import torch.distributed as dist
        if mode == "none" or (self.world == 1 and not force):
            return
        if mode == "allreduce":
            if self.r0 > 0:
                y_full[:self.r0].zero_()
            if self.r1 < self.rows:
                y_full[self.r1:self.rows].zero_()
            dist.all_reduce(y_full[:self.rows], op=dist.ReduceOp.SUM)
        elif mode == "allgather":
            if self._gather is None:
                self._gather = torch.empty(self.world * self.slice_max, dtype=y_full.dtype, device=y_full.device)
                self._send = torch.zeros(self.slice_max, dtype=y_full.dtype, device=y_full.device)
            self._send[:self.local_rows].copy_(y_full[self.r0:self.r1])
            dist.all_gather_into_tensor(self._gather, self._send)
            for k in range(self.world):
                if k == self.rank:
                    continue
                a, b = int(self.bounds[k]), int(self.bounds[k + 1])
                y_full[a:b].copy_(self._gather[k * self.slice_max:k * self.slice_max + (b - a)])
        else:
            raise ValueError(mode)

    def close(self):
        if hasattr(self.local, "close"):
            self.local.close()
        if getattr(self, "tm", None) is not None:
            from . import api
            api.Tile_destroy(self.tm)
            self.tm = None
