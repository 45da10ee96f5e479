"""N>1 path on CPU: two processes over gloo.  The local multiply is a stand-in (the oracle's
tile SpMV on the rank's row block), so what is tested is the partition, the row-block
extraction, the y placement and both combine modes — bit-exact against the 1-process y."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


class _OracleLocal:
    def __init__(self, rows, cols, rp, ci, v, dtype):
        from oracle.oracle import CpuImpl
        self.O = CpuImpl("oracle", dtype)
        self.args = (rows, cols, len(ci), rp, ci, v)
        self.tm = self.O.tile_create(rows, cols, len(ci), rp, ci, v)
        self.rows = rows

    def spmv(self, x_ptr, y_ptr, stream=0):  # noqa
        import ctypes as C
        rows, cols, nnz, rp, ci, v = self.args
        dt = self.O.dtype
        x = np.frombuffer((C.c_char * (cols * dt.itemsize)).from_address(x_ptr), dtype=dt)
        y = np.frombuffer((C.c_char * (rows * dt.itemsize)).from_address(y_ptr), dtype=dt)
        y[:] = self.O.spmv(self.tm, rows, cols, nnz, rp, ci, v, x)["y"]


def _worker(rank, world, port, name, mode, dtype_name, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cases import SMALL, MEDIUM, truncated_rows, values_for
    from tilespmv_amd.dist import ShardedSpMV
    dtype = np.dtype(dtype_name)
    m, n, rp, ci = (SMALL.get(name) or MEDIUM[name])()
    rows = truncated_rows(m)
    vals, x = values_for(name, len(ci), n, dtype)
    sh = ShardedSpMV(rank, world, rows, n, rp, ci, vals, dtype, make_local=lambda r, c, a, b, v: _OracleLocal(r, c, a, b, v, dtype))
    xt = torch.from_numpy(x.copy())
    yt = torch.full((rows + 16,), 7.0, dtype=xt.dtype)
    sh.spmv(xt, yt)
    sh.combine(yt, mode)
    if rank == 0:
        np.save(out, yt.numpy()[:rows])
    b = torch.tensor(sh.bounds.copy())
    dist.broadcast(b, 0)
    assert np.array_equal(b.numpy(), sh.bounds)  # every rank derived the same partition
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["allgather", "allreduce"])
@pytest.mark.parametrize("name,dtype", [("lap64", "float64"), ("powerlaw20k", "float64"), ("circuit8k", "float32"), ("one_long_row", "float64")])
def test_two_rank_row_partition_matches_single(tmp_path, name, dtype, mode):
    from cases import SMALL, truncated_rows, values_for
    from oracle.oracle import CpuImpl
    out = str(tmp_path / "y.npy")
    port = 29500 + (os.getpid() * 7 + hash((name, mode)) % 1000) % 2000
    mp.spawn(_worker, args=(2, port, name, mode, dtype, out), nprocs=2, join=True)
    m, n, rp, ci = SMALL[name]()
    rows = truncated_rows(m)
    vals, x = values_for(name, len(ci), n, np.dtype(dtype))
    O = CpuImpl("oracle", np.dtype(dtype))
    y1 = O.spmv(O.tile_create(rows, n, len(ci), rp, ci, vals), rows, n, len(ci), rp, ci, vals, x)["y"]
    assert np.array_equal(np.load(out), y1)


def test_partition_rows_properties():
    from tilespmv_amd.dist import partition_rows, shard_csr
    from cases import SMALL
    m, n, rp, ci = SMALL["powerlaw20k"]()
    rows = (m // 16) * 16
    for parts in (1, 2, 3, 4, 8):
        b = partition_rows(rp, rows, parts)
        assert b[0] == 0 and b[-1] == rows and (np.diff(b) >= 0).all() and (b[:-1] % 16 == 0).all()
        nnz = np.diff(rp[b])
        assert nnz.sum() == rp[rows]
        if parts > 1:
            assert nnz.max() <= 1.35 * nnz.sum() / parts + 5000
        r, c, v = shard_csr(rp, ci, ci.astype(np.float64), int(b[0]), int(b[1]))
        assert r[0] == 0 and r[-1] == len(c)
