"""Regenerates tests/golden/* from the REFERENCE ITSELF (oracle/_ref, the reference's own CPU
headers compiled in place by oracle/Makefile).  Run in the build container only:

    make -C oracle ref ref-hyb && python tests/golden/make_golden.py

Outputs are data only (inputs are regenerated from tilespmv_amd.generators, never stored as
reference text):
  kat.json            per (matrix, dtype, hyb): every scalar of Tile_matrix, FNV-1a-64 of every
                      member array, the schedule arrays, y, and a few plain numbers (format
                      histogram, sum(y), y[0..3], y[last]) that match the table in SURVEY.md §8(c)
  allfmt_f64.npz      a complete field-by-field dump for the all-format matrix (+ HYB variant)
  test.mtx            the all-format matrix as a Matrix Market file = BASELINE config 1 ("test.mtx",
                      which the reference's README names but does not ship)
  mmio_sym6.mtx/.json loader-order known answer produced by the reference reader
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))
from cases import SMALL, fnv1a64, truncated_rows, values_for  # noqa: E402
from oracle.oracle import CpuImpl  # noqa: E402
from tilespmv_amd import generators as G  # noqa: E402
from tilespmv_amd.tile_matrix import to_dict  # noqa: E402


def main():
    kat = {}
    dumps = {}
    for dtype in (np.float64, np.float32):
        for hyb in (False, True):
            R = CpuImpl("ref_hyb" if hyb else "ref", dtype)
            for name, gen in SMALL.items():
                m, n, rp, ci = gen()
                nnz = len(ci)
                rowA = truncated_rows(m)
                vals, x = values_for(name, nnz, n, dtype)
                tm = R.tile_create(rowA, n, nnz, rp, ci, vals, hyb=hyb)
                d = to_dict(tm, rowA)
                s = R.spmv(tm, rowA, n, nnz, rp, ci, vals, x)
                key = "%s/%s/%s" % (name, np.dtype(dtype).name, "hyb" if hyb else "shipped")
                rec = {"rows": rowA, "cols": n, "nnz_passed": nnz, "nnz_used": int(rp[rowA]),
                       "scalars": {k: v for k, v in d.items() if not isinstance(v, np.ndarray)},
                       "fields": {k: fnv1a64(v) for k, v in d.items() if isinstance(v, np.ndarray)},
                       "format_histogram": np.bincount(d["Format"], minlength=7).tolist(),
                       "rowblkblock": s["rowblkblock"],
                       "split_chunks": int((s["blkcoostylerowidx"] >> 31).sum()),
                       "errcount": s["errcount"],
                       "sum_y": float(s["y"].astype(np.float64).sum()),
                       "y_head": s["y"][:4].astype(float).tolist(), "y_last": float(s["y"][-1]) if rowA else 0.0,
                       "spmv": {k: fnv1a64(v) for k, v in s.items() if isinstance(v, np.ndarray)}}
                kat[key] = rec
                if name == "allfmt" and dtype == np.float64:
                    tag = "hyb_" if hyb else ""
                    for k, v in d.items():
                        dumps[tag + k] = np.asarray(v)
                    for k, v in s.items():
                        dumps[tag + "spmv_" + k] = np.asarray(v)
    with open(os.path.join(HERE, "kat.json"), "w") as f:
        json.dump(kat, f, indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(HERE, "allfmt_f64.npz"), **dumps)

    # config 1: "test.mtx"
    m, n, rp, ci = G.all_formats()
    vals = G.compat_values(len(ci))
    G.write_mtx(os.path.join(HERE, "test.mtx"), m, n, rp, ci, vals)

    # loader order KAT (symmetric file, unsorted entries)
    sym = os.path.join(HERE, "mmio_sym6.mtx")
    with open(sym, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real symmetric\n% loader-order known answer\n6 6 7\n")
        for i, j, v in [(3, 1, 5), (1, 1, 1), (6, 2, 7), (2, 2, 2), (5, 3, 9), (6, 6, 4), (4, 1, 8)]:
            f.write("%d %d %g\n" % (i, j, v))
    R = CpuImpl("ref", np.float64)
    out = {}
    for fn in ("mmio_sym6.mtx", "test.mtx"):
        r = R.mmio(os.path.join(HERE, fn))
        out[fn] = {"rc": r["rc"], "m": r["m"], "n": r["n"], "nnz": r["nnz"], "sym": r["sym"],
                   "rowptr": fnv1a64(r["rowptr"]), "colidx": fnv1a64(r["colidx"]), "val": fnv1a64(r["val"])}
        if fn == "mmio_sym6.mtx":
            out[fn]["rowptr_list"] = r["rowptr"].tolist(); out[fn]["colidx_list"] = r["colidx"].tolist()
            out[fn]["val_list"] = r["val"].tolist()
    with open(os.path.join(HERE, "mmio_kat.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", len(kat), "known-answer records")


if __name__ == "__main__":
    main()
