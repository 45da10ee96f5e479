"""Round 5 (the free host-side gate): would keeping the shard's most popular columns of x in LDS take work off the texture-address path?
The merged entry lists of the workgroup entry mode are in COLUMN order, so the entries of one popular column already sit in neighbouring lanes of one gather
instruction and share one 128-byte line.  What the address path pays for is distinct lines per 64-lane gather: this script counts them, per list chunk of 64
records, for all entries and for the entries that are NOT among the top-H columns — the difference is the most an LDS-resident hot slice of x could remove.
Groups are approximated as consecutive row blocks holding about the entries one workgroup's list has (5.7 k on the power-law matrix, 4.1 k on the webbase stand-in)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench

def stat(wl, per_group, H=4096):
    m, n, rp, ci, _ = bench.build_matrix(wl)
    nnz = len(ci)
    cnt = np.bincount(ci, minlength=n)
    hot_cols = np.argsort(cnt)[::-1][:H]
    is_hot_col = np.zeros(n, bool); is_hot_col[hot_cols] = True
    share = cnt[hot_cols].sum() / nnz
    # groups: consecutive nonzeros in CSR order, per_group each (rows are contiguous); inside a group sort by column
    ng = nnz // per_group
    ci2 = ci[:ng * per_group].astype(np.int64).reshape(ng, per_group)
    ci2.sort(axis=1)
    nch = per_group // 64
    ch = ci2[:, :nch * 64].reshape(ng, nch, 64)
    lines = ch >> 4                                   # 128-byte lines of fp64 x
    distinct_all = (np.diff(lines, axis=2) != 0).sum(axis=2) + 1
    hot = is_hot_col[ch]
    # the same chunks with the hot entries taken out: distinct lines among the cold lanes only
    cold_lines = np.where(hot, -1, lines)
    srt = np.sort(cold_lines, axis=2)
    distinct_cold = ((np.diff(srt, axis=2) != 0) & (srt[:, :, 1:] >= 0)).sum(axis=2) + (srt[:, :, 0] >= 0)
    tot_all, tot_cold = distinct_all.sum(), distinct_cold.sum()
    print("%-18s nnz %5.1f M  top-%d columns hold %.3f of the entries; distinct x lines per 64-lane gather: %.2f with all entries, %.2f of them belong to cold entries -> an LDS hot slice removes at most %.1f %% of the line requests (and adds %d staging gathers per workgroup)" % (
        wl, nnz / 1e6, H, share, tot_all / (ng * nch), tot_cold / (ng * nch), 100.0 * (1 - tot_cold / tot_all), H))

for wl, pg in (("webbase", 4096), ("powerlaw8000000", 5760), ("plaw18_6000000", 5760), ("rmat22x8", 5760)):
    for H in (512, 4096):
        stat(wl, pg, H)
