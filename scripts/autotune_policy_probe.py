import os, sys, json, numpy as np
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/bench.py") else os.environ["GRAFT_REPO_ROOT"])
os.environ["TILESPMV_AUTOTUNE_LOG"] = "/tmp/at.jsonl"
import torch, bench
from tilespmv_amd import api, generators as G
for wl in sys.argv[1:]:
    m, n, rp, ci, _ = bench.build_matrix(wl)
    rows = m // 16 * 16; nnz = int(rp[rows])
    tm = api.Tile_create(rows, n, nnz, rp, ci, G.compat_values(len(ci)))
    p = api.Plan(tm, rows, n, nnz, autotune=True)
    i = p.info(); print(wl, "nt_stream", i["nt_stream"], "stream MB", i["stream_bytes"] >> 20, flush=True)
    p.close(); api.Tile_destroy(tm)
for l in open("/tmp/at.jsonl"):
    d = json.loads(l); print(d.get("stream_policy"), d["choice"])
