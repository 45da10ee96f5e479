"""`python bench.py --gpus N` started bare starts its own ranks as a child torch.distributed.run and relays their
exit status (runs without a GPU: the ranks then stop with the "needs an MI355X" message, which is what is checked —
the launch path itself is what the driver's round-end command depends on)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bare_multi_gpu_command_starts_child_ranks():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("covered end to end by tests/test_bench_contract.py on a GPU box")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--backend", "gloo", "--workload", "laplacian64", "--steps", "2", "--warmup", "1"],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode != 0                      # the ranks' failure is relayed, not swallowed
    assert "needs an MI355X" in r.stderr          # ... and it is the ranks (WORLD_SIZE=2) that got as far as the device check
    assert "--gpus 2 but WORLD_SIZE" not in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def _record(world):
    """A full record shaped like bench.py's, with `world` ranks and every side table populated (sizes as in round 5's 30 KB line)."""
    import json
    with open(os.path.join(ROOT, "profiles", "r05_bench_line.json")) as f:
        d = json.load(f)
    d["steady_state"] = {"setup_launches": 200, "value": 1023.19, "ms_per_step": 0.16394, "kernel_ms": 0.16312}
    d["n_gpus"] = d["ranks"] = world
    d["rank_devices"] = [{"rank": r, "device": r, "name": "AMD Instinct MI355X", "uuid": "GPU-%032x" % r, "pci": "0000:%02x:00" % (5 + 16 * r), "cus": 256} for r in range(world)]
    d["rccl_version"] = "2.26.6"
    d["per_rank_ms_per_step"] = {"wall": [0.02345] * world, "device": [0.02123] * world, "min": 0.02345, "max": 0.02345}
    d["prep_seconds_per_rank"] = [dict(d["prep_seconds"]) for _ in range(world)]
    if world > 1:
        d["cpu_baseline"] = None
        d["with_y_combine"] = {"allgather": {"ms_per_step": 0.21234, "gflops": 790.1, "check_full_y_on_every_rank": "pass"},
                               "allreduce": {"ms_per_step": 1.61234, "gflops": 104.0, "check_full_y_on_every_rank": "pass"},
                               "halo": {"ms_per_step": 0.03123, "gflops": 5370.9, "halo_bytes_per_rank": 65536, "note": "x" * 100, "check_own_rows_on_every_rank": "pass"}}
    return d


def test_final_line_stays_short_for_one_and_eight_ranks():
    """Round 5's line was 30 KB and the driver did not parse it.  The line bench.py prints now is the compact record: < 4 KB for the headline run with every side table,
    and for an 8-rank record carrying devices, per-rank kernel times and the three combines; it keeps the contract keys and names the sidecar."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    for world in (1, 8):
        txt = bench.compact_line(_record(world), "bench_full.json")
        assert len(txt) < 4096, (world, len(txt))
        d = json.loads(txt)
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
            assert k in d, k
        assert d["full"] == "bench_full.json" and d["roofline"]["frac"] > 0 and d["config"]["workload"] == "laplacian4096"
        if world == 8:
            assert len(d["rank_devices"]) == 8 and len(d["per_rank_kernel_ms"]) == 8 and d["with_y_combine"]["allreduce"] == {"ms": 1.61234, "check": "pass"} and d["rccl_version"]
        else:
            assert d["cpu_baseline"]["kind"] == "reference" and len(d["other_workloads"]) == 13
    # a record that would not fit sheds its side tables, never the contract keys
    big = _record(1)
    big["other_workloads"] = {"w%03d" % i: dict(big["other_workloads"]["lap3d256"]) for i in range(400)}
    txt = bench.compact_line(big, "bench_full.json")
    assert len(txt) < 4096 and json.loads(txt)["other_workloads"].startswith("see ")


def test_phase_limit_ends_the_process_with_a_message(tmp_path):
    """A phase that hangs (first contact with an 8-GPU node: a rendezvous, a collective) ends the rank with exit code 124 and says which phase it was."""
    code = ("import sys, time; sys.path.insert(0, %r); import bench\n"
            "with bench.phase('fast', 5): pass\n"
            "with bench.phase('stuck collective', 0.3): time.sleep(30)\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=60)
    assert r.returncode == 124 and "phase 'stuck collective' exceeded" in r.stderr


def test_launcher_kills_ranks_that_do_not_finish():
    """`bench.py --gpus N` bare: the ranks are a child process group with a job limit; when it expires exactly that group is killed and the exit code is non-zero."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["TILESPMV_BENCH_JOB_TIMEOUT"] = "0.5"
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--backend", "gloo", "--workload", "laplacian64", "--steps", "2", "--warmup", "1"],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert r.returncode == 124 and "did not finish within" in r.stderr
