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
