"""CPU, world_size=2 over gloo: the data-parallel rules (big_dreamer_amd/parallel.py) reproduce single-process
gradients on the whole batch, in both free-nats regimes (clamp saturated / not saturated)."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(world, env_extra, port):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "dp_cpu_worker.py")]
    env = dict(os.environ, OMP_NUM_THREADS="2", **env_extra)
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    assert "DP_OK" in out.stdout, out.stdout[-1500:]


@pytest.mark.parametrize("free_nats,port", [("3.0", 29611), ("0.0", 29612)])
def test_two_rank_gradients_equal_full_batch(free_nats, port):
    _run(2, {"DP_FREE_NATS": free_nats}, port)


def test_shard_helpers():
    from big_dreamer_amd.parallel import DataParallel
    x = torch.arange(2 * 6 * 3, dtype=torch.float32).reshape(2, 6, 3)
    parts = [DataParallel(3, r).shard_batch({"x": x})["x"] for r in range(3)]
    assert torch.equal(torch.cat(parts, dim=1), x)
    T, B = 4, 6
    rows = torch.arange(T * B * 2, dtype=torch.float32).reshape(T * B, 2)
    sh = DataParallel(2, 1).shard_rows(rows, T, B)
    assert torch.equal(sh, rows.reshape(T, B, 2)[:, 3:].reshape(-1, 2))
    assert DataParallel(4, 0).mean_grad_scale(10) == 1.0 / 40
