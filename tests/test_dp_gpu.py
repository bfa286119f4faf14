"""GPU: the engine's data-parallel path (world_size=2, two processes on one MI355X, gloo transport) gives the
same weights as a single-process run on the whole batch -- gradients scaled by global counts, KL sum reduced
before the free-nats clamp, flat-bucket all-reduce before clip+Adam."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("case,free_nats,kl_balance,port", [
    ("gauss", "3.0", None, 29621), ("gauss", "0.0", None, 29622),
    # round 3: the models whose buckets had never crossed a process boundary.  free_nats on BOTH sides of the clamp
    # (the tiny models' mean KL lies between 0 and 3), and the summed-KL branch for the Categorical KL
    ("pixel", "3.0", None, 29623), ("pixel", "0.0", None, 29624),
    ("cat", "3.0", None, 29625), ("cat", "0.0", None, 29626), ("cat", "0.01", "-1", 29627),
    ("cat_pixel", "0.0", None, 29628)])
def test_two_ranks_one_gpu_match_full_batch(case, free_nats, kl_balance, port):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "dp_gpu_worker.py")]
    env = dict(os.environ, DP_CASE=case, DP_FREE_NATS=free_nats, OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if kl_balance is not None:
        env["DP_KL_BALANCE"] = kl_balance
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=400, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    assert f"DP_GPU_OK case={case}" in out.stdout
    print(out.stdout.strip().splitlines()[-1])
