"""GPU: the engine's data-parallel path (world_size=2, two processes on one MI355X, gloo transport) gives the
same weights as a single-process run on the whole batch -- gradients scaled by global counts, KL sum reduced
before the free-nats clamp, flat-bucket all-reduce before clip+Adam."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("free_nats,port", [("3.0", 29621), ("0.0", 29622)])
def test_two_ranks_one_gpu_match_full_batch(free_nats, port):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "dp_gpu_worker.py")]
    env = dict(os.environ, DP_FREE_NATS=free_nats, OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=400, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    assert "DP_GPU_OK" in out.stdout
