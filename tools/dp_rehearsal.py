"""One-rank RCCL rehearsal of the data-parallel schedule from a LIBRARY user's import order (torch first, then the
package): does the package's GPU_MAX_HW_QUEUES default still reach the HIP runtime?  Prints ms/step."""
import os, sys, time
import torch                                    # (HIP is not initialised by the import)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "cuda-first":
    torch.cuda.init()                           # the runtime reads its flags now: the package's default comes too late
import big_dreamer_amd
print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"))
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ["BD_FORCE_DP"] = "1"
if os.environ.get("REH_DEVICE_ID", "0") == "1":      # eager communicator, as bench.py creates it
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
else:
    dist.init_process_group("nccl", rank=0, world_size=1)
from big_dreamer_amd import synth
from big_dreamer_amd.engine import DreamerEngine
from big_dreamer_amd.memory import ExperienceReplay
d, dev = synth.CONFIG2, torch.device("cuda", 0)
torch.cuda.set_device(dev)
if os.environ.get("REH_OWN_STREAM", "0") == "1":     # the caller on a stream of its own, as bench.py
    torch.cuda.set_stream(torch.cuda.Stream(dev))
eng = DreamerEngine(d, None, dev, params=synth.make_params(d, 0), world_size=1)      # (BD_PHASE_GROUPS=0: one communicator)
rep = synth.make_replay(d, rows=5000, seed=0)
buf = ExperienceReplay(5000, d.A, 5, False, d.O, dev)
for k, v in rep.items():
    getattr(buf, k)[:] = v
buf.idx, buf.full = 0, True
buf.sync_device()
def step():
    o, a, r, n = buf.sample(d.B, d.L)
    eng.train_step({"observations": o, "actions": a, "rewards": r, "nonterminals": n}, None, sync_logs=False)
for _ in range(8):
    step()
eng.flush_optimizers(); eng.join(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(40):
    step()
eng.flush_optimizers(); eng.join(); torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / 40 * 1e3:.3f} ms/step (one-rank RCCL rehearsal, dp.force={eng.dp.force})")
dist.destroy_process_group()
