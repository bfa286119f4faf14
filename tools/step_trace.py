"""Diagnostic: GPU-side duration of each of the first N train steps of a fresh engine (events on the behaviour stream),
and the host time to enqueue them -- what the start-up transient of a short benchmark is made of."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from big_dreamer_amd import synth
from big_dreamer_amd.engine import DreamerEngine
from big_dreamer_amd.memory import ExperienceReplay
d, dev = synth.CONFIG2, torch.device("cuda", 0)
eng = DreamerEngine(d, None, dev, params=synth.make_params(d, 0))
rep = synth.make_replay(d, rows=5000, seed=0)
buf = ExperienceReplay(5000, d.A, 5, False, d.O, dev)
for k, v in rep.items():
    getattr(buf, k)[:] = v
buf.idx, buf.full = 0, True
buf.sync_device()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 80
PRE = int(sys.argv[2]) if len(sys.argv) > 2 else 0
DEPTH = int(sys.argv[3]) if len(sys.argv) > 3 else 0
if PRE:      # a burst of un-synchronised tiny launches on the engine's streams: does the runtime grow its launch pools once?
    t0 = time.perf_counter()
    z = torch.zeros(64, device=dev)
    for st in (eng._s_wm, eng._s_bh, eng._side):
        with torch.cuda.stream(st):
            for _ in range(PRE):
                z.add_(1.0)
    torch.cuda.synchronize()
    print(f"prefill: {3 * PRE} launches in {(time.perf_counter() - t0) * 1e3:.1f} ms")
evs, host = [], []
torch.cuda.synchronize()
for i in range(N):
    t0 = time.perf_counter()
    o, a, r, n = buf.sample(d.B, d.L)
    eng.train_step({"observations": o, "actions": a, "rewards": r, "nonterminals": n}, None, sync_logs=False)
    host.append((time.perf_counter() - t0) * 1e3)
    e = torch.cuda.Event(enable_timing=True)
    e.record(eng._s_bh)
    evs.append(e)
    if DEPTH and i >= DEPTH:
        evs[i - DEPTH].synchronize()        # host never more than DEPTH steps ahead of the behaviour stream
eng.flush_optimizers(); eng.join(); torch.cuda.synchronize()
gpu = [evs[i].elapsed_time(evs[i + 1]) for i in range(N - 1)]
for i in range(0, N - 1, 8):
    print(f"steps {i + 1:3d}..{min(i + 8, N - 1):3d}: gpu ms between behaviour-phase ends {np.round(gpu[i:i + 8], 2).tolist()}  host enqueue ms {np.round(host[i + 1:i + 9], 2).tolist()}")
