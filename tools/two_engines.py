#!/usr/bin/env python3
"""Two engines built one after the other in ONE process: ms/step of each (configs[1], pipelined, device replay).
Round 2 measured the second one 10-15 % slow; BD_SHARE_STREAMS=0 restores that arrangement (one stream set per engine)."""
import gc, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from big_dreamer_amd import synth
from big_dreamer_amd.engine import DreamerEngine
from big_dreamer_amd.memory import ExperienceReplay

def run(d, steps=40, warm=8):
    dev = torch.device("cuda", 0)
    eng = DreamerEngine(d, None, dev, params=synth.make_params(d, 0))
    rep = synth.make_replay(d, rows=5000, seed=0)
    buf = ExperienceReplay(5000, d.A, 5, False, d.O, dev)
    for k, v in rep.items():
        getattr(buf, k)[:] = v
    buf.idx, buf.full = 0, True
    buf.sync_device()
    def step():
        o, a, r, n = buf.sample(d.B, d.L)
        eng.train_step({"observations": o, "actions": a, "rewards": r, "nonterminals": n}, None, sync_logs=False)
    for _ in range(warm): step()
    eng.join(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): step()
    eng.join(); torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    ids = {k: hex(getattr(eng, k).cuda_stream) for k in ("_s_wm", "_s_bh", "_side")}
    del eng, buf
    gc.collect(); torch.cuda.empty_cache()
    return ms, ids

torch.cuda.set_stream(torch.cuda.Stream())
np.random.seed(0)
d = synth.CONFIG2
res = [run(d) for _ in range(3)]
print("BD_SHARE_STREAMS=" + os.environ.get("BD_SHARE_STREAMS", "1"), " ms/step of engines 1, 2, 3:", [round(r[0], 3) for r in res])
for r in res: print("   streams", r[1])
