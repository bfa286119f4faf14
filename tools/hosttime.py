#!/usr/bin/env python3
"""Host enqueue time per train step and the device timeline of the engine's spans (pipeline diagnosis).

    python tools/hosttime.py [--steps 12] [--profile]
Prints the host time each `train_step(sync_logs=False)` call took to enqueue, then for the last steps the start / end
of every span relative to a base event (ms), per stream, so overlap between dynamics learning of step k+1 and
behaviour learning of step k is visible.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--profile", action="store_true")
    ap.add_argument("--no-timers", action="store_true")
    ap.add_argument("--no-gc", action="store_true")
    args = ap.parse_args()
    from big_dreamer_amd import synth
    from big_dreamer_amd.engine import DreamerEngine
    from big_dreamer_amd.memory import ExperienceReplay
    d = synth.CONFIG2
    dev = torch.device("cuda", 0)
    if os.environ.get("BD_FORCE_DP", "0") == "1":      # one-rank rehearsal of the data-parallel schedule over RCCL
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29519")
        torch.distributed.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
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

    if os.environ.get("BD_MAIN_STREAM", "1") == "0":
        # diagnosis: put the caller back on the legacy null stream (the engine moves a data-parallel caller off it: the
        # null stream synchronises implicitly with RCCL's blocking stream and serialises the pipeline)
        torch.cuda.set_stream(torch.cuda.default_stream())
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    base = torch.cuda.Event(enable_timing=True)
    base.record()
    eng.enable_timers(not args.no_timers)
    if args.no_gc:
        import gc
        gc.disable()
    host = []
    t_all = time.perf_counter()
    for _ in range(args.steps):
        t0 = time.perf_counter()
        o, a, r, n = buf.sample(d.B, d.L)
        t1 = time.perf_counter()
        eng.train_step({"observations": o, "actions": a, "rewards": r, "nonterminals": n}, None, sync_logs=False)
        host.append((t1 - t0, time.perf_counter() - t1))
    t_enq = time.perf_counter() - t_all
    torch.cuda.synchronize()
    t_tot = time.perf_counter() - t_all
    print(f"pipeline={eng.pipeline}  enqueue of {args.steps} steps: {t_enq * 1e3:.1f} ms host; drained after {t_tot * 1e3:.1f} ms "
          f"({t_tot / args.steps * 1e3:.2f} ms/step)")
    print("host ms per step (sample, train_step):", [(round(a * 1e3, 2), round(b * 1e3, 2)) for a, b in host][:16])
    print("stalls > 3 ms at steps:", [(i, round(b * 1e3, 1)) for i, (a, b) in enumerate(host) if b > 3e-3])
    ev = eng._timer_events if not args.no_timers else {}
    names = list(ev)
    for k in range(max(0, args.steps - 3), args.steps):
        row = []
        for n in names:
            e0, e1 = ev[n][k]
            row.append((base.elapsed_time(e0), base.elapsed_time(e1), n))
        row.sort()
        print(f"step {k}: " + "  ".join(f"{n}[{a:.2f},{b:.2f}]" for a, b, n in row))
    if args.profile:
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(10):
            step()
        pr.disable()
        torch.cuda.synchronize()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(35)


if __name__ == "__main__":
    main()
