#!/usr/bin/env python3
"""bench.py -- latent transitions/sec of the Dreamer world-model training step on MI355X.

    python bench.py --gpus N --steps K --warmup W
(N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

One step = one full Dreamer.train_step (RSSM observe scan + imagination, forward + backward + three
clip/Adam updates; reference src/dreamer.py:253-393) on one replay batch of BASELINE.json configs[1]:
state observations, belief=200 state=30 hidden=200 embedding=1024, batch=50 chunk=50 horizon=15, fp32.
Per GPU the batch is fixed (weak scaling): rank r draws its own 50 chunks; world-model / actor / critic
gradients are all-reduced over RCCL.  Replay is synthetic and resident in HBM.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0         # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense fp32 MFMA peak


def algorithmic(d):
    """Per-launch algorithmic FLOPs of the persistent kernels and per-step bytes."""
    F = d.Be + d.S
    gru = 2 * 3 * d.Be * d.Be
    embed = (d.S + d.A) * d.Be
    prior = d.Be * d.Hd + d.Hd * 2 * d.S
    actor = F * d.Hd + 3 * d.Hd * d.Hd + d.Hd * 2 * d.A
    img_fwd = 2 * (actor + embed + gru + prior)                   # FLOP per imagined transition
    img_bwd = 2 * (prior + gru + embed + (actor - F * d.Hd))      # layer-0 dgrad is not needed (detached input)
    obs_fwd = 2 * (embed + gru + d.Be * d.Hd + d.Hd * 2 * d.S)    # posterior hidden (belief half) + head
    rows_img = d.Hm * d.N
    flops = {"imagine_fwd": img_fwd * rows_img, "imagine_bwd": img_bwd * rows_img,
             "observe_fwd": obs_fwd * d.N, "observe_bwd": obs_fwd * d.N}
    # SURVEY.md section 8d algorithmic bytes: observe 4*(E+A+1+2S+Be+6S), imagine 4*(101A+2S+Be+3)+66, x3 fwd+bwd
    obs_b = 4 * (d.E + d.A + 1 + 2 * d.S + d.Be + 6 * d.S)
    img_b = 4 * (101 * d.A + 2 * d.S + d.Be + 3) + 4 * (d.Be + d.S) / d.Hm
    weights = 4 * 1.22e6 + 3 * 4 * 167e3
    step_bytes = 3 * (d.N * obs_b + rows_img * img_b + weights)
    return flops, step_bytes


def measured_traffic(kernel):
    """HBM bytes per launch of `kernel` from the newest committed rocprofv3 PMC summary (profiles/*_traffic.json:
    separate --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE doubled for gfx950).  None if absent."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")))
    if not files:
        return None, None
    t = json.load(open(files[-1])).get(kernel)
    return (t["hbm_bytes_per_launch"], os.path.basename(files[-1])) if t else (None, None)


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def usable_cpus():
    """CPUs this process may really use: affinity mask capped by the cgroup quota (a GPU box exposes the
    whole host in os.cpu_count() but grants a share)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(d, budget_s=15.0, max_steps=8):   # (pixel: ~8 s per step -> 1-2 steps)
    """Oracle (CPU restatement of the reference, parity-pinned) timed on this host's cores: reported only.
    Bounded sample: full train_steps until ~budget_s of CPU work (at least 1, at most max_steps)."""
    from big_dreamer_amd import synth
    from oracle import dreamer_oracle as O
    cores = min(usable_cpus(), 32)
    torch.set_num_threads(cores)
    log(f"cpu_baseline: os.cpu_count()={os.cpu_count()} usable={usable_cpus()} -> {cores} threads")
    P, batch, noise = synth.make_params(d, 0), synth.make_batch(d, 0), synth.make_noise(d, 0)
    od = O.OracleDreamer(P, dict(planning_horizon=d.H))
    t0 = time.perf_counter()
    od.train_step(batch, noise, keep=False)            # warm-up
    log(f"cpu_baseline: warm-up step {time.perf_counter() - t0:.2f} s")
    steps, t0 = 0, time.perf_counter()
    while steps < max_steps and (steps == 0 or time.perf_counter() - t0 < budget_s):
        od.train_step(batch, noise, keep=False)
        steps += 1
    dt = time.perf_counter() - t0
    return {"value": d.transitions_per_step * steps / dt, "unit": "latent transitions/s",
            "cores": cores, "kind": "port",
            "sample": f"{steps} full train_steps (batch=50 chunk=50 H=15) after 1 warm-up, torch fp32 CPU, "
                      f"{dt / steps * 1e3:.0f} ms/step"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pixel", action="store_true",
                    help="BASELINE.json configs[2] (64x64 pixel observations, action dim 17) instead of the default configs[1]")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist_on = world > 1 or os.environ.get("BD_FORCE_DP", "0") == "1"    # BD_FORCE_DP=1: one-rank rehearsal of the RCCL path
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)

    from big_dreamer_amd import synth
    from big_dreamer_amd.engine import DreamerEngine
    from big_dreamer_amd.memory import ExperienceReplay

    # Drive the engine from a non-blocking stream rather than the legacy null stream, which synchronises implicitly with
    # every blocking stream of the process; the fences below are device-wide synchronisations.
    torch.cuda.set_stream(torch.cuda.Stream(dev))
    d = synth.CONFIG3 if args.pixel else synth.CONFIG2
    np.random.seed(rank)
    torch.manual_seed(rank)
    # (one communicator: the engine issues the actor / critic all-reduces one host step late so that they never sit in
    # front of the next step's world-model all-reduce -- engine._optimizer_step_or_defer)
    eng = DreamerEngine(d, None, dev, params=synth.make_params(d, 0), world_size=world)
    rep = synth.make_replay(d if not args.pixel else synth.Dims(A=d.A, O=3), rows=5000, seed=0)
    buf = ExperienceReplay(5000, d.A, 5, args.pixel, d.O, dev)
    if args.pixel:      # uniform uint8 frames
        rep["observations"] = np.random.default_rng(0).integers(0, 256, size=(5000, 3, 64, 64), dtype=np.uint8)
    for k, v in rep.items():
        getattr(buf, k)[:] = v
    buf.idx, buf.full = 0, True
    buf.sync_device()

    def step():
        o, a, r, n = buf.sample(d.B, d.L)
        eng.train_step({"observations": o, "actions": a, "rewards": r, "nonterminals": n}, None, sync_logs=False)

    log(f"rank {rank}/{world}: engine built, replay resident; warm-up {args.warmup} steps")
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    log("warm-up done; timing")
    # HIP-event spans on every 5th step of the timed region (each span costs two queue packets); the host's cyclic
    # garbage collector is parked for the timed loop (a generation-2 pass over the imported modules stalls the
    # enqueueing thread for ~40 ms, i.e. ten steps of GPU work)
    import gc
    gc.collect()
    gc.freeze()
    eng.enable_timers(True, every=5 if args.steps >= 10 else 1)

    def fence():
        eng.join()          # everything queued on the engine's streams, incl. optimiser steps it issues one step late
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    log(f"timed {args.steps} steps in {dt * 1e3:.1f} ms")
    logs = eng.logs()
    if dist_on:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    kt = eng.timer_summary()          # {kernel: (avg ms, launches)} from HIP events on the launch stream
    eng.enable_timers(False)

    if rank == 0:
        flops, step_bytes = algorithmic(d)
        # dominant kernel = the persistent kernel that carries most of the path's algorithmic FLOPs (the imagination
        # forward: 31.9 of the 63.6 GFLOP of the four scans); every kernel's rate is listed in kernel_tflops
        dom = max((k for k in kt if k in flops), key=lambda k: flops[k])
        ach = flops[dom] / (kt[dom][0] * 1e-3) / 1e12
        traffic, traffic_src = measured_traffic(dom)
        out = {
            "metric": "latent transitions/sec (RSSM + imagination) at batch=50 chunk=50 H=15",
            "value": d.transitions_per_step * args.steps * world / dt,
            "unit": "latent transitions/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "schedule": ("cross-step pipeline on 3 HIP streams (dynamics learning k+1 | behaviour learning k | critic k)"
                         if eng.pipeline else "serial, one stream"),
            "config": {"workload": ("BASELINE.json configs[2]: 64x64 pixel-obs Dreamer train_step (conv stacks on "
                                    + ("this library's gather-GEMM kernels" if eng.conv_hip else "MIOpen") + "), "
                                    "belief=200 state=30 hidden=200 embedding=1024 action=17, batch=50/GPU chunk=50 H=15")
                       if args.pixel else
                       ("BASELINE.json configs[1]: state-obs Dreamer train_step, belief=200 state=30 "
                        "hidden=200 embedding=1024 action=1 obs=3, batch=50/GPU chunk=50 H=15"),
                       "global_batch": d.B * world, "parallelism": f"dp{world}"},
            "roofline": {"bound": "mfma", "kernel": dom, "achieved": ach, "peak": FP32_MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": ach / FP32_MFMA_PEAK_TFLOPS, "traffic": traffic,
                         "traffic_source": traffic_src,
                         "avg_launch_ms": kt[dom][0], "launches_timed": kt[dom][1],
                         "algorithmic_flop_per_launch": flops[dom]},
            "hbm_roofline_whole_step": {"achieved": step_bytes / (dt / args.steps) / 1e9, "peak": HBM_PEAK_GBPS,
                                        "unit": "GB/s", "frac": step_bytes / (dt / args.steps) / 1e9 / HBM_PEAK_GBPS,
                                        "algorithmic_bytes_per_step": step_bytes},
            "kernel_tflops": {k: round(flops[k] / (kt[k][0] * 1e-3) / 1e12, 2) for k in flops if k in kt},
            "kernel_ms": {k: round(v[0], 4) for k, v in kt.items()},
            "losses": {k: round(v, 5) for k, v in logs.items()},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(d)
        print(json.dumps(out))
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
