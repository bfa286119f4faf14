#!/usr/bin/env python3
"""bench.py -- latent transitions/sec of the Dreamer world-model training step on MI355X.

    python bench.py --gpus N --steps K --warmup W
(N>1: either under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py
--gpus N ...`, or plain `python bench.py --gpus N`, which then spawns its own N rank processes -- big_dreamer_amd/launch.py)

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

# (before torch is imported: the HIP runtime reads it when it initialises; big_dreamer_amd/__init__.py has the measurements)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

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
    # Categorical latents: the one-hot state enters a first layer as a gather of D rows (D adds per output), not as S MACs
    s_in = d.cat_D if d.categorical else d.S
    embed = (s_in + d.A) * d.Be
    prior = d.Be * d.Hd + d.Hd * d.head_out
    actor = (d.Be + s_in) * d.Hd + 3 * d.Hd * d.Hd + d.Hd * 2 * d.A
    F = d.Be + s_in
    img_fwd = 2 * (actor + embed + gru + prior)                   # FLOP per imagined transition
    # backward: layer-0 dgrad of the actor is not needed (detached input).  With BD_ACTOR_BWD_CHAIN=1 (default,
    # engine._behaviour_phase) the actor's hidden-layer dgrads (a4^T, a3^T, a2^T, a1^T) leave the scan and run as one
    # dense chain over all Hm x N rows: they are credited to the `actor_hidden_bwd` span, not to the scan
    actor_hidden = 3 * d.Hd * d.Hd + d.Hd * 2 * d.A
    actor_chain = os.environ.get("BD_ACTOR_BWD_CHAIN", "1") == "1"
    img_bwd = 2 * (prior + gru + embed + (0 if actor_chain else actor_hidden))
    obs_fwd = 2 * (embed + gru + d.Be * d.Hd + d.Hd * d.head_out)  # posterior hidden (belief half) + head
    rows_img = d.Hm * d.N
    flops = {"imagine_fwd": img_fwd * rows_img, "imagine_bwd": img_bwd * rows_img,
             "observe_fwd": obs_fwd * d.N, "observe_bwd": obs_fwd * d.N}
    if actor_chain:
        flops["actor_hidden_bwd"] = 2 * actor_hidden * rows_img
    # dense head chains over the imagined rows (F -> Hd x4 -> 1) and the weight-gradient GEMMs of the three passes:
    # spans of several launches, reported in kernel_tflops only (never the dominant kernel)
    head = 2 * (F * d.Hd + 3 * d.Hd * d.Hd + d.Hd)
    flops.update({"img_heads_fwd": 2 * head * rows_img,                 # reward + value heads
                  "img_heads_bwd": 2 * head * rows_img,                 # both with d/d features
                  "critic_fwd_bwd": (2 * head - 2 * F * d.Hd) * rows_img,   # forward + backward without d/d features
                  "wgrad_gemm_critic": head * rows_img,
                  "wgrad_gemm_actor": 2 * (actor) * rows_img})
    # SURVEY.md section 8d algorithmic bytes: observe 4*(E+A+1+2S+Be+6S), imagine 4*(101A+2S+Be+3)+66, x3 fwd+bwd
    if d.categorical:   # per transition: read emb, action, mask, S draws; write belief, D indices (as floats), 2 x S logits
        obs_b = 4 * (d.E + d.A + 1 + d.S + d.Be + d.cat_D + 2 * d.S)
        img_b = 4 * (101 * d.A + d.S + d.Be + d.cat_D + 3) + 4 * (d.Be + d.cat_D) / d.Hm
    else:
        obs_b = 4 * (d.E + d.A + 1 + 2 * d.S + d.Be + 6 * d.S)
        img_b = 4 * (101 * d.A + 2 * d.S + d.Be + 3) + 4 * (d.Be + d.S) / d.Hm
    weights = 4 * 1.22e6 + 3 * 4 * 167e3
    step_bytes = 3 * (d.N * obs_b + rows_img * img_b + weights)
    return flops, step_bytes


def _traffic_record(kernel, kind=""):
    """(record, file name) of `kernel` in the newest committed rocprofv3 PMC summary of the SAME workload
    (profiles/*_traffic.json for configs[1], *_pixel_traffic.json for configs[2], *_cat_traffic.json / *_catstate_traffic.json
    for configs[4] on pixels / state observations), or None."""
    import glob
    tagged = lambda f: next((k for k in ("pixel", "catstate", "cat") if f"_{k}_traffic" in os.path.basename(f)), "")
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")) if tagged(f) == kind)
    if not files:
        return None
    t = json.load(open(files[-1])).get(kernel)
    return (t, os.path.basename(files[-1])) if t else None


def measured_traffic(kernel, kind=""):
    """HBM bytes per launch of `kernel` from that record (separate --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE
    doubled for gfx950).  (None, None) if absent."""
    rec = _traffic_record(kernel, kind)
    return (rec[0]["hbm_bytes_per_launch"], rec[1]) if rec else (None, None)


MFMA_F32_16X16X4_FLOP = 2048      # 16 x 16 x 4 MACs x 2 per v_mfma_f32_16x16x4_f32 (every kernel of the path issues this shape)


def counter_checked_tflops(flops, kt, kind=""):
    """kernel_tflops with a self-check: an entry whose ALGORITHMIC FLOPs per launch exceed what the kernel EXECUTES per
    launch by the newest committed PMC record of the same workload (SQ_INSTS_VALU_MFMA_F32 x 2048,
    profiles/*_traffic.json) is a wrong FLOP model, not a fast kernel: it is refused (listed under `refused`) rather
    than printed.  Returns (tflops dict, refused dict)."""
    ok, refused = {}, {}
    for k in flops:
        if k not in kt:
            continue
        rate = round(flops[k] / (kt[k][0] * 1e-3) / 1e12, 2)
        rec = _traffic_record(k, kind)
        if rec is not None and rec[0].get("mfma_f32_insts"):
            executed = rec[0]["mfma_f32_insts"] * MFMA_F32_16X16X4_FLOP
            if flops[k] > 1.02 * executed:
                refused[k] = {"algorithmic_flop": flops[k], "executed_flop_by_counter": executed, "source": rec[1]}
                continue
        ok[k] = rate
    return ok, refused


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
    hp = dict(planning_horizon=d.H)
    if d.categorical:
        hp["categorical"] = (d.cat_D, d.cat_C)
    od = O.OracleDreamer(P, hp)
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
            "sample": f"{steps} full train_steps (batch={d.B} chunk={d.L} H={d.H}) after 1 warm-up, torch fp32 CPU, "
                      f"{dt / steps * 1e3:.0f} ms/step"}


def decoder_wgrad_flops(d):
    """Algorithmic FLOPs of ONE launch of the decoder's grouped weight-gradient GEMM (the dominant kernel of the pixel
    step): dW of ConvT(32,3,k6) / ConvT(64,32,k6) / ConvT(128,64,k5) as gathered-window GEMMs, the 1x1 -> 5x5 layer and
    Linear(Be+S, E) as plain GEMMs (conv_stack.backward_decoder), 2*M*N*K each, per image x N images."""
    F = d.Be + d.S
    per_img = 2 * (900 * 32 * 108 + 169 * 64 * 1152 + 25 * 128 * 1600 + d.E * 3200 + d.E * F)
    return per_img * d.N


def pixel_roofline(d, kt):
    """Roofline record of the pixel step's dominant kernel: the decoder's grouped weight-gradient GEMM (24 % of the GPU
    time of configs[2], profiles/*_pixel_kernel_stats.csv), bracketed alone by HIP events (span wgrad_gemm_model_early)."""
    dom = "wgrad_gemm_model_early"
    fl = decoder_wgrad_flops(d)
    ach = fl / (kt[dom][0] * 1e-3) / 1e12
    traffic, src = measured_traffic("wgrad_decoder", "pixel")
    return {"bound": "mfma", "kernel": "wgrad_wide_kernel (decoder conv weight gradients, one grouped launch)",
            "achieved": ach, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP32_MFMA_PEAK_TFLOPS,
            "traffic": traffic, "traffic_source": src, "avg_launch_ms": kt[dom][0], "launches_timed": kt[dom][1],
            "algorithmic_flop_per_launch": fl}


def run_workload(d, pixel, args, rank, world, dev, dist_on, steps, warmup, timers=True):
    """Build engine + HBM-resident synthetic replay for `d`, run warmup + exactly `steps` timed train steps between
    barrier + synchronise fences.  Returns (dt max over ranks, HIP-event spans, last logs, engine)."""
    import torch.distributed as dist
    from big_dreamer_amd import synth
    from big_dreamer_amd.engine import DreamerEngine
    from big_dreamer_amd.memory import ExperienceReplay
    eng = DreamerEngine(d, None, dev, params=synth.make_params(d, 0), world_size=world)
    rep = synth.make_replay(d if not pixel else synth.Dims(A=d.A, O=3), rows=5000, seed=0)
    buf = ExperienceReplay(5000, d.A, 5, pixel, d.O, dev)
    if pixel:      # uniform uint8 frames
        rep["observations"] = np.random.default_rng(0).integers(0, 256, size=(5000, 3, 64, 64), dtype=np.uint8)
    for k, v in rep.items():
        getattr(buf, k)[:] = v
    buf.idx, buf.full = 0, True
    buf.sync_device()

    def step():
        o, a, r, n = buf.sample(d.B, d.L)
        eng.train_step({"observations": o, "actions": a, "rewards": r, "nonterminals": n}, None, sync_logs=False)

    log(f"rank {rank}/{world}: engine built ({'pixel' if pixel else 'state'} obs, A={d.A}), replay resident; warm-up {warmup} steps")
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    # HIP-event spans on every 5th step of the timed region (each span costs two queue packets)
    if timers:
        eng.enable_timers(True, every=5 if steps >= 10 else 1)

    def fence():
        eng.flush_optimizers()   # optimiser steps the data-parallel schedule issues one host step late (collectives)
        eng.join()               # everything queued on the engine's streams
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    log(f"timed {steps} steps in {dt * 1e3:.1f} ms")
    logs = eng.logs()
    if dist_on:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    kt = eng.timer_summary() if timers else {}
    eng.enable_timers(False)
    return dt, kt, logs, eng


def surface_ms_per_step(d, dev, steps, warmup, bursts_of=(5, 50)):
    """The same workload through the reference's call surface: ``Dreamer(params, env).train_step()`` in bursts of
    collect_interval steps, reading the LAST log dict of each burst as src/main.py:103-108 does.  One agent, timed with
    every burst length of `bursts_of` (5 = the reference's config default, 50 = its README's example command).
    Returns ({burst: ms per step}, last logs)."""
    import gc
    from big_dreamer_amd import synth
    from big_dreamer_amd.config import load_config
    from big_dreamer_amd.dreamer import Dreamer

    class _Env:
        action_size, observation_size = d.A, d.O

    params = load_config([f"batch_size={d.B}", f"seq_len={d.L}", f"planning_horizon={d.H}", f"belief_size={d.Be}",
                          f"state_size={d.S}", f"hidden_size={d.Hd}", f"embedding_size={d.E}", "experience_size=5000",
                          "pixel_observation=false"])
    torch.manual_seed(0)
    agent = Dreamer(params, _Env(), device=str(dev))
    rep = synth.make_replay(d, rows=5000, seed=0)
    for k, v in rep.items():
        getattr(agent.buffer, k)[:] = v
    agent.buffer.idx, agent.buffer.full = 0, True
    agent.buffer.sync_device()
    gc.collect()
    gc.freeze()          # as for the engine timing: a generation-2 collection stalls the enqueueing thread for ~40 ms
    logs = None
    # (the replay hands out batches from a ring of four buffers and the features are double-buffered by step parity: the
    # weight-gradient descriptor tables are built once per operand-pointer set, with a synchronising H2D copy each)
    for _ in range(max(warmup, 12)):
        logs = agent.train_step()
    float(logs["model_loss"])
    torch.cuda.synchronize()
    res = {}
    for burst in bursts_of:
        bursts = max(1, steps // burst)
        t0 = time.perf_counter()
        for _ in range(bursts):
            for _ in range(burst):
                logs = agent.train_step()
            float(logs["model_loss"])            # the loop reads the burst's last dict (weight_update_per_sec, logging)
        torch.cuda.synchronize()
        res[burst] = (time.perf_counter() - t0) / (bursts * burst) * 1e3
    return res, {k: round(float(v), 5) for k, v in logs.items()}


def child_json(extra_args, timeout=600):
    """Run this script again as a CHILD process (fresh HIP context) and return the JSON object it prints: the legs of the
    default run are separate processes, one after the other, while this one holds no engine.  (Round 2 needed this because
    a second engine in one process ran 10-15 % slow; the cause -- its pool streams landing on shared hardware queues -- is
    fixed in engine._engine_stream and tested, the isolation stays because it also keeps the legs' allocator states apart.)"""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__)] + list(extra_args)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
    if r.returncode != 0:
        raise RuntimeError(f"child {' '.join(extra_args)} failed ({r.returncode}):\n{r.stderr[-2000:]}")
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the surface and pixel-config legs (profiling runs)")
    ap.add_argument("--surface-only", action="store_true", help=argparse.SUPPRESS)    # child process of the default run
    ap.add_argument("--pixel", action="store_true",
                    help="BASELINE.json configs[2] (64x64 pixel observations, action dim 17) instead of the default configs[1]")
    ap.add_argument("--categorical", choices=["pixel", "state"], default=None,
                    help="BASELINE.json configs[4] per GPU: 32x32 Categorical latents (algorithm=dreamerV2), batch 100 = 800/8, "
                         "64x64 pixel observations with action dim 17 ('pixel') or state observations ('state')")
    ap.add_argument("--launch-timeout", type=float, default=900.0,
                    help="self-launched --gpus N only: seconds after which the rank processes are terminated (a stuck "
                         "rendezvous or collective must not hold the GPUs forever)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo: tests only)")
    ap.add_argument("--same-device", action="store_true",
                    help="tests only: every rank on cuda:0 (needs --backend gloo: RCCL refuses two ranks on one device)")
    args = ap.parse_args()

    from big_dreamer_amd import launch
    if args.gpus > 1 and not launch.launched_by_torchrun():
        # plain `python bench.py --gpus N`: become the launcher -- N fresh rank processes of this script, one per GPU;
        # this parent never touches HIP.  Rank 0 prints the JSON line.
        sys.exit(launch.spawn_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus,
                                    timeout=args.launch_timeout))

    if args.surface_only:       # child of the default run: the workload through Dreamer.train_step, nothing else in the process
        from big_dreamer_amd import synth
        torch.cuda.set_device(0)
        res, slogs = surface_ms_per_step(synth.CONFIG2, torch.device("cuda", 0), args.steps, args.warmup)
        print(json.dumps({"ms_per_step": res, "losses": slogs}), flush=True)
        return

    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.same_device else int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist_on = world > 1 or os.environ.get("BD_FORCE_DP", "0") == "1"    # BD_FORCE_DP=1: one-rank rehearsal of the RCCL path
    rccl_ranks = 1
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
        rccl_ranks = dist.get_world_size()

    from big_dreamer_amd import synth

    # Drive the engine from a non-blocking stream rather than the legacy null stream, which synchronises implicitly with
    # every blocking stream of the process; the fences are device-wide synchronisations.
    torch.cuda.set_stream(torch.cuda.Stream(dev))
    d = synth.CONFIG3 if args.pixel else synth.CONFIG2
    if args.categorical:
        d = synth.CONFIG5 if args.categorical == "pixel" else synth.CONFIG5_STATE
        args.pixel = bool(d.pixel)
    np.random.seed(rank)
    torch.manual_seed(rank)
    # the host's cyclic garbage collector is parked for the timed loops (a generation-2 pass over the imported modules
    # stalls the enqueueing thread for ~40 ms, i.e. ten steps of GPU work)
    import gc
    gc.collect()
    gc.freeze()
    # (one communicator: the engine issues the actor / critic all-reduces one host step late so that they never sit in
    # front of the next step's world-model all-reduce -- engine._optimizer_step_or_defer)
    dt, kt, logs, eng = run_workload(d, args.pixel, args, rank, world, dev, dist_on, args.steps, args.warmup)

    out = None
    if rank == 0:
        flops, step_bytes = algorithmic(d)
        # dominant kernel = the persistent kernel that carries most of the path's algorithmic FLOPs (the imagination
        # forward: 31.9 of the 63.6 GFLOP of the four scans); every kernel's rate is listed in kernel_tflops
        scans = ("imagine_fwd", "imagine_bwd", "observe_fwd", "observe_bwd")
        dom = max((k for k in kt if k in flops and k in scans), key=lambda k: flops[k])
        ach = flops[dom] / (kt[dom][0] * 1e-3) / 1e12
        kind = ({"pixel": "cat", "state": "catstate"}[args.categorical] if args.categorical else ("pixel" if args.pixel else ""))
        traffic, traffic_src = measured_traffic(dom, kind)
        ktf, ktf_refused = counter_checked_tflops(flops, kt, kind)
        out = {
            "metric": "latent transitions/sec (RSSM + imagination) at batch=50 chunk=50 H=15" if not args.categorical
            else "latent transitions/sec (RSSM + imagination) at batch=100/GPU chunk=50 H=15, Categorical latents",
            "value": d.transitions_per_step * args.steps * world / dt,
            "unit": "latent transitions/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "rccl_ranks": rccl_ranks, "backend": (args.backend if dist_on else None),
            "schedule": ("cross-step pipeline on 3 HIP streams (dynamics learning k+1 | behaviour learning k | critic k)"
                         if eng.pipeline else "serial, one stream"),
            "config": {"workload": (f"BASELINE.json configs[4] per GPU: dreamerV2 {d.cat_D}x{d.cat_C} Categorical latents, "
                                    + ("64x64 pixel obs, action=17" if d.pixel else "state obs, action=1")
                                    + f", belief=200 hidden=200 embedding=1024, batch={d.B}/GPU (800/8) chunk=50 H=15")
                       if args.categorical else
                       ("BASELINE.json configs[2]: 64x64 pixel-obs Dreamer train_step (conv stacks on "
                                    "this library's gather-GEMM kernels), "
                                    "belief=200 state=30 hidden=200 embedding=1024 action=17, batch=50/GPU chunk=50 H=15")
                       if args.pixel else
                       ("BASELINE.json configs[1]: state-obs Dreamer train_step, belief=200 state=30 "
                        "hidden=200 embedding=1024 action=1 obs=3, batch=50/GPU chunk=50 H=15"),
                       "global_batch": d.B * world, "parallelism": f"dp{world}"},
            "roofline": (pixel_roofline(d, kt) if (args.pixel and not args.categorical and "wgrad_gemm_model_early" in kt) else
                         {"bound": "mfma", "kernel": dom, "achieved": ach, "peak": FP32_MFMA_PEAK_TFLOPS,
                          "unit": "TFLOP/s", "frac": ach / FP32_MFMA_PEAK_TFLOPS, "traffic": traffic,
                          "traffic_source": traffic_src,
                          "avg_launch_ms": kt[dom][0], "launches_timed": kt[dom][1],
                          "algorithmic_flop_per_launch": flops[dom]}),
            "hbm_roofline_whole_step": {"achieved": step_bytes / (dt / args.steps) / 1e9, "peak": HBM_PEAK_GBPS,
                                        "unit": "GB/s", "frac": step_bytes / (dt / args.steps) / 1e9 / HBM_PEAK_GBPS,
                                        "algorithmic_bytes_per_step": step_bytes},
            "kernel_tflops": ktf,
            "kernel_ms": {k: round(v[0], 4) for k, v in kt.items()},
            "losses": {k: round(v, 5) for k, v in logs.items()},
        }
        if ktf_refused:
            out["kernel_tflops_refused"] = ktf_refused
            log(f"kernel_tflops: refused {sorted(ktf_refused)} (algorithmic FLOPs above the MFMA counter of the PMC record)")
        assert dom in ktf, f"the dominant kernel's FLOP model ({dom}) exceeds its MFMA counter: {ktf_refused.get(dom)}"
    del eng
    torch.cuda.empty_cache()
    if rank == 0 and world == 1 and not args.no_secondary and not args.pixel and not args.categorical:
        # (a) the same workload through the reference's call surface (Dreamer.train_step, lazy log dicts): child process
        try:
            sc = child_json(["--surface-only", "--steps", str(args.steps), "--warmup", str(args.warmup)])
            sms = sc["ms_per_step"]["5"]
            out["surface_ms_per_step"] = sms
            out["surface"] = {"ms_per_step": sms, "value": d.transitions_per_step / (sms * 1e-3),
                              "ms_per_step_burst50": sc["ms_per_step"].get("50"),
                              "how": "Dreamer(params, env).train_step() in bursts of collect_interval=5 (burst50: 50, the "
                                     "reference README's example), last log dict of each burst read (src/main.py:103-108), in "
                                     "its own process; `value`/`ms_per_step` above drive the engine directly",
                              "losses": sc["losses"]}
        except Exception as e:          # a leg that fails must not take the headline line with it
            log(f"surface leg failed: {e}")
            out["surface"] = {"error": str(e)[-500:]}
        # (b) BASELINE.json configs[2] (pixels, A=17): its step time and the roofline of ITS dominant kernel, the decoder's
        #     grouped weight-gradient GEMM -- the line `bench.py --pixel` prints, from a child process
        psteps = max(5, min(20, args.steps))
        try:
            pc = child_json(["--pixel", "--no-secondary", "--no-cpu-baseline", "--steps", str(psteps), "--warmup", "3"])
            out["secondary"] = {"workload": pc["config"]["workload"], "value": pc["value"], "unit": pc["unit"],
                                "ms_per_step": pc["ms_per_step"], "steps": pc["steps"], "warmup": pc["warmup"],
                                "roofline": pc["roofline"], "kernel_ms": pc["kernel_ms"], "losses": pc["losses"]}
        except Exception as e:
            log(f"pixel leg failed: {e}")
            out["secondary"] = {"error": str(e)[-500:]}
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(d)
        print(json.dumps(out), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
