"""Worker of tests/test_dp_gpu.py: one rank of a 2-process data-parallel run of the HIP engine on ONE GPU
(gloo transport).  DP_CASE selects the model: Gaussian state observations (default), 64x64 pixels, Categorical latents,
pixels + Categorical.  Two train steps on this rank's batch shard; rank 0 then runs the same two steps in a
single-process engine on the whole batch and compares losses, gradient norms and weights."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from big_dreamer_amd import synth  # noqa: E402
from big_dreamer_amd.engine import DreamerEngine  # noqa: E402
from big_dreamer_amd.parallel import DataParallel  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    free_nats = float(os.environ.get("DP_FREE_NATS", "3.0"))
    case = os.environ.get("DP_CASE", "gauss")
    d = {"gauss": synth.Dims(B=8, L=7, H=5, Be=40, S=10, Hd=36, E=72, A=2, O=4),
         # 64x64 pixel observations: the 20.8 MB-class model bucket with NHWC-stored conv weights crosses the process
         # boundary (flat-bucket all-reduce is layout-blind, but only a run shows it); one sequence per rank
         "pixel": synth.TINY_PIXEL,
         # Categorical latents: the Categorical KL's scalar all-reduce before the free-nats clamp (engine._dynamics_phase)
         "cat": synth.Dims(B=4, L=5, H=4, Be=24, S=15, Hd=20, E=40, A=2, O=5, cat_D=3, cat_C=5),
         # configs[4] as stated, tiny: pixels AND Categorical latents
         "cat_pixel": synth.CAT_PIXEL_TINY}[case]
    hp = dict(free_nats=free_nats)
    if "DP_KL_BALANCE" in os.environ:
        hp["kl_balance"] = float(os.environ["DP_KL_BALANCE"])      # -1: the summed-KL branch (no scalar all-reduce)
    P = synth.make_params(d, 6)
    cu = lambda dct: {k: torch.as_tensor(v).cuda().contiguous() for k, v in dct.items()}
    dp = DataParallel(world, rank)
    eng = DreamerEngine(d, hp, "cuda:0", params=P, world_size=world)
    for step in range(2):
        batch = cu(synth.make_batch(d, 6 + step))
        noise = cu(synth.make_noise(d, 6 + step))
        lb = dp.shard_batch(batch)
        ln = {"obs_prior": dp.shard_batch({"x": noise["obs_prior"]})["x"],
              "obs_post": dp.shard_batch({"x": noise["obs_post"]})["x"],
              "action": dp.shard_rows(noise["action"], d.T, d.B), "entropy": dp.shard_rows(noise["entropy"], d.T, d.B),
              "img_prior": dp.shard_rows(noise["img_prior"], d.T, d.B)}
        logs = eng.train_step(lb, ln)
    torch.cuda.synchronize()
    if rank == 0:
        ref = DreamerEngine(d, hp, "cuda:0", params=P, world_size=1)
        for step in range(2):
            rlogs = ref.train_step(cu(synth.make_batch(d, 6 + step)), cu(synth.make_noise(d, 6 + step)))
        torch.cuda.synchronize()
        # (logged losses are rank-local shard means; kl_loss is global only in the balanced form, whose KL sum is all-reduced
        # before the clamp -- in the summed form rank 0 logs the mean over ITS rows)
        for k in ("grad_norm_model", "grad_norm_actor", "grad_norm_critic") + (("kl_loss",) if hp.get("kl_balance") != -1 else ()):
            assert abs(logs[k] - rlogs[k]) <= 1e-5 + 1e-4 * abs(rlogs[k]), (k, logs[k], rlogs[k])
        worst = 0.0
        for g in ("model", "actor", "critic"):
            a, b = eng.groups[g].flat, ref.groups[g].flat
            worst = max(worst, float((a - b).abs().max()))
        assert worst < 2e-6, f"weights after 2 DP steps differ from the full-batch run by {worst}"
        print(f"DP_GPU_OK case={case} world={world} free_nats={free_nats} kl_loss={rlogs['kl_loss']:.4f} "
              f"max_weight_diff={worst:.3e}")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
