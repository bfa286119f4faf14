"""Worker of tests/test_parallel_cpu.py: one gloo rank.  Each rank computes the world-model, actor and critic
gradients of ITS batch shard with the CPU oracle, using the data-parallel rules of big_dreamer_amd.parallel
(global-count scaling, KL sum all-reduce before the free-nats clamp, SUM all-reduce of the flat gradients), and
rank 0 compares the result with the oracle's single-process gradients on the whole batch."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from big_dreamer_amd import synth  # noqa: E402
from big_dreamer_amd.parallel import DataParallel  # noqa: E402
from oracle import dreamer_oracle as O  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.set_num_threads(2)
    free_nats = float(os.environ.get("DP_FREE_NATS", "3.0"))
    d = synth.Dims(B=6, L=5, H=4, Be=24, S=6, Hd=20, E=40, A=2, O=5)
    P = synth.make_params(d, 4)
    batch = {k: torch.as_tensor(v) for k, v in synth.make_batch(d, 4).items()}
    noise = {k: torch.as_tensor(v) for k, v in synth.make_noise(d, 4).items()}
    hp = dict(planning_horizon=d.H, free_nats=free_nats)
    # one process group per optimiser, as bench.py builds them for the pipelined engine (here: gloo)
    dp = DataParallel(world, rank, groups=DataParallel.make_phase_groups("gloo"))

    # ---- this rank's shard ----
    lb = dp.shard_batch(batch)
    ln = {"obs_prior": dp.shard_batch({"x": noise["obs_prior"]})["x"], "obs_post": dp.shard_batch({"x": noise["obs_post"]})["x"],
          "action": dp.shard_rows(noise["action"], d.T, d.B), "entropy": dp.shard_rows(noise["entropy"], d.T, d.B),
          "img_prior": dp.shard_rows(noise["img_prior"], d.T, d.B)}
    od = O.OracleDreamer(P, hp)
    Bl = d.B // world
    Nl = d.T * Bl
    _, obs_loss, rew_loss, _, inter = od.world_model_forward(lb, ln)
    # local SUMS scaled by 1/global count (mean_grad_scale), exactly what the kernels do
    scale = dp.mean_grad_scale(Nl)
    qm, qs, pm, ps = (inter[k] for k in ("posterior_means", "posterior_stds", "prior_means", "prior_stds"))
    kl_el_lhs = O.kl_normal(qm.detach(), qs.detach(), pm, ps)
    kl_el_rhs = O.kl_normal(qm, qs, pm.detach(), ps.detach())
    kl_sum = kl_el_rhs.detach().sum().reshape(1).clone()
    dp.allreduce_sum_(kl_sum, "model")                                     # ONE float before the clamp decision
    kl_mean = kl_sum / (d.N * d.S)
    gate = 1.0 if float(kl_mean) > free_nats else (0.5 if float(kl_mean) == free_nats else 0.0)
    kl_scale = gate / (d.N * d.S)
    loss = (obs_loss * Nl + rew_loss * Nl) * scale + od.hp["kl_loss_weight"] * kl_scale * (
        od.hp["kl_balance"] * kl_el_lhs.sum() + (1 - od.hp["kl_balance"]) * kl_el_rhs.sum())
    g = torch.autograd.grad(loss, od.model_params, allow_unused=True)
    flat = torch.cat([(torch.zeros_like(p) if gi is None else gi).reshape(-1) for gi, p in zip(g, od.model_params)])
    dp.allreduce_sum_(flat, "model")                                       # SUM of 1/global-count-scaled grads = global mean grad

    # ---- reference: the whole batch in one process ----
    if rank == 0:
        full = O.OracleDreamer(P, hp)
        model_loss, _, _, kl, _ = full.world_model_forward(batch, noise)
        gf = torch.autograd.grad(model_loss, full.model_params, allow_unused=True)
        want = torch.cat([(torch.zeros_like(p) if gi is None else gi).reshape(-1) for gi, p in zip(gf, full.model_params)])
        err = float((flat - want).abs().max())
        ref = float(want.abs().max())
        assert err <= 2e-6 + 1e-5 * ref, f"DP gradient mismatch: max err {err} (ref scale {ref})"
        want_kl = float(kl)
        got_kl = od.hp["kl_balance"] * max(float(kl_mean), free_nats) + (1 - od.hp["kl_balance"]) * max(float(kl_mean), free_nats)
        assert abs(got_kl - want_kl) < 1e-5, (got_kl, want_kl)
        print(f"DP_OK world={world} free_nats={free_nats} max_err={err:.3e} ref={ref:.3e}")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
