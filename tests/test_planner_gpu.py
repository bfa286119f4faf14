"""GPU: the CEM planner (MPCPlanner.forward, src/planner.py) and the PlaNet train step (src/planet.py:310-368) on the
HIP kernels, through the C ABI, against the CPU oracle and the golden vectors generated from the reference."""
import ctypes as C

import numpy as np
import pytest
import torch

from big_dreamer_amd import synth
from tests.helpers import PLANNER_CASES, assert_close, compare_tensor, load_golden

pytestmark = pytest.mark.gpu


def _agent(d, seed, cls="dreamer", extra=()):
    from big_dreamer_amd.config import load_config
    from big_dreamer_amd.dreamer import Dreamer
    from big_dreamer_amd.env import SyntheticEnv
    from big_dreamer_amd.planet import Planet
    params = load_config([f"belief_size={d.Be}", f"state_size={d.S}", f"hidden_size={d.Hd}", f"embedding_size={d.E}",
                          f"batch_size={d.B}", f"seq_len={d.L}", f"planning_horizon={d.H}", "experience_size=400",
                          "seed_steps=120", "max_episode_length=40", *extra])
    env = SyntheticEnv(d.O, d.A, 40, 2, 0)
    agent = (Planet if cls == "planet" else Dreamer)(params, env)
    P = synth.make_params(d, seed)
    for mod in ("transition_model", "observation_model", "reward_model", "encoder"):
        getattr(agent, mod).load_state_dict({k: torch.from_numpy(v) for k, v in P[mod].items()})
    return agent, P, env


def test_cem_refit_kernel_matches_topk_mean_std():
    """bd_cem_refit vs torch.topk + mean + std(unbiased=False) (src/planner.py:74-87); a NaN return ranks first, as
    torch.topk orders it."""
    from big_dreamer_amd import _cabi as cabi
    rng = np.random.Generator(np.random.PCG64(5))
    for H, B, cand, top, A, nan in ((15, 1, 1000, 100, 1, False), (7, 3, 257, 19, 17, False), (4, 2, 64, 64, 2, False),
                                    (5, 2, 100, 10, 3, True)):
        ret = rng.standard_normal((B, cand), dtype=np.float32)
        if nan:
            ret[1, 7] = np.nan
        act = rng.standard_normal((H, B * cand, A), dtype=np.float32)
        tr, ta = torch.from_numpy(ret), torch.from_numpy(act)
        _, topk = tr.topk(top, dim=1, largest=True, sorted=False)
        if nan:
            assert 7 in topk[1].tolist()
        topk = topk + cand * torch.arange(0, B).unsqueeze(1)
        best = ta[:, topk.view(-1)].reshape(H, B, top, A)
        want_m, want_s = best.mean(dim=2), best.std(dim=2, unbiased=False)
        dr, da = tr.cuda(), ta.cuda()
        m, s = torch.empty(H, B, A, device="cuda"), torch.empty(H, B, A, device="cuda")
        cabi.check(cabi.lib.bd_cem_refit(dr.data_ptr(), 1, da.data_ptr(), H, B, cand, top, A, m.data_ptr(), s.data_ptr(),
                                         cabi.stream()))
        torch.cuda.synchronize()
        assert_close("refit mean", m.cpu().numpy(), want_m.numpy(), 1e-6, 1e-5)
        assert_close("refit std", s.cpu().numpy(), want_s.numpy(), 1e-6, 1e-5)
    assert cabi.lib.bd_cem_refit(dr.data_ptr(), 1, da.data_ptr(), H, B, cand, cand + 1, A, m.data_ptr(), s.data_ptr(),
                                 cabi.stream()) != 0
    assert b"top_candidates" in cabi.lib.bd_last_error()


@pytest.mark.parametrize("fuse", ["1", "0"])
@pytest.mark.parametrize("name", list(PLANNER_CASES))
def test_planner_vs_oracle_and_reference_golden(name, fuse, monkeypatch):
    """Both forms of the rollout (reward model inside the persistent kernel / batched over all H steps afterwards).
    MPCPlanner.forward on the HIP kernels: candidate returns of every CEM iteration and the planned action against
    the oracle and the reference's own run (golden), same injected noise.  A return is a sum of H reward predictions
    after an H-step fp32 recurrence: 1e-4 abs; the action is a mean over the selected candidates: 2e-4."""
    from big_dreamer_amd.planner import MPCPlanner
    from oracle import dreamer_oracle as O
    monkeypatch.setenv("BD_PLAN_FUSE", fuse)
    d, B, H, iters, cand, top, seed, full = PLANNER_CASES[name]
    g = load_golden(name)
    agent, P, _ = _agent(d, seed)
    mpc = MPCPlanner(d.A, H, iters, cand, top, agent.transition_model, agent.reward_model)
    nz = synth.make_planner_noise(d, B, H, iters, cand, seed)
    trace = []
    act = mpc(torch.from_numpy(g["belief"]).cuda(), torch.from_numpy(g["state"]).cuda(),
              _noise={k: torch.from_numpy(v).cuda() for k, v in nz.items()}, _trace=trace)
    torch.cuda.synchronize()
    assert tuple(act.shape) == (B, d.A) and len(trace) == iters
    tP = {m: {k: torch.as_tensor(v) for k, v in sd.items()} for m, sd in P.items()}
    otrace = []
    with torch.no_grad():
        want = O.mpc_planner(tP, torch.as_tensor(g["belief"]), torch.as_tensor(g["state"]), d.A, H, iters, cand, top,
                             torch.as_tensor(nz["action"]), torch.as_tensor(nz["state"]), otrace)
    for it in range(iters):
        got = trace[it].cpu().numpy()
        assert_close(f"returns{it} (oracle)", got, otrace[it][0].numpy(), 1e-4, 1e-4)
        compare_tensor(g, f"returns{it}", got, full, 1e-4, 1e-4)
    assert_close("action (oracle)", act.cpu().numpy(), want.numpy(), 2e-4, 2e-4)
    assert_close("action (golden)", act.cpu().numpy(), g["action"], 2e-4, 2e-4)


@pytest.mark.parametrize("fuse", ["1", "0"])
def test_planner_rollout_matches_unfused_modules(fuse, monkeypatch):
    """The rollout (both forms) against the same agent's own modules, one launch per piece: TransitionModel.forward(
    embeddings=None) + reward_model, as the reference composes them (src/planner.py:65-72)."""
    from big_dreamer_amd import _cabi as cabi
    monkeypatch.setenv("BD_PLAN_FUSE", fuse)
    d, seed, B, cand, H = synth.SMALL, 3, 3, 50, 6
    agent, P, _ = _agent(d, seed)
    eng = agent.engine
    gen = torch.Generator(device="cuda").manual_seed(1)
    rn = lambda *s: torch.randn(*s, device="cuda", generator=gen)
    belief, state = 0.5 * rn(B, d.Be), rn(B, d.S)
    eps_a, eps_s = rn(1, H, B, cand, d.A), rn(1, H, B * cand, d.S)
    trace = []
    eng.plan(belief, state, H, 1, cand, 5, eps_a, eps_s, trace)
    actions = eng.buf("plan_actions", H, B * cand, d.A).clone()
    torch.cuda.synchronize()
    assert_close("actions", actions.cpu().numpy(), eps_a[0].reshape(H, B * cand, d.A).cpu().numpy(), 0, 0)   # N(0, I) start
    xb = belief.unsqueeze(1).expand(B, cand, d.Be).reshape(-1, d.Be)
    xs = state.unsqueeze(1).expand(B, cand, d.S).reshape(-1, d.S)
    beliefs, states, _, _, _ = agent.transition_model(xs, actions, xb, _noise=(eps_s[0], None))
    want = agent.reward_model(beliefs.view(-1, d.Be), states.view(-1, d.S)).view(H, -1).sum(dim=0)
    assert_close("returns", trace[0].cpu().numpy(), want.cpu().numpy(), 2e-5, 2e-5)
    bad = cabi.PlanArgs()
    assert cabi.lib.bd_plan_rollout(C.byref(bad), cabi.stream()) != 0 and b"bad dims" in cabi.lib.bd_last_error()


def test_planet_train_steps_vs_oracle_and_reference_golden():
    """Planet.train_step x2 (dynamics learning, summed free-nats KL): logs, clipped gradients, post-Adam weights."""
    from oracle import dreamer_oracle as O
    d, seed = synth.TINY, 8
    g = load_golden("tiny_planet")
    agent, P, _ = _agent(d, seed, cls="planet", extra=("free_nats=0.05",))
    eng = agent.engine
    od = O.OracleDreamer(P, dict(kl_balance=-1, free_nats=0.05, planning_horizon=d.H))
    batch = synth.make_batch(d, seed)
    db = {k: torch.as_tensor(v).cuda().contiguous() for k, v in batch.items()}
    for step in range(2):
        nz = synth.make_noise(d, seed + step)
        ologs = od.planet_train_step(batch, nz)
        logs = eng.world_model_step(db, {k: torch.as_tensor(v).cuda() for k, v in nz.items() if k.startswith("obs_")})
        torch.cuda.synchronize()
        for k, v in ologs.items():
            assert_close(f"s{step}.{k}", logs[k], v, 2e-5, 5e-5)
            assert_close(f"s{step}.{k} (golden)", logs[k], g[f"step{step}.log.{k}"], 2e-5, 5e-5)
        gn = od.last["grad_norms"]["model"]
        assert_close(f"s{step}.grad_norm", logs["grad_norm_model"], gn, 1e-6, 1e-3)
        coef = min(1.0, od.hp["grad_clip_norm"] / (gn + 1e-6))
        i = 0
        for mod in O.MODEL_MODULES:
            for k, p in od.P[mod].items():
                want = od.last["model_grads"][i].numpy() * coef
                scale = float(np.abs(want).max()) + 1e-12
                assert_close(f"s{step}.grad.{mod}.{k}", eng.G(mod, k).cpu().numpy(), want, 2e-3 * scale + 1e-9, 2e-3)
                got = eng.W(mod, k).cpu().numpy()
                assert_close(f"s{step}.param.{mod}.{k}", got, p.detach().numpy(), 2e-5, 1e-5)
                assert_close(f"s{step}.param.{mod}.{k} (golden)", got, g[f"step{step}.param.{mod}.{k}"], 2e-5, 1e-5)
                i += 1


def test_planet_agent_loop_surface():
    """Replay fill -> Planet.train_step (the reference's four log keys) -> planning step in the collect loop."""
    d = synth.SMALL
    agent, P, env = _agent(d, 2, cls="planet", extra=("MPC.candidates=200", "MPC.top_candidates=20",
                                                      "MPC.optimisation_iters=3"))
    np.random.seed(0)
    agent.randomly_initialize_replay_buffer()
    logs = agent.train_step()
    assert set(logs) == {"observation_loss", "reward_loss", "kl_loss", "model_loss"}
    assert all(np.isfinite(v) for v in logs.values())
    obs = env.reset()
    belief, state, action = torch.zeros(1, d.Be).cuda(), torch.zeros(1, d.S).cuda(), torch.zeros(1, d.A).cuda()
    belief, state, action, nobs, reward, done = agent.update_belief_and_act(env, belief, state, action, obs, explore=True)
    assert belief.shape == (1, d.Be) and state.shape == (1, d.S) and action.shape == (1, d.A)
    assert float(action.abs().max()) <= 1.0 and np.isfinite(reward)
    with pytest.raises(NotImplementedError):
        agent.update_critic()
