"""GPU: the reference-compatible Python surface (Dreamer / TransitionModel / lambda_return / CLI) against the
oracle, through the same HIP kernels as the engine tests."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from big_dreamer_amd import synth
from tests.helpers import assert_close

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _agent(d, seed=1):
    from big_dreamer_amd.config import load_config
    from big_dreamer_amd.dreamer import Dreamer
    from big_dreamer_amd.env import SyntheticEnv
    params = load_config([f"belief_size={d.Be}", f"state_size={d.S}", f"hidden_size={d.Hd}", f"embedding_size={d.E}",
                          f"batch_size={d.B}", f"seq_len={d.L}", f"planning_horizon={d.H}", "experience_size=400",
                          "seed_steps=120", "max_episode_length=40"])
    env = SyntheticEnv(d.O, d.A, 40, 2, 0)
    agent = Dreamer(params, env)
    P = synth.make_params(d, seed)
    for mod in ("transition_model", "observation_model", "reward_model", "encoder", "actor", "critic", "critic_target"):
        m = getattr(agent, mod)
        assert list(m.state_dict().keys()) == [n for n, _ in synth.param_shapes(d)[mod if mod != "critic_target" else "critic"]]
        m.load_state_dict({k: torch.from_numpy(v) for k, v in P[mod].items()})     # strict: names + shapes
    return agent, P, env


def test_state_dict_roundtrip_and_module_forward_match_oracle():
    from oracle import dreamer_oracle as O
    d = synth.SMALL
    agent, P, env = _agent(d)
    tP = {m: {k: torch.tensor(v) for k, v in sd.items()} for m, sd in P.items()}
    batch, noise = synth.make_batch(d, 1), synth.make_noise(d, 1)
    tb = {k: torch.tensor(v) for k, v in batch.items()}
    cu = lambda x: torch.as_tensor(x).cuda()
    # encoder + TransitionModel.forward (5-tuple, reference shapes)
    emb = agent.encoder(cu(batch["observations"][1:]))
    want_emb = O.mlp(tb["observations"][1:], tP["encoder"])
    assert_close("encoder", emb.cpu().numpy(), want_emb.numpy(), 2e-5, 2e-5)
    out = agent.transition_model(torch.zeros(d.B, d.S).cuda(), cu(batch["actions"][:-1]), torch.zeros(d.B, d.Be).cuda(),
                                 emb, cu(batch["nonterminals"][:-1]), _noise=(cu(noise["obs_prior"]), cu(noise["obs_post"])))
    want = O.transition_forward(tP["transition_model"], torch.zeros(d.B, d.S), tb["actions"][:-1], torch.zeros(d.B, d.Be),
                                want_emb, tb["nonterminals"][:-1], torch.tensor(noise["obs_prior"]),
                                torch.tensor(noise["obs_post"]))
    flat = lambda o: [o[0], o[1], o[2][0], o[2][1], o[3], o[4][0], o[4][1]]
    for name, g, w in zip(["beliefs", "prior_states", "prior_means", "prior_stds", "post_states", "post_means", "post_stds"],
                          flat(out), flat(want)):
        assert tuple(g.shape) == tuple(w.shape)
        assert_close(name, g.cpu().numpy(), w.numpy(), 2e-5, 2e-5)
    # prior-only mode (embeddings=None): what the MPC planner calls (src/planner.py:65)
    pout = agent.transition_model(torch.zeros(d.B, d.S).cuda(), cu(batch["actions"][:-1]), torch.zeros(d.B, d.Be).cuda(),
                                  None, None, _noise=(cu(noise["obs_prior"]), None))
    pwant = O.transition_forward(tP["transition_model"], torch.zeros(d.B, d.S), tb["actions"][:-1], torch.zeros(d.B, d.Be),
                                 None, None, torch.tensor(noise["obs_prior"]), None)
    assert pout[3] is None and pout[4] is None
    for name, g_, w_ in zip(["prior-only beliefs", "prior-only states", "prior-only means", "prior-only stds"],
                            [pout[0], pout[1], pout[2][0], pout[2][1]], [pwant[0], pwant[1], pwant[2][0], pwant[2][1]]):
        assert_close(name, g_.cpu().numpy(), w_.numpy(), 2e-5, 2e-5)
    # imagine_ahead / get_action / heads / lambda_return
    from big_dreamer_amd.dreamer import lambda_return
    nz = {"action": cu(noise["action"]), "entropy": cu(noise["entropy"]), "img_prior": cu(noise["img_prior"])}
    ib, is_, (im, isd), ent = agent.imagine_ahead(out[3], out[0], _noise=nz)
    wb, ws, (wm, wsd), went = O.imagine_ahead(tP, want[3], want[0], d.H, torch.tensor(noise["action"]),
                                              torch.tensor(noise["entropy"]), torch.tensor(noise["img_prior"]))
    assert_close("imagine beliefs", ib.cpu().numpy(), wb.numpy(), 5e-5, 5e-5)
    assert_close("imagine states", is_.cpu().numpy(), ws.numpy(), 5e-5, 5e-5)
    assert_close("imagine prior std", isd.cpu().numpy(), wsd.numpy(), 5e-5, 5e-5)
    assert_close("entropy", ent.cpu().numpy(), went.numpy(), 2e-2, 1e-3)
    r = agent.reward_model(ib, is_)
    v = agent.critic_target(ib, is_)
    assert tuple(r.shape) == (d.Hm, d.N, 1)
    ret = lambda_return(r, v, bootstrap=v[-1], discount=0.995, lambda_=0.95)
    wret = O.lambda_return(O.dense_on_features(wb, ws, tP["reward_model"]), O.dense_on_features(wb, ws, tP["critic_target"]),
                           O.dense_on_features(wb, ws, tP["critic_target"])[-1], 0.995, 0.95)
    assert_close("lambda_return", ret.cpu().numpy(), wret.numpy(), 2e-4, 5e-5)
    act, ent1 = agent.get_action(out[0][0], out[3][0], _noise={"action": nz["action"][:1, :d.B], "entropy":
                                                               nz["entropy"][:1, :, :d.B].contiguous(),
                                                               "img_prior": nz["img_prior"][:1, :d.B]})
    wact, went1 = O.get_action(want[0][0], want[3][0], tP["actor"], torch.tensor(noise["action"][0, :d.B]),
                               torch.tensor(noise["entropy"][0, :, :d.B]))
    assert_close("get_action", act.cpu().numpy(), wact.numpy(), 2e-5, 2e-5)
    assert_close("get_action entropy", ent1.cpu().numpy(), went1.numpy(), 2e-2, 1e-3)


def test_agent_loop_surface():
    """Replay fill -> train_step (reference log keys) -> update_critic -> update_belief_and_act."""
    d = synth.SMALL
    agent, P, env = _agent(d)
    np.random.seed(0)
    steps, episodes = agent.randomly_initialize_replay_buffer()
    assert steps >= 120 and episodes >= 1
    logs = agent.train_step()
    assert set(logs) == {"observation_loss", "reward_loss", "kl_loss", "model_loss", "actor_loss", "policy_entropy",
                         "value_loss"}
    assert all(np.isfinite(v) for v in logs.values())
    agent.update_critic()
    obs = env.reset()
    belief = torch.zeros(1, d.Be).cuda()
    state = torch.zeros(1, d.S).cuda()
    action = torch.zeros(1, d.A).cuda()
    belief, state, action, nobs, reward, done = agent.update_belief_and_act(env, belief, state, action, obs, explore=True)
    assert belief.shape == (1, d.Be) and state.shape == (1, d.S) and action.shape == (1, d.A)
    assert float(action.abs().max()) <= 1.0 and np.isfinite(reward)
    # on a FIXED batch and FIXED noise the world-model and value losses must fall (sanity of the update path)
    agent.engine.hp.update(model_learning_rate=1e-3, value_learning_rate=1e-3)
    o, a, r, n = agent.buffer.sample(d.B, d.L)
    batch = {"observations": o, "actions": a, "rewards": r, "nonterminals": n}
    noise = {k: torch.as_tensor(v).cuda() for k, v in synth.make_noise(d, 5).items()}
    first = agent.engine.train_step(batch, noise)
    for _ in range(40):
        last = agent.engine.train_step(batch, noise)
    assert last["model_loss"] < first["model_loss"] - 1e-3, (first, last)


def test_cli_runs_like_the_reference():
    """python src/main.py key=value ... (reference README.md:19-28), tiny sizes, a handful of updates."""
    cmd = [sys.executable, os.path.join(ROOT, "src", "main.py"), "belief_size=32", "hidden_size=32", "embedding_size=64",
           "state_size=8", "batch_size=6", "seq_len=8", "planning_horizon=5", "experience_size=500", "seed_steps=100",
           "max_episode_length=30", "train_steps=140", "log_freq=10", "collect_interval=2"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "Initialized with" in out.stdout and "model_loss" in out.stdout


def test_pixel_agent_surface():
    """pixel_observation=True: uint8 replay + HIP dequantise, conv encoder/decoder modules with the reference's
    state_dict names, train_step and update_belief_and_act."""
    from big_dreamer_amd.config import load_config
    from big_dreamer_amd.dreamer import Dreamer
    from big_dreamer_amd.env import SyntheticPixelEnv
    d = synth.TINY_PIXEL
    params = load_config([f"belief_size={d.Be}", f"state_size={d.S}", f"hidden_size={d.Hd}", "embedding_size=1024",
                          "batch_size=2", "seq_len=4", "planning_horizon=3", "experience_size=200", "seed_steps=60",
                          "max_episode_length=20", "pixel_observation=true"])
    env = SyntheticPixelEnv(3, d.A, 20, 2, 0)
    agent = Dreamer(params, env)
    P = synth.make_params(d, 4)
    for mod in ("transition_model", "observation_model", "reward_model", "encoder", "actor", "critic"):
        m = getattr(agent, mod)
        assert list(m.state_dict().keys()) == [n for n, _ in synth.param_shapes(d)[mod]]
        m.load_state_dict({k: torch.from_numpy(v) for k, v in P[mod].items()})
    np.random.seed(0)
    agent.randomly_initialize_replay_buffer()
    assert agent.buffer.observations.dtype == np.uint8
    logs = agent.train_step()
    assert all(np.isfinite(v) for v in logs.values()) and logs["observation_loss"] > 1000.0   # 12288 dims * ~0.92
    obs = env.reset()
    out = agent.update_belief_and_act(env, torch.zeros(1, d.Be).cuda(), torch.zeros(1, d.S).cuda(),
                                      torch.zeros(1, d.A).cuda(), obs, explore=True)
    assert out[0].shape == (1, d.Be) and out[2].shape == (1, d.A)
    img = agent.observation_model(out[0], out[1])
    assert tuple(img.shape) == (1, 3, 64, 64)


def test_actor_forward_and_deterministic_get_action_match_reference_golden():
    """ActorModel.forward -> (mean, std) and Dreamer.get_action(deterministic=True) (SampleDist.mode, then the entropy
    estimate) on the reference's draws; golden from the reference itself (tests/golden/action_mode.npz)."""
    from tests.helpers import load_golden
    d, seed = synth.SMALL, 12
    g = load_golden("action_mode")
    agent, P, env = _agent(d, seed)
    N = g["belief"].shape[0]
    ns = synth.NoiseStream(seed)
    eps_mode, eps_ent = ns.normal((d.n_entropy, N, d.A)), ns.normal((d.n_entropy, N, d.A))
    b, s = torch.from_numpy(g["belief"]).cuda(), torch.from_numpy(g["state"]).cuda()
    mean, std = agent.actor(b, s)
    assert_close("mean", mean.cpu().numpy(), g["mean"], 2e-5, 2e-5)
    assert_close("std", std.cpu().numpy(), g["std"], 2e-5, 2e-5)
    act, ent = agent.get_action(b, s, deterministic=True,
                                _noise={"mode": torch.from_numpy(eps_mode), "entropy": torch.from_numpy(eps_ent)})
    assert_close("action", act.cpu().numpy(), g["action"], 2e-5, 2e-5)
    assert_close("entropy", ent.cpu().numpy(), g["entropy"], 2e-2, 1e-3)      # tolerance of the other entropy checks
    act2, ent2 = agent.get_action(b, s, deterministic=True)                     # device noise
    assert act2.shape == (N, d.A) and float(act2.abs().max()) <= 1.0 and torch.isfinite(ent2).all()
