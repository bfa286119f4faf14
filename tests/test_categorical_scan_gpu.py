"""GPU: latent_distribution="Categorical" end to end (BASELINE configs[4] latents: D one-hot factors of C classes) -- the
fused Categorical scans (csrc/scan_cat.hip) and the engine around them against the CPU oracle AND against golden vectors
produced by the reference's own Dreamer code under the two repairs it needs at HEAD (oracle/gen_golden.py,
CategoricalShims; tests/golden/cat_*.npz).  Sampled one-hot states must be exact; fp32 tolerances as in test_hip_parity."""
import numpy as np
import pytest
import torch

from big_dreamer_amd import synth
from tests.helpers import CAT_CASES, assert_close, check_fingerprints, compare_tensor, load_golden

pytestmark = pytest.mark.gpu


def _dev(dct):
    return {k: torch.as_tensor(v).cuda().contiguous() for k, v in dct.items()}


def _rel(name, got, want, atol, rtol, report):
    got = np.asarray(got, dtype=np.float64).reshape(np.asarray(want).shape)
    want = np.asarray(want, dtype=np.float64)
    err = np.abs(got - want)
    report.append(f"{name:28s} max|err|={err.max():.3e}  max|ref|={np.abs(want).max():.3e}")
    assert_close(name, got, want, atol, rtol)


def _setup(name, cluster=True):
    from big_dreamer_amd.engine import DreamerEngine
    from oracle import dreamer_oracle as O
    d, seed, hp, full = CAT_CASES[name]
    g = load_golden(name)
    P, batch, noise = synth.make_params(d, seed), synth.make_batch(d, seed), synth.make_noise(d, seed)
    check_fingerprints(g, P, batch, noise)
    eng = DreamerEngine(d, hp, "cuda", params=P)
    # 32 x 32 latents run the multi-CU cluster observe scan by default (csrc/observe_cat_cluster.hip); cluster=False
    # selects the one-workgroup-per-tile kernels (csrc/scan_cat.hip).  Ragged factor shapes (3 x 5) always use the latter.
    if not cluster:
        eng.use_obs_cluster = False
    assert bool(eng._cat_cluster(d.B)) == (cluster and d.cat_D == 32), "observe-scan kernel selection"
    od = O.OracleDreamer(P, dict(hp, planning_horizon=d.H, categorical=(d.cat_D, d.cat_C)))
    return d, seed, hp, full, g, P, batch, noise, eng, od


@pytest.mark.parametrize("name,cluster", [(n, True) for n in CAT_CASES] + [("cat_32", False), ("cat_pixel_32", False)])
def test_categorical_forward_pieces(name, cluster):
    """Observe scan (posterior logits, exact one-hot samples, beliefs), batched prior head, imagination rollout
    (beliefs, exact sampled states, prior logits, entropy), reward / value heads on [h; one-hot s], lambda-returns."""
    from big_dreamer_amd import _cabi as cabi
    from oracle import dreamer_oracle as O
    d, seed, hp, full, g, P, batch, noise, eng, od = _setup(name, cluster)
    cat = (d.cat_D, d.cat_C)
    tb = {k: torch.as_tensor(v) for k, v in batch.items()}
    tn = {k: torch.as_tensor(v) for k, v in noise.items()}
    with torch.no_grad():
        _, _, _, _, inter = od.world_model_forward(tb, tn)
        ib, is_, (il,), ent = O.imagine_ahead(od.P, inter["posterior_states"], inter["beliefs"], d.H, tn["action"],
                                              tn["entropy"], tn["img_prior"], cat)
        r = O.dense_on_features(ib, is_, od.P["reward_model"])
        v = O.dense_on_features(ib, is_, od.P["critic_target"])
        ret = O.lambda_return(r, v, v[-1], od.hp["discount"], od.hp["disclam"])
    db, dn = _dev(batch), _dev(noise)
    T, B, N, Hm, F = d.T, d.B, d.N, d.Hm, d.Be + d.S
    if d.pixel:     # configs[4] as stated: 64x64 pixel observations + Categorical latents
        emb, pre = eng.encode_pixels(db["observations"][1:].reshape(N, 3, 64, 64))
    else:
        emb, pre = eng.encode(db["observations"][1:].reshape(N, d.O), N)
    feat, ql, _ = eng.observe(db["actions"][:-1], db["nonterminals"][:-1], pre, dn["obs_post"],
                              torch.zeros(B, d.Be, device="cuda"), torch.zeros(B, d.S, device="cuda"), T, B)
    pst, pl, _ = eng.prior_head(feat, N, dn["obs_prior"].reshape(N, d.S))
    ifeat, e_ent, act = eng.imagine(feat, N, Hm, dn, start_sidx=eng._buf["sidx"])
    r_out, _, _ = eng.dense_forward("reward_model", "rew", "ir", ifeat, F, Hm * N, 1)
    v_out, _, _ = eng.dense_forward("critic_target", "tgt", "iv", ifeat, F, Hm * N, 1)
    returns = torch.zeros(Hm * N, device="cuda")
    cabi.check(cabi.lib.bd_lambda_return_forward(r_out.data_ptr(), v_out.data_ptr(), Hm, N, eng.hp["discount"],
                                                 eng.hp["disclam"], returns.data_ptr(), cabi.stream()))
    torch.cuda.synchronize()
    rep = []
    c = lambda t: t.detach().cpu().numpy()
    try:
        f = c(feat).reshape(T, B, F)
        _rel("beliefs", f[..., :d.Be], inter["beliefs"], 2e-5, 2e-5, rep)
        assert np.array_equal(f[..., d.Be:], inter["posterior_states"].numpy()), "posterior one-hot samples differ"
        idx = c(eng._buf["sidx"]).reshape(T, B, d.cat_D)
        assert np.array_equal(idx, inter["posterior_states"].numpy().reshape(T, B, d.cat_D, d.cat_C).argmax(-1))
        _rel("posterior_logits", c(ql), inter["posterior_logits"].reshape(N, -1), 2e-5, 2e-5, rep)
        _rel("prior_logits", c(pl), inter["prior_logits"].reshape(N, -1), 2e-5, 2e-5, rep)
        assert np.array_equal(c(pst).reshape(T, B, -1), inter["prior_states"].numpy()), "prior one-hot samples differ"
        compare_tensor(g, "piece.posterior_logits", c(ql).reshape(T, B, d.cat_D, d.cat_C), full, 2e-5, 2e-5)
        compare_tensor(g, "piece.posterior_states", f[..., d.Be:], full, 0.0, 0.0)
        fi = c(ifeat).reshape(Hm, N, F)
        _rel("imged_beliefs", fi[..., :d.Be], ib, 5e-5, 5e-5, rep)
        assert np.array_equal(fi[..., d.Be:], is_.numpy()), "imagined one-hot samples differ"
        _rel("imged_prior_logits", c(eng._buf["iprior_logits"]), il.reshape(Hm * N, -1), 5e-5, 5e-5, rep)
        _rel("action_entropy", c(e_ent).reshape(Hm, N), ent, 2e-2, 1e-3, rep)
        # mean: tight; with a dozen rows ONE ill-conditioned row at its own tolerance (2e-2) may move it by 2e-2 / rows
        assert abs(float(c(e_ent).mean()) - float(ent.mean())) < max(2e-4, 2e-2 / (Hm * N))
        _rel("imged_reward", c(r_out).reshape(Hm, N, 1), r, 5e-5, 5e-5, rep)
        _rel("value_pred", c(v_out).reshape(Hm, N, 1), v, 5e-5, 5e-5, rep)
        _rel("returns", c(returns).reshape(Hm, N, 1), ret, 2e-4, 5e-5, rep)
        compare_tensor(g, "piece.imged_states", fi[..., d.Be:], full, 0.0, 0.0)
        compare_tensor(g, "piece.returns", c(returns).reshape(Hm, N, 1), full, 2e-4, 5e-5)
    finally:
        print("\n".join(rep))


@pytest.mark.parametrize("name,cluster", [(n, True) for n in CAT_CASES] + [("cat_32", False), ("cat_32_v2", False),
                                                                          ("cat_pixel_32", False)])
def test_categorical_train_steps_vs_oracle_and_golden(name, cluster):
    """Two whole train steps with Categorical latents: logs, clipped gradients of every parameter tensor, gradient norms,
    post-Adam weights -- against the oracle and against the reference's own run (golden)."""
    from oracle import dreamer_oracle as O
    d, seed, hp, full, g, P, batch, noise, eng, od = _setup(name, cluster)
    db = _dev(batch)
    rep = []
    try:
        for step in range(2):
            nz = synth.make_noise(d, seed + step)
            ologs = od.train_step(batch, nz)
            logs = eng.train_step(db, _dev(nz))
            if step == 0:
                od.update_critic()
                eng.update_critic()
            torch.cuda.synchronize()
            eng.cluster_status(d.B)          # no cluster member timed out waiting for its peers
            for k, v in ologs.items():
                tol = (2e-4, 2e-4) if k in ("policy_entropy", "actor_loss") else (2e-5, 5e-5)
                _rel(f"s{step}.{k}", logs[k], v, tol[0], tol[1], rep)
                _rel(f"s{step}.{k}(golden)", logs[k], g[f"step{step}.log.{k}"], tol[0], tol[1], rep)
            gn = od.last["grad_norms"]
            _rel(f"s{step}.grad_norms", [logs["grad_norm_model"], logs["grad_norm_actor"], logs["grad_norm_critic"]],
                 [gn["model"], gn["actor"], gn["critic"]], 1e-6, 1e-3, rep)
            coef = {k: min(1.0, od.hp["grad_clip_norm"] / (gn[k] + 1e-6)) for k in gn}
            groups = {"model": (O.MODEL_MODULES, od.last["model_grads"]), "actor": (("actor",), od.last["actor_grads"]),
                      "critic": (("critic",), od.last["critic_grads"])}
            for grp, (mods, grads) in groups.items():
                i = 0
                for mod in mods:
                    for k in od.P[mod]:
                        want = grads[i].numpy() * coef[grp]
                        got = eng.G(mod, k).detach().cpu().numpy()
                        scale = float(np.abs(want).max()) + 1e-12
                        _rel(f"s{step}.grad.{mod}.{k}", got, want, 2e-3 * scale + 1e-9, 2e-3, rep)
                        i += 1
            for mod in list(O.MODEL_MODULES) + ["actor", "critic", "critic_target"]:
                for k, p in od.P[mod].items():
                    got = eng.W(mod, k).detach().cpu().numpy()
                    _rel(f"s{step}.param.{mod}.{k}", got, p.detach().numpy(), 2e-5, 1e-5, rep)
                    compare_tensor(g, f"step{step}.param.{mod}.{k}", got, full, atol=2e-5, rtol=1e-5)
    finally:
        print("\n".join(rep[-400:]))


def test_categorical_surface_matches_oracle():
    """The drop-in surface with latent_distribution=Categorical: Dreamer(params, env) builds (state_size = D*C,
    src/planet.py:56-57), TransitionModel.forward returns 1-tuples of (T, B, D, C) logits, imagine_ahead / get_action /
    update_belief_and_act run on the Categorical kernels, train_step returns the reference's log keys."""
    from big_dreamer_amd.config import load_config
    from big_dreamer_amd.dreamer import DreamerV2
    from oracle import dreamer_oracle as O
    d = synth.CAT_32
    seed = 61

    class Env:
        action_size, observation_size = d.A, d.O

        def step(self, a):
            return torch.zeros(1, d.O), 0.0, False

    params = load_config([f"belief_size={d.Be}", "state_size=30", f"hidden_size={d.Hd}", f"embedding_size={d.E}",
                          f"batch_size={d.B}", f"seq_len={d.L}", f"planning_horizon={d.H}", "experience_size=200",
                          "algorithm=dreamerV2", "latent_distribution=Categorical", f"discrete_latent_dimensions={d.cat_D}",
                          f"discrete_latent_classes={d.cat_C}"])
    agent = DreamerV2(params, Env())
    assert agent.state_size == d.S and agent.transition_model.latent_distribution == "Categorical"
    P = synth.make_params(d, seed)
    for mod in ("transition_model", "observation_model", "reward_model", "encoder", "actor", "critic", "critic_target"):
        m = getattr(agent, mod)
        assert list(m.state_dict().keys()) == [n for n, _ in synth.param_shapes(d)[mod if mod != "critic_target" else "critic"]]
        m.load_state_dict({k: torch.from_numpy(v) for k, v in P[mod].items()})
    tP = {m: {k: torch.tensor(v) for k, v in sd.items()} for m, sd in P.items()}
    batch, noise = synth.make_batch(d, seed), synth.make_noise(d, seed)
    tb = {k: torch.tensor(v) for k, v in batch.items()}
    cu = lambda x: torch.as_tensor(x).cuda()
    cat = (d.cat_D, d.cat_C)
    emb = agent.encoder(cu(batch["observations"][1:]))
    want_emb = O.mlp(tb["observations"][1:], tP["encoder"])
    out = agent.transition_model(torch.zeros(d.B, d.S).cuda(), cu(batch["actions"][:-1]), torch.zeros(d.B, d.Be).cuda(), emb,
                                 cu(batch["nonterminals"][:-1]), _noise=(cu(noise["obs_prior"]), cu(noise["obs_post"])))
    want = O.transition_forward(tP["transition_model"], torch.zeros(d.B, d.S), tb["actions"][:-1], torch.zeros(d.B, d.Be),
                                want_emb, tb["nonterminals"][:-1], torch.tensor(noise["obs_prior"]),
                                torch.tensor(noise["obs_post"]), cat)
    assert len(out[2]) == 1 and len(out[4]) == 1 and tuple(out[2][0].shape) == (d.T, d.B, d.cat_D, d.cat_C)
    assert_close("beliefs", out[0].cpu().numpy(), want[0].numpy(), 2e-5, 2e-5)
    assert np.array_equal(out[1].cpu().numpy(), want[1].numpy()), "prior states"
    assert np.array_equal(out[3].cpu().numpy(), want[3].numpy()), "posterior states"
    assert_close("prior logits", out[2][0].cpu().numpy(), want[2][0].numpy(), 2e-5, 2e-5)
    assert_close("posterior logits", out[4][0].cpu().numpy(), want[4][0].numpy(), 2e-5, 2e-5)
    # prior-only rollout (embeddings=None): the sampled prior state is fed back
    pout = agent.transition_model(torch.zeros(d.B, d.S).cuda(), cu(batch["actions"][:-1]), torch.zeros(d.B, d.Be).cuda(), None,
                                  None, _noise=(cu(noise["obs_prior"]), None))
    pwant = O.transition_forward(tP["transition_model"], torch.zeros(d.B, d.S), tb["actions"][:-1], torch.zeros(d.B, d.Be),
                                 None, None, torch.tensor(noise["obs_prior"]), None, cat)
    assert pout[3] is None and pout[4] is None
    assert np.array_equal(pout[1].cpu().numpy(), pwant[1].numpy()) and len(pout[2]) == 1
    assert_close("prior-only beliefs", pout[0].cpu().numpy(), pwant[0].numpy(), 2e-5, 2e-5)
    # imagine_ahead from the posteriors
    nz = {"action": cu(noise["action"]), "entropy": cu(noise["entropy"]), "img_prior": cu(noise["img_prior"])}
    ib, is_, (il,), ent = agent.imagine_ahead(out[3], out[0], _noise=nz)
    wb, ws, (wl,), went = O.imagine_ahead(tP, want[3], want[0], d.H, torch.tensor(noise["action"]),
                                          torch.tensor(noise["entropy"]), torch.tensor(noise["img_prior"]), cat)
    assert tuple(il.shape) == (d.Hm, d.N, d.cat_D, d.cat_C)
    assert_close("imagine beliefs", ib.cpu().numpy(), wb.numpy(), 5e-5, 5e-5)
    assert np.array_equal(is_.cpu().numpy(), ws.numpy())
    assert_close("entropy", ent.cpu().numpy(), went.numpy(), 2e-2, 1e-3)
    # an all-zero start state through the public surface (the collect loop's initial state, src/main.py:91-95): the actor
    # and the embed layer must see zeros, as in the reference -- not "class 0 of every factor" 
    nz0 = {"action": cu(noise["action"][:, :d.B].copy()), "entropy": cu(noise["entropy"][:, :, :d.B].copy()),
           "img_prior": cu(noise["img_prior"][:, :d.B].copy())}
    zb, zs, (zl,), zent = agent.imagine_ahead(torch.zeros(1, d.B, d.S).cuda(), out[0][:1].contiguous(), _noise=nz0)
    wzb, wzs, (wzl,), wzent = O.imagine_ahead(tP, torch.zeros(1, d.B, d.S), want[0][:1], d.H,
                                              torch.tensor(noise["action"][:, :d.B].copy()),
                                              torch.tensor(noise["entropy"][:, :, :d.B].copy()),
                                              torch.tensor(noise["img_prior"][:, :d.B].copy()), cat)
    assert_close("zero-start beliefs", zb.cpu().numpy(), wzb.numpy(), 5e-5, 5e-5)
    assert np.array_equal(zs.cpu().numpy(), wzs.numpy()), "zero-start sampled states"
    assert_close("zero-start entropy", zent.cpu().numpy(), wzent.numpy(), 2e-2, 1e-3)
    with pytest.raises(ValueError, match="one-hot per factor"):
        agent.imagine_ahead(torch.full((1, d.B, d.S), 0.5).cuda(), out[0][:1].contiguous(), _noise=nz0)
    # one collect-loop decision from the all-zero start state (src/main.py:91-95), then one from the sampled state
    belief, state = torch.zeros(1, d.Be).cuda(), torch.zeros(1, agent.state_size).cuda()
    action = torch.zeros(1, d.A).cuda()
    for _ in range(2):
        belief, state, action, _, _, _ = agent.update_belief_and_act(Env(), belief, state, action, torch.zeros(1, d.O),
                                                                    explore=True)
        assert tuple(state.shape) == (1, d.S) and float(state.sum()) == d.cat_D and float(action.abs().max()) <= 1.0
    # train_step through the surface
    rep = synth.make_replay(d, rows=200, seed=2)
    for k, v in rep.items():
        getattr(agent.buffer, k)[:] = v
    agent.buffer.idx, agent.buffer.full = 0, True
    np.random.seed(0)
    logs = agent.train_step()
    assert set(logs.keys()) == {"observation_loss", "reward_loss", "kl_loss", "model_loss", "actor_loss", "policy_entropy",
                                "value_loss"}
    assert all(np.isfinite(float(v)) for v in logs.values())


@pytest.mark.parametrize("which", ["state", "pixel"])
def test_categorical_full_size_step_vs_oracle(which):
    """BASELINE configs[4] per GPU at full size: 32 x 32 latents, batch 100 (= 800 / 8), chunk 50, H 15, belief / hidden
    200, embedding 1024 -- "state": on state observations (A = 1); "pixel": AS STATED, 64x64 pixel observations, A = 17
    (the decoder reads [h; one-hot s], K = 1224; src/models.py:319-362, src/planet.py:56-57).  One whole train step
    against the oracle (no reference run at this size; the path is pinned at small sizes by tests/golden/cat_*.npz):
    losses, gradient norms, the clipped gradient of EVERY parameter tensor, every weight after Adam."""
    from big_dreamer_amd.engine import DreamerEngine
    from oracle import dreamer_oracle as O
    d = synth.CONFIG5 if which == "pixel" else synth.CONFIG5_STATE
    P, batch, nz = synth.make_params(d, 71), synth.make_batch(d, 71), synth.make_noise(d, 71)
    torch.set_num_threads(16)
    od = O.OracleDreamer(P, dict(planning_horizon=d.H, categorical=(d.cat_D, d.cat_C), free_nats=0.0))
    eng = DreamerEngine(d, dict(free_nats=0.0), "cuda", params=P)
    ologs = od.train_step(batch, nz)
    logs = eng.train_step(_dev(batch), _dev(nz))
    torch.cuda.synchronize()
    assert eng._cat_cluster(d.B) == 16, "batch 100 = 7 row tiles x 16 members: the cluster observe scan is the product path"
    eng.cluster_status(d.B)
    rep = []
    try:
        for k, v in ologs.items():
            tol = (5e-4, 5e-4) if k in ("policy_entropy", "actor_loss") else (2e-5, 5e-5)
            _rel(k, logs[k], v, tol[0], tol[1], rep)
        gn = od.last["grad_norms"]
        _rel("grad_norms", [logs["grad_norm_model"], logs["grad_norm_actor"], logs["grad_norm_critic"]],
             [gn["model"], gn["actor"], gn["critic"]], 1e-6, 2e-3, rep)
        # per-tensor clipped gradients: after ONE Adam step a weight moves by ~lr * sign(g), so the weights alone do
        # not check gradient magnitudes.  A handful of the 2.3 M sampled factors are near-ties that resolve differently
        # between fp32 summation orders (the small cases compare samples exactly): tolerance relative to the tensor's
        # own scale, as in the small cases.
        coef = {k: min(1.0, od.hp["grad_clip_norm"] / (gn[k] + 1e-6)) for k in gn}
        groups = {"model": (O.MODEL_MODULES, od.last["model_grads"]), "actor": (("actor",), od.last["actor_grads"]),
                  "critic": (("critic",), od.last["critic_grads"])}
        for grp, (mods, grads) in groups.items():
            i = 0
            for mod in mods:
                for k in od.P[mod]:
                    want = grads[i].numpy() * coef[grp]
                    scale = float(np.abs(want).max()) + 1e-12
                    _rel(f"grad.{mod}.{k}", eng.G(mod, k).detach().cpu().numpy(), want, 2e-3 * scale + 1e-9, 2e-3, rep)
                    i += 1
        for mod in list(O.MODEL_MODULES) + ["actor", "critic"]:
            for k, p in od.P[mod].items():
                _rel(f"param.{mod}.{k}", eng.W(mod, k).cpu().numpy(), p.detach().numpy(), 2e-5, 1e-5, rep)
    finally:
        print("\n".join(rep[-160:]))


def test_categorical_cluster_scan_matches_one_workgroup_scan():
    """The multi-CU cluster observe scan (csrc/observe_cat_cluster.hip: GRU column blocks and factor groups of the head
    split over 16 members per tile, two hand-offs per step) against the one-workgroup-per-tile scan (csrc/scan_cat.hip) at
    the full model size (belief / hidden 200, 32 x 32 latents), batch 40 = three row tiles, one ragged: identical sampled
    states, beliefs / logits / every weight after a whole train step to fp32 summation-order noise."""
    from big_dreamer_amd.engine import DreamerEngine
    d = synth.Dims(B=40, L=9, H=4, S=1024, cat_D=32, cat_C=32, A=3, O=7)
    P, batch, nz = synth.make_params(d, 81), synth.make_batch(d, 81), synth.make_noise(d, 81)
    out = {}
    for cluster in (True, False):
        eng = DreamerEngine(d, dict(free_nats=0.0), "cuda", params=P)
        eng.use_obs_cluster = cluster
        assert bool(eng._cat_cluster(d.B)) == cluster
        logs = eng.train_step(_dev(batch), _dev(nz))
        torch.cuda.synchronize()
        eng.cluster_status(d.B)
        out[cluster] = (logs, {k: eng._buf[k].clone() for k in ("p0_feat", "post_logits", "p0_sidx", "d_embed_pre", "d_q1_pre", "d_q2_out")},
                        {g: eng.groups[g].flat.clone() for g in ("model", "actor", "critic")})
    (la, ba, wa), (lb, bb, wb) = out[True], out[False]
    assert torch.equal(ba["p0_sidx"], bb["p0_sidx"]), "sampled posterior classes differ between the two scans"
    for k in ("p0_feat", "post_logits", "d_embed_pre", "d_q1_pre", "d_q2_out"):
        scale = float(bb[k].abs().max()) + 1e-12
        assert float((ba[k] - bb[k]).abs().max()) <= 2e-5 * scale + 1e-7, (k, float((ba[k] - bb[k]).abs().max()), scale)
    for g in wa:
        assert float((wa[g] - wb[g]).abs().max()) < 2e-6, g
    for k in la:
        assert abs(la[k] - lb[k]) <= 1e-5 + 1e-5 * abs(lb[k]), (k, la[k], lb[k])
