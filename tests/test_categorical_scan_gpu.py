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


def _setup(name):
    from big_dreamer_amd.engine import DreamerEngine
    from oracle import dreamer_oracle as O
    d, seed, hp, full = CAT_CASES[name]
    g = load_golden(name)
    P, batch, noise = synth.make_params(d, seed), synth.make_batch(d, seed), synth.make_noise(d, seed)
    check_fingerprints(g, P, batch, noise)
    eng = DreamerEngine(d, hp, "cuda", params=P)
    od = O.OracleDreamer(P, dict(hp, planning_horizon=d.H, categorical=(d.cat_D, d.cat_C)))
    return d, seed, hp, full, g, P, batch, noise, eng, od


@pytest.mark.parametrize("name", list(CAT_CASES))
def test_categorical_forward_pieces(name):
    """Observe scan (posterior logits, exact one-hot samples, beliefs), batched prior head, imagination rollout
    (beliefs, exact sampled states, prior logits, entropy), reward / value heads on [h; one-hot s], lambda-returns."""
    from big_dreamer_amd import _cabi as cabi
    from oracle import dreamer_oracle as O
    d, seed, hp, full, g, P, batch, noise, eng, od = _setup(name)
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
        assert abs(float(c(e_ent).mean()) - float(ent.mean())) < 2e-4
        _rel("imged_reward", c(r_out).reshape(Hm, N, 1), r, 5e-5, 5e-5, rep)
        _rel("value_pred", c(v_out).reshape(Hm, N, 1), v, 5e-5, 5e-5, rep)
        _rel("returns", c(returns).reshape(Hm, N, 1), ret, 2e-4, 5e-5, rep)
        compare_tensor(g, "piece.imged_states", fi[..., d.Be:], full, 0.0, 0.0)
        compare_tensor(g, "piece.returns", c(returns).reshape(Hm, N, 1), full, 2e-4, 5e-5)
    finally:
        print("\n".join(rep))


@pytest.mark.parametrize("name", list(CAT_CASES))
def test_categorical_train_steps_vs_oracle_and_golden(name):
    """Two whole train steps with Categorical latents: logs, clipped gradients of every parameter tensor, gradient norms,
    post-Adam weights -- against the oracle and against the reference's own run (golden)."""
    from oracle import dreamer_oracle as O
    d, seed, hp, full, g, P, batch, noise, eng, od = _setup(name)
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
