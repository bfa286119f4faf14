"""GPU: Categorical latents on the HIP kernels against the reference's own outputs
(tests/golden/categorical.npz) and the oracle: CategoricalBeliefModel forward / backward and the Categorical KL."""
import numpy as np
import pytest
import torch

from big_dreamer_amd import synth
from tests.helpers import assert_close, compare_tensor, load_golden

pytestmark = pytest.mark.gpu


def _scaled(name, got, want, rel=2e-4):
    want = np.asarray(want)
    assert_close(name, got, want, rel * (float(np.abs(want).max()) + 1e-12) + 1e-9, rel)


@pytest.mark.parametrize("name", list(synth.CATEGORICAL_CASES))
def test_categorical_belief_model_vs_reference_golden(name):
    """forward: sampled one-hot states bit-exact, logits 2e-5; backward through the straight-through sample, the
    softmax and both dense layers (input, weight and bias gradients) relative to each tensor's scale."""
    from big_dreamer_amd.categorical import CategoricalBeliefModel
    rows, inp, hid, D, C, seed = synth.CATEGORICAL_CASES[name]
    g = load_golden("categorical")
    c = synth.make_categorical_case(rows, inp, hid, D, C, seed)
    m = CategoricalBeliefModel(inp, hid, D, C, "ELU")
    assert list(m.state_dict().keys()) == ["model.0.weight", "model.0.bias", "model.2.weight", "model.2.bias"]
    m.load_state_dict({k: torch.from_numpy(c[k]) for k in m.state_dict()})
    x = torch.from_numpy(c["x"]).cuda().requires_grad_(True)
    state, (logits,) = m(x, _noise=torch.from_numpy(g[f"{name}.q"]))
    assert tuple(state.shape) == (rows, D * C) and tuple(logits.shape) == (rows, D, C)
    assert np.array_equal(state.detach().cpu().numpy(), g[f"{name}.state"]), "sampled one-hot states differ"
    assert_close("logits", logits.detach().cpu().numpy(), g[f"{name}.logits"], 2e-5, 2e-5)
    ((state * torch.from_numpy(c["g_state"]).cuda()).sum() + (logits * torch.from_numpy(c["g_logits"]).cuda()).sum()).backward()
    torch.cuda.synchronize()
    _scaled("dx", x.grad.cpu().numpy(), g[f"{name}.dx"])
    for k, p in m.named_parameters():
        got = p.grad.cpu().numpy()
        if f"{name}.grad.{k}" in g:
            _scaled(f"grad.{k}", got, g[f"{name}.grad.{k}"])
        else:
            scale = float(np.abs(g[f"{name}.grad.{k}.sample"]).max())
            compare_tensor(g, f"{name}.grad.{k}", got, False, 2e-4 * scale, 2e-4)
    # leading dimensions are kept (time-major (T, B, .) inputs) and fresh device noise gives valid one-hots
    with torch.no_grad():
        s2, (l2,) = m(x.detach().reshape(1, rows, inp))
    assert tuple(s2.shape) == (1, rows, D * C) and tuple(l2.shape) == (1, rows, D, C)
    assert torch.equal(s2.reshape(rows, D, C).sum(-1), torch.ones(rows, D, device="cuda"))
    assert set(s2.unique().tolist()) <= {0.0, 1.0}


@pytest.mark.parametrize("name", list(synth.CATEGORICAL_CASES))
@pytest.mark.parametrize("tag,bal", [("bal_clamped", 0.8), ("bal_free", 0.8), ("sum_mixed", -1)])
def test_categorical_kl_vs_reference_golden(name, tag, bal):
    """Dreamer._kl_loss, Categorical branch: value and gradients w.r.t. posterior and prior logits."""
    from big_dreamer_amd.categorical import kl_loss_categorical
    rows, inp, hid, D, C, seed = synth.CATEGORICAL_CASES[name]
    g = load_golden("categorical")
    c = synth.make_categorical_case(rows, inp, hid, D, C, seed)
    ql = torch.from_numpy(g[f"{name}.logits"]).reshape(1, rows, D, C).cuda().requires_grad_(True)
    pl = torch.from_numpy(c["other_logits"]).reshape(1, rows, D, C).cuda().requires_grad_(True)
    kl = kl_loss_categorical(ql, pl, bal, float(g[f"{name}.kl.{tag}.free_nats"]))
    assert tuple(kl.shape) == (() if bal == -1 else (1,))      # the reference's shapes
    kl.sum().backward()
    torch.cuda.synchronize()
    assert_close("kl", kl.detach().cpu().numpy(), g[f"{name}.kl.{tag}"], 2e-6, 2e-5)
    want_q, want_p = g[f"{name}.kl.{tag}.dpost"], g[f"{name}.kl.{tag}.dprior"]
    if tag == "bal_clamped":
        assert not want_q.any() and not want_p.any()          # the clamp is active: the reference's gradients are zero
    _scaled("dpost", ql.grad.cpu().numpy(), want_q)
    _scaled("dprior", pl.grad.cpu().numpy(), want_p)

