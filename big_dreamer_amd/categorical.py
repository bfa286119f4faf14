"""Categorical (DreamerV2-style) latents on the HIP kernels -- SURVEY.md section 8a row R2c.

``CategoricalBeliefModel`` keeps the reference's constructor, ``state_dict`` names and return value
(src/models.py:76-117): ``forward(belief) -> (state, (logits,))`` with ``state`` the flattened straight-through
one-hot sample of D categorical factors with C classes.  ``kl_loss_categorical`` is the Categorical branch of
``Dreamer._kl_loss`` (src/dreamer.py:102-146).  Both are ``torch.autograd.Function``s over the C ABI: dense layers
on ``bd_mlp_forward/backward`` + ``bd_wgrad``, the softmax / sample / straight-through tail on
``bd_categorical_head_forward/backward``, the KL on ``bd_kl_categorical_forward/backward``.

What is NOT here: the Categorical variant of the fused RSSM scan (``TransitionModel(latent_distribution=
"Categorical")``).  The reference itself crashes on that path at HEAD (src/models.py:284), so only these pieces can
be pinned against it (tests/golden/categorical.npz); see DESIGN.md section 5.

Sampling noise: the reference reaches ``torch.multinomial(probs, 1, True)``, whose single-draw algorithm is
``q = empty_like(probs).exponential_(1); argmax(probs / q)``.  ``forward(..., _noise=q)`` takes those Exp(1) draws
(rows, D, C) explicitly; without it they are drawn on the device.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch
from torch import Tensor, nn

from . import _cabi as cabi

lib, ptr = cabi.lib, cabi.ptr


def _pack(w: Tensor, transpose: bool) -> Tensor:
    """Packed MFMA-fragment copy of a row-major (N, K) matrix (or of its transpose)."""
    N, K = w.shape
    dst = torch.zeros(cabi.packed_floats(N, K), dtype=torch.float32, device=w.device)
    d = (cabi.PackDesc * 1)(cabi.PackDesc(w.data_ptr(), dst.data_ptr(), w.stride(0), N, K, int(transpose)))
    raw = torch.frombuffer(bytearray(bytes(d)), dtype=torch.uint8).to(w.device)
    cabi.check(lib.bd_pack_weights(raw.data_ptr(), 1, cabi.stream()))
    raw.record_stream(torch.cuda.current_stream())
    return dst


def _wgrad(dpre: Tensor, act: Tensor, M: int, N: int, K: int):
    dW = torch.empty(N, K, dtype=torch.float32, device=dpre.device)
    db = torch.empty(N, dtype=torch.float32, device=dpre.device)
    ws = torch.zeros(max(1, int(lib.bd_wgrad_ws_floats(M, N, K))), dtype=torch.float32, device=dpre.device)
    cabi.check(lib.bd_wgrad(ptr(dpre), N, ptr(act), K, M, N, K, ptr(dW), K, ptr(db), 0, ptr(ws), ws.numel(), cabi.stream()))
    return dW, db


class _CategoricalBelief(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, W0, b0, W2, b2, q, D, Cc):
        M, K = x.shape
        Hd, N = W0.shape[0], W2.shape[0]
        dev = x.device
        hid = torch.empty(M, Hd, dtype=torch.float32, device=dev)
        logits = torch.empty(M, N, dtype=torch.float32, device=dev)
        pk0, pk2 = _pack(W0, False), _pack(W2, False)
        a = cabi.MlpFwdArgs()
        a.M, a.in0, a.ld0, a.w0 = M, ptr(x), K, K
        a.in1, a.ld1, a.w1 = None, 0, 0
        a.n_layers = 2
        a.layer[0] = cabi.Layer(ptr(pk0), ptr(b0), Hd, K, cabi.ACT_ELU, ptr(hid))
        a.layer[1] = cabi.Layer(ptr(pk2), ptr(b2), N, Hd, cabi.ACT_NONE, None)
        a.out, a.ldo = ptr(logits), N
        cabi.check(lib.bd_mlp_forward(C.byref(a), cabi.stream()))
        state = torch.empty(M, N, dtype=torch.float32, device=dev)
        probs = torch.empty(M, N, dtype=torch.float32, device=dev)
        cabi.check(lib.bd_categorical_head_forward(ptr(logits), ptr(q), M, D, Cc, ptr(state), ptr(probs), cabi.stream()))
        ctx.save_for_backward(x, W0, W2, hid, probs)
        ctx.dims = (M, K, Hd, N, D, Cc)
        return state, logits

    @staticmethod
    def backward(ctx, dstate, dlogits_out):
        x, W0, W2, hid, probs = ctx.saved_tensors
        M, K, Hd, N, D, Cc = ctx.dims
        dev = x.device
        dl = torch.empty(M, N, dtype=torch.float32, device=dev)
        cabi.check(lib.bd_categorical_head_backward(ptr(dstate.contiguous().float()), ptr(probs), M, D, Cc, ptr(dl),
                                                    cabi.stream()))
        dl += dlogits_out                     # the logits are an output too (they feed the KL)
        dpre0 = torch.empty(M, Hd, dtype=torch.float32, device=dev)
        dx = torch.empty(M, K, dtype=torch.float32, device=dev)
        wt0, wt2 = _pack(W0, True), _pack(W2, True)
        a = cabi.MlpBwdArgs()
        a.M, a.dout, a.lddo, a.dout_scale = M, ptr(dl), N, 1.0
        a.n_layers = 2
        a.layer[0] = cabi.LayerBwd(ptr(wt0), ptr(hid), Hd, K, cabi.ACT_ELU, ptr(dpre0))
        a.layer[1] = cabi.LayerBwd(ptr(wt2), None, N, Hd, cabi.ACT_NONE, None)
        a.din0, a.ld0, a.w0 = ptr(dx), K, K
        a.din1, a.ld1, a.w1 = None, 0, 0
        a.accumulate = 0
        cabi.check(lib.bd_mlp_backward(C.byref(a), cabi.stream()))
        dW2, db2 = _wgrad(dl, hid, M, N, Hd)
        dW0, db0 = _wgrad(dpre0, x, M, Hd, K)
        return dx, dW0, db0, dW2, db2, None, None, None


class CategoricalBeliefModel(nn.Module):
    """src/models.py:76-117.  ``model`` = build_mlp(input, hidden, D*C, 1, activation): ``model.0``, ``model.2``."""

    def __init__(self, input_size: int, hidden_size: int, discrete_latent_dimensions: int, discrete_latent_classes: int,
                 activation: str = "ELU", device: Optional[str] = None) -> None:
        super().__init__()
        if activation != "ELU":
            raise NotImplementedError("the HIP path implements the reference default activation (ELU)")
        if not torch.cuda.is_available():
            raise RuntimeError("big_dreamer_amd runs on MI355X only: there is no CPU path")
        assert discrete_latent_classes
        self.discrete_latent_dimensions = discrete_latent_dimensions
        self.discrete_latent_classes = discrete_latent_classes
        self.dim = discrete_latent_classes * discrete_latent_dimensions
        self.model = nn.Sequential(nn.Linear(input_size, hidden_size), nn.ELU(), nn.Linear(hidden_size, self.dim),
                                   nn.Identity()).to(device or f"cuda:{torch.cuda.current_device()}")

    def forward(self, belief: Tensor, _noise: Optional[Tensor] = None) -> Tuple[Tensor, Tuple[Tensor, ...]]:
        """belief (..., input_size) -> state (..., D*C), (logits (..., D, C),)."""
        D, Cc = self.discrete_latent_dimensions, self.discrete_latent_classes
        lead = belief.shape[:-1]
        x = belief.reshape(-1, belief.shape[-1]).contiguous().float()
        q = (torch.empty(x.shape[0], D, Cc, device=x.device).exponential_(1) if _noise is None
             else _noise.to(x.device).contiguous().float())
        l0, l2 = self.model[0], self.model[2]
        state, logits = _CategoricalBelief.apply(x, l0.weight, l0.bias, l2.weight, l2.bias, q, D, Cc)
        return state.view(*lead, self.dim), (logits.view(*lead, D, Cc),)


class _KlCategorical(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ql, pl, kl_balance, free_nats):
        T, B, D, Cc = ql.shape
        rows = T * B
        dev = ql.device
        sum_form = int(kl_balance == -1)
        scalars = torch.zeros(4, dtype=torch.float32, device=dev)
        ws = torch.zeros(int(lib.bd_reduce_ws_floats()), dtype=torch.float32, device=dev)
        cabi.check(lib.bd_kl_categorical_forward(ptr(ql), ptr(pl), rows, D, Cc, free_nats, sum_form, ptr(scalars), 0, ptr(ws),
                                                 cabi.stream()))
        ctx.save_for_backward(ql, pl, scalars)
        ctx.cfg = (rows, D, Cc, kl_balance, free_nats)
        if sum_form:
            return scalars[0] / rows                                    # max(KL.sum(2), free_nats).mean((0, 1)): 0-dim
        mean = scalars[:1] / (rows * D)
        clamped = torch.clamp_min(mean, free_nats)
        return kl_balance * clamped + (1 - kl_balance) * clamped        # both sides have the same value (src/dreamer.py:134-144)

    @staticmethod
    def backward(ctx, dkl):
        ql, pl, scalars = ctx.saved_tensors
        rows, D, Cc, kl_balance, free_nats = ctx.cfg
        dq, dp = torch.empty_like(ql), torch.empty_like(pl)
        cabi.check(lib.bd_kl_categorical_backward(ptr(ql), ptr(pl), rows, D, Cc, free_nats, kl_balance, 1.0,
                                                  1.0 / (rows * D), ptr(scalars), 0, ptr(dq), ptr(dp), cabi.stream()))
        g = dkl.reshape(-1)[:1]
        return dq * g, dp * g, None, None


def kl_loss_categorical(post_logits: Tensor, prior_logits: Tensor, kl_balance: float, free_nats: float) -> Tensor:
    """Dreamer._kl_loss, Categorical branch (src/dreamer.py:102-146): logits (T, B, D, C) -> loss of shape (1,)
    (0-dim for kl_balance == -1, as the reference's mean over (0, 1) gives)."""
    assert post_logits.shape == prior_logits.shape and post_logits.dim() == 4
    return _KlCategorical.apply(post_logits.contiguous().float(), prior_logits.contiguous().float(), float(kl_balance),
                                float(free_nats))
