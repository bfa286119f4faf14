"""Reference-compatible model classes (src/models.py) on top of the HIP engine.

The classes keep the reference's constructor signatures, attribute names and ``state_dict`` keys/shapes
(SURVEY.md section 8b), so ``models=`` checkpoints of the reference load unchanged.  Their parameters are
*views into the engine's flat fp32 buffers* (one per optimiser), so the HIP kernels, the fused clip+Adam and
the RCCL all-reduce all see the same memory; after ``load_state_dict`` the packed kernel copies are refreshed.

``forward`` runs the HIP kernels (inference-style: no autograd graph; training goes through
``Dreamer.train_step`` whose backward schedule is explicit).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import Tensor, nn

from .engine import DreamerEngine
from .synth import DENSE_LAYERS


def _alias(module: nn.Module, eng: DreamerEngine, mod: str, prefix: str = "") -> None:
    """Point every parameter of `module` at the engine's storage for (mod, prefix+name)."""
    for name, p in module.named_parameters():
        view = eng.W(mod, prefix + name)
        assert tuple(view.shape) == tuple(p.shape), (mod, prefix + name, view.shape, p.shape)
        p.data = view
        p.requires_grad_(False)     # gradients live in the engine's flat gradient buffers


class _EngineBacked(nn.Module):
    """Base: re-pack the kernel weight copies whenever a state dict is loaded."""

    def _bind(self, eng: DreamerEngine, mod: str, group: str) -> None:
        object.__setattr__(self, "_eng", eng)
        self._mod, self._group = mod, group
        _alias(self, eng, mod)
        self.register_load_state_dict_post_hook(lambda m, keys: m._eng.pack(m._group))


class DenseModel(_EngineBacked):
    """src/models.py:365-408: build_mlp(input, hidden, output, n_layers=4) -> ``model.{0,2,4,6,8}``."""

    def __init__(self, input_size: int, hidden_size: int, output_size: int = 1, activation: str = "ELU",
                 n_layers: int = DENSE_LAYERS, distribution: str = "normal", *, engine: DreamerEngine, module: str,
                 prefix: str):
        super().__init__()
        assert activation == "ELU" and n_layers == DENSE_LAYERS, "the HIP path implements the reference defaults"
        layers, i = [], input_size
        for _ in range(n_layers):
            layers += [nn.Linear(i, hidden_size), nn.ELU()]
            i = hidden_size
        layers += [nn.Linear(i, output_size), nn.Identity()]
        self.model = nn.Sequential(*layers)
        self.distribution = distribution
        self._in, self._out, self._prefix = input_size, output_size, prefix
        group = {"critic": "critic", "critic_target": "critic_target", "actor": "actor"}.get(module, "model")
        self._bind(engine, module, group)

    def forward(self, *args: Tensor) -> Tensor:
        self._eng.join()      # order after any queued pipeline work (engine.train_step)
        x = torch.cat(args, dim=-1) if len(args) == 2 else args[0]
        lead = x.shape[:-1]
        x2 = x.reshape(-1, self._in).contiguous().float()
        out, _, _ = self._eng.dense_forward(self._mod, self._prefix, "api_" + self._mod, x2, self._in, x2.shape[0],
                                            self._out)
        return out.clone().view(*lead, self._out)


class GaussianBeliefHolder(nn.Module):
    """Parameter container with the reference's names (``model.0``, ``model.2``; src/models.py:44-73)."""

    def __init__(self, input_size: int, hidden_size: int, state_size: int, min_std_dev: float):
        super().__init__()
        self.min_std_dev = min_std_dev
        self.model = nn.Sequential(nn.Linear(input_size, hidden_size), nn.ELU(), nn.Linear(hidden_size, 2 * state_size),
                                   nn.Identity())


class CategoricalBeliefHolder(nn.Module):
    """Parameter container of CategoricalBeliefModel (src/models.py:76-100): build_mlp(in, hidden, D*C, 1) ->
    ``model.0``, ``model.2``."""

    def __init__(self, input_size: int, hidden_size: int, discrete_latent_dimensions: int, discrete_latent_classes: int):
        super().__init__()
        self.discrete_latent_dimensions = discrete_latent_dimensions
        self.discrete_latent_classes = discrete_latent_classes
        self.dim = discrete_latent_classes * discrete_latent_dimensions
        self.model = nn.Sequential(nn.Linear(input_size, hidden_size), nn.ELU(), nn.Linear(hidden_size, self.dim),
                                   nn.Identity())


class TransitionModel(_EngineBacked):
    """src/models.py:120-299.  ``forward`` returns the reference's 5-tuple; with latent_distribution="Categorical" the
    params are the 1-tuples ``(logits (T, B, D, C),)`` the reference intends (it stores the tuple itself in its
    per-step lists and crashes in ``stack`` at HEAD, src/models.py:259-260,270-271,295)."""

    def __init__(self, belief_size: int, state_size: int, action_size: int, hidden_size: int, embedding_size: int,
                 activation: Optional[str] = "ELU", min_std_dev: float = 0.1,
                 latent_distribution: Optional[str] = "Gaussian", discrete_latent_dimensions: Optional[int] = 32,
                 discrete_latent_classes: Optional[int] = 32, *, engine: DreamerEngine):
        super().__init__()
        assert latent_distribution in ["Gaussian", "Categorical"], f"{latent_distribution}"      # src/models.py:144
        assert activation == "ELU"
        self.min_std_dev = min_std_dev
        self.latent_distribution = latent_distribution
        self.rnn = nn.GRUCell(belief_size, belief_size)
        self.fc_embed_state_action = nn.Sequential(nn.Linear(state_size + action_size, belief_size), nn.ELU())
        if latent_distribution == "Gaussian":
            self.belief_prior = GaussianBeliefHolder(belief_size, hidden_size, state_size, min_std_dev)
            self.belief_posterior = GaussianBeliefHolder(belief_size + embedding_size, hidden_size, state_size, min_std_dev)
            self._cat = None
        else:
            D, Cc = discrete_latent_dimensions, discrete_latent_classes
            assert state_size == D * Cc, "Categorical latents: state_size = dimensions * classes (src/planet.py:56-57)"
            assert engine.d.categorical and (engine.d.cat_D, engine.d.cat_C) == (D, Cc)
            self.belief_prior = CategoricalBeliefHolder(belief_size, hidden_size, D, Cc)
            self.belief_posterior = CategoricalBeliefHolder(belief_size + embedding_size, hidden_size, D, Cc)
            self._cat = (D, Cc)
        self._dims = (belief_size, state_size, action_size, hidden_size, embedding_size)
        self._bind(engine, "transition_model", "model")

    @torch.no_grad()
    def forward(self, init_state: Tensor, actions: Tensor, init_belief: Tensor, embeddings: Optional[Tensor] = None,
                nonterminals: Optional[Tensor] = None, _noise: Optional[Tuple[Tensor, Tensor]] = None):
        """init_state (B,S), actions (T,B,A), init_belief (B,Be), embeddings (T,B,E), nonterminals (T,B,1) ->
        beliefs (T,B,Be), prior_states, (prior_means, prior_stds), posterior_states, (post_means, post_stds)."""
        self._eng.join()      # order after any queued pipeline work (engine.train_step)
        eng = self._eng
        Be, S, A, Hd, E = self._dims
        T, B = actions.shape[0], actions.shape[1]
        M = T * B
        f = lambda t: t.contiguous().float()
        cat = self._cat
        # state draws: standard normals, or -- Categorical -- the sampler's Exp(1) variates per class (T, B, D*C)
        draw = (lambda: torch.empty(T, B, S, device=eng.dev).exponential_()) if cat else \
            (lambda: torch.randn(T, B, S, device=eng.dev))
        params = (lambda a, b: (a.clone().view(T, B, cat[0], cat[1]),)) if cat else (lambda a, b: (v(a), v(b)))
        if embeddings is None:
            # prior-only rollout (src/models.py:241,296-297; the MPC planner's call, src/planner.py:65): the sampled
            # prior state is fed back; posterior outputs are None
            eps_p = draw() if _noise is None else f(_noise[0])
            feat, pm, ps = eng.observe(f(actions), None if nonterminals is None else f(nonterminals), None, eps_p,
                                       f(init_belief), f(init_state), T, B, save=False, tag="api_", prior_only=True)
            feat = feat.view(T, B, Be + S)
            v = lambda t: t.clone().view(T, B, S)
            return feat[..., :Be].clone(), feat[..., Be:].clone(), params(pm, ps), None, None
        pre = eng.buf("api_pre_emb", M, Hd)
        eng.mlp_forward(M, f(embeddings).view(M, E), E, E, [("q1e", None, Hd, E, 0)], None, pre, Hd)
        if _noise is None:      # (prior eps, posterior eps); the parity tests inject the oracle's draws
            eps_p, eps_q = draw(), draw()
        else:
            eps_p, eps_q = f(_noise[0]), f(_noise[1])
        feat, qm, qs = eng.observe(f(actions), None if nonterminals is None else f(nonterminals), pre, eps_q,
                                   f(init_belief), f(init_state), T, B, save=False, tag="api_")
        pst, pm, ps = eng.prior_head(feat, M, eps_p.view(M, S) if cat else eps_p, tag="api_")
        feat = feat.view(T, B, Be + S)
        v = lambda t: t.clone().view(T, B, S)
        return (feat[..., :Be].clone(), v(pst), params(pm, ps), feat[..., Be:].clone(), params(qm, qs))


class ActorModel(_EngineBacked):
    """src/models.py:466-524 (Gaussian action distribution).  ``forward`` returns (action_mean, action_std)."""

    def __init__(self, belief_size: int, state_size: int, hidden_size: int, action_size: int,
                 activation_function: str = "ELU", action_distribution: str = "Gaussian", min_std: float = 1e-4,
                 init_std: float = 5, mean_scale: float = 5, n_layers: int = DENSE_LAYERS, *, engine: DreamerEngine):
        super().__init__()
        if action_distribution != "Gaussian":
            raise NotImplementedError("only the Gaussian action distribution is on the hot path")
        layers, i = [], belief_size + state_size
        for _ in range(n_layers):
            layers += [nn.Linear(i, hidden_size), nn.ELU()]
            i = hidden_size
        layers += [nn.Linear(i, 2 * action_size), nn.Identity()]
        self.model = nn.Sequential(*layers)
        self._min_std, self._init_std, self._mean_scale = min_std, init_std, mean_scale
        self.raw_init_std = torch.log(torch.exp(torch.tensor(float(init_std))) - 1)
        self.action_distribution = action_distribution
        self._sizes = (belief_size + state_size, hidden_size, 2 * action_size)
        self._bind(engine, "actor", "actor")

    @torch.no_grad()
    def forward(self, belief: Tensor, state: Tensor) -> Tuple[Tensor, Tensor]:
        """src/models.py:506-517: (action_mean, action_std) = (5 tanh(m / 5), softplus(r + raw_init_std) + min_std).
        Inference entry point (the training step runs the actor inside the imagination kernels): the dense chain runs
        on bd_mlp_forward with the weights packed per call."""
        from . import _cabi as cabi
        from .categorical import _pack
        import ctypes as C
        self._eng.join()      # order after any queued pipeline work (engine.train_step)
        F, Hd, out = self._sizes
        lead = belief.shape[:-1]
        x = torch.cat([belief, state], dim=-1).reshape(-1, F).contiguous().float()
        M = x.shape[0]
        res = torch.empty(M, out, dtype=torch.float32, device=x.device)
        a = cabi.MlpFwdArgs()
        a.M, a.in0, a.ld0, a.w0 = M, x.data_ptr(), F, F
        a.in1, a.ld1, a.w1 = None, 0, 0
        a.n_layers = DENSE_LAYERS + 1
        keep, k = [], F
        for l in range(DENSE_LAYERS + 1):
            lin = self.model[2 * l]
            n = out if l == DENSE_LAYERS else Hd
            keep.append(_pack(lin.weight, False))
            a.layer[l] = cabi.Layer(keep[-1].data_ptr(), lin.bias.data_ptr(), n, k,
                                    cabi.ACT_NONE if l == DENSE_LAYERS else cabi.ACT_ELU, None)
            k = n
        a.out, a.ldo = res.data_ptr(), out
        cabi.check(cabi.lib.bd_mlp_forward(C.byref(a), cabi.stream()))
        m, r = torch.chunk(res, 2, dim=1)
        mean = self._mean_scale * torch.tanh(m / self._mean_scale)
        std = torch.nn.functional.softplus(r + self.raw_init_std.to(r.device)) + self._min_std
        return mean.view(*lead, -1), std.view(*lead, -1)


class CnnImageEncoder(_EngineBacked):
    """src/models.py:527-564: 4 x (Conv2d k4 s2 + ELU), Flatten, Identity | Linear(1024, E) -- ``model.{0,2,4,6[,9]}``.
    The convolutions run on this library's gather-GEMM kernels (csrc/conv.hip through conv_stack.ConvStacks)."""

    def __init__(self, embedding_size: int, activation: str = "ELU", *, engine: DreamerEngine):
        super().__init__()
        assert activation == "ELU"
        self.model = nn.Sequential(nn.Conv2d(3, 32, 4, 2), nn.ELU(), nn.Conv2d(32, 64, 4, 2), nn.ELU(),
                                   nn.Conv2d(64, 128, 4, 2), nn.ELU(), nn.Conv2d(128, 256, 4, 2), nn.ELU(), nn.Flatten(),
                                   nn.Identity() if embedding_size == 1024 else nn.Linear(1024, embedding_size))
        self._bind(engine, "encoder", "model")

    @torch.no_grad()
    def forward(self, observation: Tensor) -> Tensor:
        self._eng.join()      # order after any queued pipeline work (engine.train_step)
        lead = observation.shape[:-3]
        emb, _ = self._eng.encode_pixels(observation.reshape(-1, 3, 64, 64).contiguous().float(), grad=False)
        return emb.view(*lead, -1).clone()


class ObservationModel(_EngineBacked):
    """src/models.py:319-362: Linear + 4 x ConvTranspose2d -- ``decoder.{0,2,4,6,8}``."""

    def __init__(self, belief_size: int, state_size: int, embedding_size: int, activation: str = "ELU", *,
                 engine: DreamerEngine):
        super().__init__()
        assert activation == "ELU"
        self.output_shape = (3, 64, 64)
        self.decoder = nn.Sequential(nn.Linear(belief_size + state_size, embedding_size), nn.Identity(),
                                     nn.ConvTranspose2d(embedding_size, 128, 5, 2), nn.ELU(),
                                     nn.ConvTranspose2d(128, 64, 5, 2), nn.ELU(), nn.ConvTranspose2d(64, 32, 6, 2), nn.ELU(),
                                     nn.ConvTranspose2d(32, 3, 6, 2))
        self._bind(engine, "observation_model", "model")

    @torch.no_grad()
    def forward(self, belief: Tensor, state: Tensor) -> Tensor:
        self._eng.join()      # order after any queued pipeline work (engine.train_step)
        lead = belief.shape[:-1]
        x = torch.cat([belief, state], dim=-1).reshape(-1, belief.shape[-1] + state.shape[-1]).contiguous().float()
        return self._eng.decode_pixels(x).view(*lead, 3, 64, 64).clone()


def encoder_for(engine: DreamerEngine, observation_size: int, hidden_size: int, embedding_size: int) -> DenseModel:
    """State-observation encoder (src/planet.py:195-200)."""
    return DenseModel(observation_size, hidden_size, embedding_size, engine=engine, module="encoder", prefix="enc")
