"""Deterministic synthetic inputs for the Dreamer world-model training step.

Everything here is derived from numpy ``Generator(PCG64(seed))`` streams so that the golden
generator (which imports the reference in the build container), the parity tests and ``bench.py``
(which run on the GPU box, where the reference does not exist) all see bit-identical weights,
replay contents, batches and noise without any of them having to be stored.

Shapes follow SURVEY.md section 8: time-major ``(time, batch, feature)`` fp32.
Parameter names/shapes follow the reference ``state_dict`` (``src/models.py:149-167`` for the RSSM,
``src/utils.py:368-404`` ``build_mlp`` for the dense heads).
"""
from __future__ import annotations

import dataclasses
from typing import Dict, List, Tuple

import numpy as np


@dataclasses.dataclass(frozen=True)
class Dims:
    """Problem dimensions (names as in SURVEY.md section 8)."""

    B: int = 50      # batch_size            (conf/config.yaml:22)
    L: int = 50      # seq_len / chunk       (conf/config.yaml:23)
    H: int = 15      # planning_horizon      (conf/config.yaml:31)
    Be: int = 200    # belief_size
    S: int = 30      # state_size
    Hd: int = 200    # hidden_size
    E: int = 1024    # embedding_size
    A: int = 1       # action_size (Pendulum 1)
    O: int = 3       # observation_size (Pendulum 3, state observations)
    n_entropy: int = 100  # SampleDist samples (src/models.py:681)
    pixel: bool = False   # 64x64x3 pixel observations (conv encoder / decoder, src/models.py:319-362,527-564)
    cat_D: int = 0        # latent_distribution="Categorical": discrete_latent_dimensions (0 = Gaussian latents)
    cat_C: int = 0        #                                    discrete_latent_classes; S must equal cat_D * cat_C (src/planet.py:56-57)
    use_discount: bool = False   # discount_model head + Bernoulli loss + discounted actor objective (src/dreamer.py:80-86,239-251,346-351)

    def __post_init__(self):
        assert (self.cat_D == 0) == (self.cat_C == 0) and (self.cat_D == 0 or self.S == self.cat_D * self.cat_C), \
            "Categorical latents: state_size = discrete_latent_dimensions * discrete_latent_classes"

    @property
    def categorical(self) -> bool:
        return self.cat_D > 0

    @property
    def head_out(self) -> int:
        """Output width of belief_prior / belief_posterior: (mean, raw std) or D*C logits (src/models.py:44-117)."""
        return self.S if self.cat_D else 2 * self.S

    @property
    def T(self) -> int:
        return self.L - 1

    @property
    def N(self) -> int:
        return self.T * self.B

    @property
    def Hm(self) -> int:  # H' = number of imagination steps actually run (src/dreamer.py:213)
        return self.H - 1

    @property
    def transitions_per_step(self) -> int:
        """GRU-cell applications per train_step: (L-1)*B*H."""
        return self.T * self.B * self.H


CONFIG1 = Dims(B=50, L=50, H=15, Be=32, S=30, Hd=32, E=1024, A=1, O=3)     # BASELINE.json configs[0]
CONFIG2 = Dims()                                                           # BASELINE.json configs[1]
PIXEL_SHAPE = (3, 64, 64)
TINY = Dims(B=3, L=5, H=4, Be=24, S=6, Hd=20, E=40, A=2, O=5, n_entropy=100)
TINY_DISCOUNT = Dims(B=3, L=5, H=4, Be=24, S=6, Hd=20, E=40, A=2, O=5, n_entropy=100, use_discount=True)
TINY_PIXEL = Dims(B=2, L=4, H=3, Be=24, S=6, Hd=20, E=1024, A=2, O=12288, pixel=True)      # Identity after Flatten
TINY_PIXEL_LIN = Dims(B=2, L=3, H=3, Be=20, S=5, Hd=24, E=48, A=1, O=12288, pixel=True)     # Linear(1024, E) tail
CONFIG3 = Dims(A=17, O=12288, pixel=True)                                                   # BASELINE.json configs[2]
SMALL = Dims(B=7, L=9, H=6, Be=48, S=10, Hd=36, E=72, A=3, O=4, n_entropy=100)
# Categorical latents (algorithm=dreamerV2 latent_distribution=Categorical): ragged factors, then the reference's 32 x 32
CAT_TINY = Dims(B=3, L=5, H=4, Be=24, S=15, Hd=20, E=40, A=2, O=5, cat_D=3, cat_C=5)
CAT_32 = Dims(B=18, L=6, H=5, Be=64, S=1024, Hd=48, E=96, A=3, O=6, cat_D=32, cat_C=32)       # 32 x 32 latents, two row tiles
# BASELINE configs[4] as stated -- pixel observations AND Categorical latents (decoder reads [h; one-hot s], K = Be + D*C)
CAT_PIXEL_TINY = Dims(B=2, L=4, H=3, Be=24, S=15, Hd=20, E=1024, A=2, O=12288, pixel=True, cat_D=3, cat_C=5)
CAT_PIXEL_32 = Dims(B=3, L=3, H=3, Be=40, S=1024, Hd=32, E=1024, A=17, O=12288, pixel=True, cat_D=32, cat_C=32)   # 32 x 32, A = 17
CONFIG5 = Dims(B=100, A=17, O=12288, pixel=True, S=1024, cat_D=32, cat_C=32)   # BASELINE configs[4] per GPU: batch 800 / 8
CONFIG5_STATE = Dims(B=100, S=1024, cat_D=32, cat_C=32)                         # same latents on state observations

DENSE_LAYERS = 4  # DenseModel / ActorModel n_layers (src/models.py:378,482)


def _mlp_shapes(prefix: str, sizes: List[int]) -> List[Tuple[str, Tuple[int, ...]]]:
    """``build_mlp`` puts Linear layers at even Sequential indices (src/utils.py:396-404)."""
    out = []
    for i in range(len(sizes) - 1):
        out.append((f"{prefix}.model.{2 * i}.weight", (sizes[i + 1], sizes[i])))
        out.append((f"{prefix}.model.{2 * i}.bias", (sizes[i + 1],)))
    return out


def param_shapes(d: Dims) -> Dict[str, List[Tuple[str, Tuple[int, ...]]]]:
    """Parameter (name, shape) lists per module, in ``module.parameters()`` order."""
    tm = [
        ("rnn.weight_ih", (3 * d.Be, d.Be)),
        ("rnn.weight_hh", (3 * d.Be, d.Be)),
        ("rnn.bias_ih", (3 * d.Be,)),
        ("rnn.bias_hh", (3 * d.Be,)),
        ("fc_embed_state_action.0.weight", (d.Be, d.S + d.A)),
        ("fc_embed_state_action.0.bias", (d.Be,)),
        ("belief_prior.model.0.weight", (d.Hd, d.Be)),
        ("belief_prior.model.0.bias", (d.Hd,)),
        ("belief_prior.model.2.weight", (d.head_out, d.Hd)),
        ("belief_prior.model.2.bias", (d.head_out,)),
        ("belief_posterior.model.0.weight", (d.Hd, d.Be + d.E)),
        ("belief_posterior.model.0.bias", (d.Hd,)),
        ("belief_posterior.model.2.weight", (d.head_out, d.Hd)),
        ("belief_posterior.model.2.bias", (d.head_out,)),
    ]
    feat = d.Be + d.S
    hid = [d.Hd] * DENSE_LAYERS

    def strip(lst):
        return [(n.split(".", 1)[1], s) for n, s in lst]

    if d.pixel:
        # CnnImageEncoder.model (src/models.py:538-552) / ObservationModel.decoder (src/models.py:338-348)
        enc = []
        for i, (ci, co) in enumerate([(3, 32), (32, 64), (64, 128), (128, 256)]):
            enc += [(f"model.{2 * i}.weight", (co, ci, 4, 4)), (f"model.{2 * i}.bias", (co,))]
        if d.E != 1024:
            enc += [("model.9.weight", (d.E, 1024)), ("model.9.bias", (d.E,))]
        obs = [("decoder.0.weight", (d.E, feat)), ("decoder.0.bias", (d.E,))]
        for idx, (ci, co, k) in zip((2, 4, 6, 8), [(d.E, 128, 5), (128, 64, 5), (64, 32, 6), (32, 3, 6)]):
            obs += [(f"decoder.{idx}.weight", (ci, co, k, k)), (f"decoder.{idx}.bias", (co,))]
    else:
        enc = strip(_mlp_shapes("x", [d.O] + hid + [d.E]))
        obs = strip(_mlp_shapes("x", [feat] + hid + [d.O]))
    out = {
        "transition_model": tm,
        "observation_model": obs,
        "reward_model": strip(_mlp_shapes("x", [feat] + hid + [1])),
        "encoder": enc,
        "actor": strip(_mlp_shapes("x", [feat] + hid + [2 * d.A])),
        "critic": strip(_mlp_shapes("x", [feat] + hid + [1])),
    }
    if d.use_discount:      # DenseModel(belief + state, hidden) (src/dreamer.py:80-85); joins the model optimiser last (:167-169)
        out["discount_model"] = strip(_mlp_shapes("x", [feat] + hid + [1]))
    return out


# Order in which the reference concatenates the world-model parameters for its optimiser
# (src/dreamer.py:160-165).
MODEL_MODULES = ("transition_model", "observation_model", "reward_model", "encoder")


def model_modules(d: "Dims"):
    """Modules of the world-model optimiser in the reference's parameter order (src/dreamer.py:160-169)."""
    return MODEL_MODULES + (("discount_model",) if d.use_discount else ())


def make_params(d: Dims, seed: int = 0) -> Dict[str, Dict[str, np.ndarray]]:
    """PyTorch-default-like init U(-1/sqrt(fan_in), 1/sqrt(fan_in)) from a numpy stream."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out: Dict[str, Dict[str, np.ndarray]] = {}
    for mod, lst in param_shapes(d).items():
        sd = {}
        fan_in = None
        for name, shape in lst:
            if name.startswith("rnn."):
                bound = 1.0 / np.sqrt(d.Be)
            elif len(shape) == 4:      # conv / conv-transpose kernels: fan_in = dim1 * kh * kw (PyTorch convention)
                fan_in = shape[1] * shape[2] * shape[3]
                bound = 1.0 / np.sqrt(fan_in)
            elif len(shape) == 2:
                fan_in = shape[1]
                bound = 1.0 / np.sqrt(fan_in)
            else:
                bound = 1.0 / np.sqrt(fan_in)
            sd[name] = rng.uniform(-bound, bound, size=shape).astype(np.float32)
        out[mod] = sd
    out["critic_target"] = {k: v.copy() for k, v in out["critic"].items()}
    return out


def make_replay(d: Dims, rows: int = 5000, seed: int = 0) -> Dict[str, np.ndarray]:
    """Synthetic replay contents: obs N(0,1), actions U(-1,1), rewards N(0,1),
    nonterminals Bernoulli(0.999)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return {
        "observations": rng.standard_normal((rows, d.O), dtype=np.float32),
        "actions": rng.uniform(-1.0, 1.0, size=(rows, d.A)).astype(np.float32),
        "rewards": rng.standard_normal((rows,), dtype=np.float32),
        "nonterminals": (rng.random((rows, 1)) < 0.999).astype(np.float32),
    }


def make_batch(d: Dims, seed: int = 0, p_terminal: float = 0.02) -> Dict[str, np.ndarray]:
    """A time-major batch as ``ExperienceReplay.sample`` returns it (src/memory.py:87-104).

    Terminals are made more frequent than in the replay so that small batches exercise the
    nonterminal mask (src/models.py:247)."""
    rng = np.random.Generator(np.random.PCG64(seed + 1000))
    if d.pixel:   # as ExperienceReplay.sample returns pixels: 5-bit quantised, centred, dequantised (src/utils.py:299-317)
        q = rng.integers(0, 32, size=(d.L, d.B) + PIXEL_SHAPE).astype(np.float32)
        obs = q / 32.0 - 0.5 + rng.random((d.L, d.B) + PIXEL_SHAPE, dtype=np.float32) / 32.0
    else:
        obs = rng.standard_normal((d.L, d.B, d.O), dtype=np.float32)
    return {
        "observations": obs,
        "actions": rng.uniform(-1.0, 1.0, size=(d.L, d.B, d.A)).astype(np.float32),
        "rewards": rng.standard_normal((d.L, d.B), dtype=np.float32),
        "nonterminals": (rng.random((d.L, d.B, 1)) >= p_terminal).astype(np.float32),
    }


class NoiseStream:
    """Standard-normal draws in the reference's RNG call order (SURVEY.md section 8a, R-RNG)."""

    def __init__(self, seed: int = 0):
        self.rng = np.random.Generator(np.random.PCG64(seed + 2000))
        self.calls: List[Tuple[int, ...]] = []

    def normal(self, shape) -> np.ndarray:
        shape = tuple(int(s) for s in shape)
        self.calls.append(shape)
        return self.rng.standard_normal(shape, dtype=np.float32)

    def exponential(self, shape) -> np.ndarray:
        """Exp(1) draws: what torch.multinomial's single-draw path consumes per class (q; sample = argmax(probs / q))."""
        shape = tuple(int(s) for s in shape)
        self.calls.append(("exp",) + shape)
        return self.rng.standard_exponential(shape, dtype=np.float32)


def make_noise(d: Dims, seed: int = 0) -> Dict[str, np.ndarray]:
    """All noise of one train_step, drawn in reference order:
    for t<T: prior (B,S) then posterior (B,S)  (src/models.py:256,267 -> :72);
    for t<H': action (N,A), entropy (100,N,A) (src/dreamer.py:443-444), prior (N,S) (src/dreamer.py:223).
    """
    ns = NoiseStream(seed)
    # Categorical latents: the state draws are the Exp(1) variates of the sampler, one per class, viewed (rows*D, C) as
    # OneHotCategorical.sample hands them to torch.multinomial; stored (.., rows, D*C)
    state_draw = (lambda rows: ns.exponential((rows * d.cat_D, d.cat_C)).reshape(rows, d.S)) if d.categorical else \
        (lambda rows: ns.normal((rows, d.S)))
    obs_prior = np.empty((d.T, d.B, d.S), np.float32)
    obs_post = np.empty((d.T, d.B, d.S), np.float32)
    for t in range(d.T):
        obs_prior[t] = state_draw(d.B)
        obs_post[t] = state_draw(d.B)
    act = np.empty((d.Hm, d.N, d.A), np.float32)
    ent = np.empty((d.Hm, d.n_entropy, d.N, d.A), np.float32)
    img = np.empty((d.Hm, d.N, d.S), np.float32)
    for t in range(d.Hm):
        act[t] = ns.normal((d.N, d.A))
        ent[t] = ns.normal((d.n_entropy, d.N, d.A))
        img[t] = state_draw(d.N)
    return {"obs_prior": obs_prior, "obs_post": obs_post, "action": act, "entropy": ent,
            "img_prior": img}


def make_planner_noise(d: Dims, B: int, horizon: int, iters: int, candidates: int, seed: int = 0) -> Dict[str, np.ndarray]:
    """Noise of one MPCPlanner.forward in reference order (src/planner.py:53-65): per CEM iteration the action draws
    (H,B,candidates,A), then one prior-state draw (B*candidates,S) per rollout step (src/models.py:256)."""
    ns = NoiseStream(seed)
    act = np.empty((iters, horizon, B, candidates, d.A), np.float32)
    st = np.empty((iters, horizon, B * candidates, d.S), np.float32)
    for it in range(iters):
        act[it] = ns.normal((horizon, B, candidates, d.A))
        for t in range(horizon):
            st[it, t] = ns.normal((B * candidates, d.S))
    return {"action": act, "state": st}


# CategoricalBeliefModel cases (src/models.py:76-117): name -> (rows, input, hidden, D groups, C classes, seed)
CATEGORICAL_CASES = {
    "cat_small": (7, 24, 20, 4, 6, 21),
    "cat_wide_classes": (5, 16, 12, 3, 40, 22),          # more classes than a half wave has lanes
    "cat_reference": (16, 200, 200, 32, 32, 23),        # discrete_latent_dimensions x classes of conf/config.yaml
}


def make_categorical_case(rows: int, inp: int, hid: int, D: int, C: int, seed: int) -> Dict[str, np.ndarray]:
    """Weights (PyTorch-default-like uniform init), input, output cotangents and a second set of logits for the KL.
    (The Exp(1) sampling noise is drawn inside ATen by the reference run; the golden file stores it.)"""
    rng = np.random.Generator(np.random.PCG64(seed + 7000))
    u = lambda shape, k: rng.uniform(-1.0 / np.sqrt(k), 1.0 / np.sqrt(k), size=shape).astype(np.float32)
    f = lambda *shape: rng.standard_normal(shape, dtype=np.float32)
    return {"model.0.weight": u((hid, inp), inp), "model.0.bias": u((hid,), inp),
            "model.2.weight": (3.0 * u((D * C, hid), hid)).astype(np.float32), "model.2.bias": u((D * C,), hid),
            "x": f(rows, inp),
            "g_state": f(rows, D * C), "g_logits": (0.1 * f(rows, D, C)).astype(np.float32),
            "other_logits": (1.5 * f(rows, D, C)).astype(np.float32)}
