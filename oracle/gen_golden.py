"""Generate golden vectors by running the *reference itself* (read-only at /root/reference) on the
deterministic synthetic inputs of ``big_dreamer_amd.synth``.  TEST INFRASTRUCTURE.

Runs only in the build container (the reference does not exist on the GPU box).  Output:
``tests/golden/<config>.npz`` -- data only (inputs are re-derivable from seeds, outputs are stored).

The reference imports need four inert stubs for packages that are absent here and do no arithmetic
(SURVEY.md section 8c): torchtyping, typeguard (decorators), cv2, gym.

Noise is *injected* rather than captured: ``torch.randn_like`` and
``torch.distributions.normal._standard_normal`` are replaced by draws from ``synth.NoiseStream`` so the
reference consumes exactly the arrays ``synth.make_noise`` reproduces later.
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from big_dreamer_amd import synth  # noqa: E402

REF = "/root/reference/src"


def _import_reference():
    tt = types.ModuleType("torchtyping")

    class _TT:
        def __class_getitem__(cls, item):
            return torch.Tensor

    tt.TensorType = _TT
    tt.patch_typeguard = lambda: None
    tg = types.ModuleType("typeguard")
    tg.typechecked = lambda f: f
    sys.modules.update({"torchtyping": tt, "typeguard": tg, "cv2": types.ModuleType("cv2"),
                        "gym": types.ModuleType("gym")})
    sys.path.insert(0, REF)
    import utils
    utils.init_gpu(use_gpu=False)          # must precede dreamer/planet imports (they bind `device` by value)
    import dreamer
    import memory
    return dreamer, memory


class FakeEnv:
    def __init__(self, d):
        self.action_size = d.A
        self.observation_size = (3, 64, 64) if d.pixel else d.O


def ref_params(d: synth.Dims, **over):
    with open(os.path.join(REF, "conf", "config.yaml")) as f:
        p = yaml.safe_load(f)

    def coerce(x):   # PyYAML (YAML 1.1) reads "2e-4" as str; OmegaConf/hydra read it as float
        if isinstance(x, dict):
            return {k: coerce(v) for k, v in x.items()}
        if isinstance(x, str):
            try:
                return float(x)
            except ValueError:
                return x
        return x

    p = coerce(p)
    p.update(dict(disable_cuda=True, pixel_observation=bool(d.pixel), belief_size=d.Be, state_size=d.S,
                  hidden_size=d.Hd, embedding_size=d.E, batch_size=d.B, seq_len=d.L,
                  planning_horizon=d.H, experience_size=6000 if not d.pixel else 64))
    p.update(over)
    return p


class Inject:
    """Context manager replacing the reference's normal draws by a NoiseStream."""

    def __init__(self, stream: synth.NoiseStream):
        self.s = stream

    def __enter__(self):
        import torch.distributions.normal as tdn
        self._rl, self._sn, self._rn = torch.randn_like, tdn._standard_normal, torch.randn
        torch.randn_like = lambda x, **kw: torch.from_numpy(self.s.normal(x.shape))
        torch.randn = lambda *size, **kw: torch.from_numpy(self.s.normal(size))          # src/planner.py:53
        tdn._standard_normal = lambda shape, dtype, device: torch.from_numpy(self.s.normal(tuple(shape)))
        return self

    def __exit__(self, *a):
        import torch.distributions.normal as tdn
        torch.randn_like, tdn._standard_normal, torch.randn = self._rl, self._sn, self._rn


class CategoricalShims:
    """The two repairs without which the reference cannot run latent_distribution="Categorical" at HEAD (SURVEY.md
    section 8c), applied AROUND its code -- no control flow of the reference is restated:
      1. ``stack`` (src/utils.py:36-43, imported by name into models.py and dreamer.py) is handed lists whose entries are
         the 1-tuples ``(logits,)`` that ``TransitionModel.forward`` (src/models.py:259-260,270-271) and
         ``Dreamer.imagine_ahead`` (src/dreamer.py:224-227) store: the shim unwraps them before ``torch.stack``.
      2. ``TransitionModel.forward`` returns ``posterior_params = (stack(...))`` -- parentheses, not a tuple
         (src/models.py:295) -- which ``_get_dist`` then cannot unpack: the wrapper returns ``(tensor,)``.
    Sampling: ``OneHotCategoricalStraightThrough.rsample`` reaches ``torch.multinomial(probs, 1, True)``, whose draws
    happen inside ATen.  Its single-draw algorithm is ``q = empty_like(probs).exponential_(1); argmax(probs / q)``; the
    shim computes exactly that with q taken from the injected NoiseStream, after checking on the live call (same probs,
    re-seeded generator) that the formula reproduces the library's sample."""

    def __init__(self, stream: synth.NoiseStream):
        self.s = stream
        self.checked = 0

    def __enter__(self):
        import dreamer as ref_dreamer
        import models as ref_models
        self._stack_m, self._stack_d = ref_models.stack, ref_dreamer.stack
        self._fwd, self._mn = ref_models.TransitionModel.forward, torch.multinomial

        def stack(x):
            return torch.stack([e[0] if isinstance(e, tuple) else e for e in x[1:]], dim=0)

        fwd = self._fwd

        def forward(self_, *a, **kw):
            out = fwd(self_, *a, **kw)
            if out[4] is not None and not isinstance(out[4], tuple):
                out = out[:4] + ((out[4],),)
            return out

        real_mn = self._mn

        def multinomial(probs, num_samples, replacement=False, **kw):
            assert num_samples == 1 and probs.dim() == 2
            if self.checked < 3:       # the formula IS the library's draw: same probs, same generator state
                torch.manual_seed(1234 + self.checked)
                q_lib = torch.empty_like(probs).exponential_(1)
                torch.manual_seed(1234 + self.checked)
                assert torch.equal(real_mn(probs, 1, True), (probs / q_lib).argmax(-1, keepdim=True))
                self.checked += 1
            q = torch.from_numpy(self.s.exponential(probs.shape))
            return (probs / q).argmax(-1, keepdim=True)

        ref_models.stack = ref_dreamer.stack = stack
        ref_models.TransitionModel.forward = forward
        torch.multinomial = multinomial
        return self

    def __exit__(self, *a):
        import dreamer as ref_dreamer
        import models as ref_models
        ref_models.stack, ref_dreamer.stack = self._stack_m, self._stack_d
        ref_models.TransitionModel.forward = self._fwd
        torch.multinomial = self._mn


def build_agent(dreamer_mod, d, P, **over):
    if d.categorical:
        over = dict(latent_distribution="Categorical", discrete_latent_dimensions=d.cat_D,
                    discrete_latent_classes=d.cat_C, **over)
    if d.use_discount:
        over = dict(use_discount=True, **over)
    agent = dreamer_mod.Dreamer(ref_params(d, **over), FakeEnv(d))
    for mod in ("transition_model", "observation_model", "reward_model", "encoder", "actor", "critic",
                "critic_target") + (("discount_model",) if d.use_discount else ()):
        getattr(agent, mod).load_state_dict({k: torch.from_numpy(v.copy()) for k, v in P[mod].items()})
    return agent


def t2n(x):
    return x.detach().numpy().copy()


def store(out, key, a, full):
    """Full tensor for the small cases; (sum, abssum, strided 257-sample) for the full-size configs."""
    if full or a.size <= 64:
        out[key] = a
    else:
        out[key + ".sum"] = np.array(a.astype(np.float64).sum())
        out[key + ".abssum"] = np.array(np.abs(a.astype(np.float64)).sum())
        out[key + ".sample"] = a.reshape(-1)[:: max(1, a.size // 257)][:257].copy()


def run_config(dreamer_mod, name: str, d: synth.Dims, full: bool, seed: int = 0, **over):
    out = {}
    P = synth.make_params(d, seed)
    batch = synth.make_batch(d, seed)
    tb = {k: torch.from_numpy(v) for k, v in batch.items()}

    # ---- piecewise: R-enc, R1, R3 on the initial weights ------------------------------------
    agent = build_agent(dreamer_mod, d, P, **over)
    ns = synth.NoiseStream(seed)
    with Inject(ns):
        emb = agent.encoder(tb["observations"][1:])
        beliefs, prior_states, prior_params, post_states, post_params = agent.transition_model(
            torch.zeros(d.B, d.S), tb["actions"][:-1], torch.zeros(d.B, d.Be), emb, tb["nonterminals"][:-1])
        obs_loss = agent._observation_loss(beliefs, post_states, tb["observations"][1:])
        rew_loss = agent._reward_loss(beliefs, post_states, tb["rewards"][:-1])
        kl = agent._kl_loss(post_params, prior_params)
        # R4-R6 on the same weights (imagination from detached posteriors)
        img_b, img_s, img_params, ent = agent.imagine_ahead(post_states.detach(), beliefs.detach())
        img_r = agent.reward_model(img_b, img_s)
        img_v = agent.critic_target(img_b, img_s)
        ret = dreamer_mod.lambda_return(img_r, img_v, bootstrap=img_v[-1], discount=agent.discount,
                                        lambda_=agent.disclam)
    # the stream must have been consumed in exactly make_noise order
    ref_calls = list(ns.calls)
    ns3 = synth.NoiseStream(seed)
    for t in range(d.T):
        ns3.normal((d.B, d.S)); ns3.normal((d.B, d.S))
    for t in range(d.Hm):
        ns3.normal((d.N, d.A)); ns3.normal((d.n_entropy, d.N, d.A)); ns3.normal((d.N, d.S))
    assert ref_calls == ns3.calls, "reference RNG call order differs from synth.make_noise"

    piece = dict(embeddings=emb, beliefs=beliefs, prior_states=prior_states, prior_means=prior_params[0],
                 prior_stds=prior_params[1], posterior_states=post_states, posterior_means=post_params[0],
                 posterior_stds=post_params[1], observation_loss=obs_loss, reward_loss=rew_loss, kl_loss=kl,
                 imged_beliefs=img_b, imged_states=img_s, imged_prior_means=img_params[0],
                 imged_prior_stds=img_params[1], action_entropy=ent, imged_reward=img_r, value_pred=img_v,
                 returns=ret)
    for k, v in piece.items():
        store(out, f"piece.{k}", t2n(v), full)

    # ---- kl_balance == -1 branch (dreamer.py:122-128) ---------------------------------------
    agent.kl_balance = -1
    out["piece.kl_loss_sum_branch"] = t2n(agent._kl_loss(post_params, prior_params))
    agent.kl_balance = ref_params(d, **over)["kl_balance"]

    # ---- whole train_step x2, fresh agent ----------------------------------------------------
    agent = build_agent(dreamer_mod, d, P, **over)
    agent.buffer.sample = lambda n, L: [tb["observations"], tb["actions"], tb["rewards"], tb["nonterminals"]]
    norms = []
    orig_clip = torch.nn.utils.clip_grad_norm_

    def rec_clip(params, max_norm, norm_type=2):
        r = orig_clip(params, max_norm, norm_type=norm_type)
        norms.append(float(r))
        return r

    torch.nn.utils.clip_grad_norm_ = rec_clip
    try:
        for step in range(2):
            with Inject(synth.NoiseStream(seed + step)):
                logs = agent.train_step()
            if step == 0:
                agent.update_critic()      # exercised between steps (main.py:110-112)
            for k, v in logs.items():
                out[f"step{step}.log.{k}"] = np.array(float(v), dtype=np.float64)     # (discount_loss is logged as a tensor, :290)
            out[f"step{step}.grad_norms"] = np.array(norms[-3:], dtype=np.float64)
            for mod in ("transition_model", "observation_model", "reward_model", "encoder", "actor", "critic",
                        "critic_target") + (("discount_model",) if d.use_discount else ()):
                for k, p in getattr(agent, mod).state_dict().items():
                    a = t2n(p)
                    store(out, f"step{step}.param.{mod}.{k}", a, full)
                if mod == "critic_target":
                    continue
                for k, p in getattr(agent, mod).named_parameters():
                    g = t2n(p.grad)           # gradients after clipping (dreamer.py:301,364,388)
                    store(out, f"step{step}.grad.{mod}.{k}", g, full)
    finally:
        torch.nn.utils.clip_grad_norm_ = orig_clip

    # fingerprints of the synthetic inputs, so a drifted numpy stream is detected, not mis-compared
    out["fingerprint.params"] = np.array(sum(float(np.abs(v.astype(np.float64)).sum()) for sd in P.values()
                                             for v in sd.values()))
    out["fingerprint.batch"] = np.array(sum(float(np.abs(v.astype(np.float64)).sum()) for v in batch.values()))
    nz = synth.make_noise(d, seed)
    out["fingerprint.noise"] = np.array(sum(float(np.abs(v.astype(np.float64)).sum()) for v in nz.values()))
    path = os.path.join(ROOT, "tests", "golden", f"{name}.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path) / 1024:.1f} KiB")


def run_config_categorical(dreamer_mod, name: str, d: synth.Dims, seed: int, **over):
    """latent_distribution="Categorical": the reference's own Dreamer code (TransitionModel.forward, _kl_loss Categorical
    branch, imagine_ahead, train_step x2) under CategoricalShims.  The tiny cases are stored in full, the 32 x 32 ones as
    sums + strided samples (`store`)."""
    full = d.S <= 64 and not d.pixel
    assert d.categorical
    out = {}
    P = synth.make_params(d, seed)
    batch = synth.make_batch(d, seed)
    tb = {k: torch.from_numpy(v) for k, v in batch.items()}
    agent = build_agent(dreamer_mod, d, P, **over)
    assert agent.state_size == d.S and agent.latent_distribution == "Categorical"
    ns = synth.NoiseStream(seed)
    with Inject(ns), CategoricalShims(ns) as shims:
        emb = agent.encoder(tb["observations"][1:])
        beliefs, prior_states, prior_params, post_states, post_params = agent.transition_model(
            torch.zeros(d.B, d.S), tb["actions"][:-1], torch.zeros(d.B, d.Be), emb, tb["nonterminals"][:-1])
        obs_loss = agent._observation_loss(beliefs, post_states, tb["observations"][1:])
        rew_loss = agent._reward_loss(beliefs, post_states, tb["rewards"][:-1])
        kl = agent._kl_loss(post_params, prior_params)
        img_b, img_s, img_params, ent = agent.imagine_ahead(post_states.detach(), beliefs.detach())
        img_r = agent.reward_model(img_b, img_s)
        img_v = agent.critic_target(img_b, img_s)
        ret = dreamer_mod.lambda_return(img_r, img_v, bootstrap=img_v[-1], discount=agent.discount, lambda_=agent.disclam)
        assert shims.checked == 3
    # the stream must have been consumed in exactly synth.make_noise order
    ns3 = synth.NoiseStream(seed)
    for t in range(d.T):
        ns3.exponential((d.B * d.cat_D, d.cat_C)); ns3.exponential((d.B * d.cat_D, d.cat_C))
    for t in range(d.Hm):
        ns3.normal((d.N, d.A)); ns3.normal((d.n_entropy, d.N, d.A)); ns3.exponential((d.N * d.cat_D, d.cat_C))
    assert ns.calls == ns3.calls, "reference RNG call order differs from synth.make_noise (Categorical)"
    assert len(prior_params) == 1 and len(post_params) == 1 and len(img_params) == 1
    piece = dict(embeddings=emb, beliefs=beliefs, prior_states=prior_states, prior_logits=prior_params[0],
                 posterior_states=post_states, posterior_logits=post_params[0], observation_loss=obs_loss,
                 reward_loss=rew_loss, kl_loss=kl, imged_beliefs=img_b, imged_states=img_s,
                 imged_prior_logits=img_params[0], action_entropy=ent, imged_reward=img_r, value_pred=img_v, returns=ret)
    for k, v in piece.items():
        store(out, f"piece.{k}", t2n(v), full)
    agent.kl_balance = -1
    out["piece.kl_loss_sum_branch"] = t2n(agent._kl_loss(post_params, prior_params))

    agent = build_agent(dreamer_mod, d, P, **over)
    agent.buffer.sample = lambda n, L: [tb["observations"], tb["actions"], tb["rewards"], tb["nonterminals"]]
    norms = []
    orig_clip = torch.nn.utils.clip_grad_norm_

    def rec_clip(params, max_norm, norm_type=2):
        r = orig_clip(params, max_norm, norm_type=norm_type)
        norms.append(float(r))
        return r

    torch.nn.utils.clip_grad_norm_ = rec_clip
    try:
        for step in range(2):
            ns = synth.NoiseStream(seed + step)
            with Inject(ns), CategoricalShims(ns):
                logs = agent.train_step()
            if step == 0:
                agent.update_critic()
            for k, v in logs.items():
                out[f"step{step}.log.{k}"] = np.array(v, dtype=np.float64)
            out[f"step{step}.grad_norms"] = np.array(norms[-3:], dtype=np.float64)
            for mod in ("transition_model", "observation_model", "reward_model", "encoder", "actor", "critic",
                        "critic_target"):
                for k, p in getattr(agent, mod).state_dict().items():
                    store(out, f"step{step}.param.{mod}.{k}", t2n(p), full)
                if mod == "critic_target":
                    continue
                for k, p in getattr(agent, mod).named_parameters():
                    store(out, f"step{step}.grad.{mod}.{k}", t2n(p.grad), full)
    finally:
        torch.nn.utils.clip_grad_norm_ = orig_clip
    out["fingerprint.params"] = np.array(sum(float(np.abs(v.astype(np.float64)).sum()) for sd in P.values()
                                             for v in sd.values()))
    out["fingerprint.batch"] = np.array(sum(float(np.abs(v.astype(np.float64)).sum()) for v in batch.values()))
    nz = synth.make_noise(d, seed)
    out["fingerprint.noise"] = np.array(sum(float(np.abs(v.astype(np.float64)).sum()) for v in nz.values()))
    path = os.path.join(ROOT, "tests", "golden", f"{name}.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path) / 1024:.1f} KiB")


CATEGORICAL_RUNS = {        # name -> (Dims, seed, overrides); mirrored by tests/helpers.py CAT_CASES
    "cat_tiny": (synth.CAT_TINY, 51, dict(free_nats=0.0)),                 # un-saturated clamp: KL gradients flow
    "cat_tiny_klsum": (synth.CAT_TINY, 52, dict(kl_balance=-1, free_nats=0.01)),
    "cat_32": (synth.CAT_32, 53, dict(free_nats=0.0)),                     # the reference's 32 x 32 latents, batch 18
    "cat_32_v2": (synth.CAT_32, 54, dict()),                               # default free_nats=3 (clamp saturated)
    # BASELINE configs[4] as stated: 64x64 pixel observations + Categorical latents (src/models.py:319-362 with
    # state_size = D*C, src/planet.py:56-57), ragged 3 x 5 factors and the reference's 32 x 32 with A = 17
    "cat_pixel_tiny": (synth.CAT_PIXEL_TINY, 55, dict(free_nats=0.0)),
    "cat_pixel_32": (synth.CAT_PIXEL_32, 56, dict()),
}


def run_replay(memory_mod):
    """R0: ExperienceReplay.sample index/gather semantics (src/memory.py:51-104)."""
    d = synth.TINY
    rows = 64
    rep = synth.make_replay(d, rows=rows, seed=3)
    out = {}
    for case, (idx, full) in {"partial": (40, False), "wrapped": (17, True)}.items():
        buf = memory_mod.ExperienceReplay(rows, d.A, 5, False, d.O, torch.device("cpu"))
        buf.observations[:] = rep["observations"]; buf.actions[:] = rep["actions"]
        buf.rewards[:] = rep["rewards"]; buf.nonterminals[:] = rep["nonterminals"]
        buf.idx, buf.full = idx, full
        np.random.seed(11)
        o, a, r, n = buf.sample(6, 7)
        out[f"{case}.observations"] = t2n(o); out[f"{case}.actions"] = t2n(a)
        out[f"{case}.rewards"] = t2n(r); out[f"{case}.nonterminals"] = t2n(n)
    path = os.path.join(ROOT, "tests", "golden", "replay.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}")


def run_pixel_preprocess():
    """R0 (pixels): preprocess_observation_ (src/utils.py:299-317) on uint8 frames with injected uniform noise."""
    import utils as ref_utils
    rng = np.random.Generator(np.random.PCG64(77))
    u8 = rng.integers(0, 256, size=(6, 3, 64, 64), dtype=np.uint8)
    noise = rng.random((6, 3, 64, 64), dtype=np.float32)
    out = {"u8": u8, "noise": noise}
    for bits in (5, 8, 3):
        x = torch.as_tensor(u8.astype(np.float32))
        orig = torch.rand_like
        torch.rand_like = lambda t, **kw: torch.from_numpy(noise.copy())
        try:
            ref_utils.preprocess_observation_(x, bits)
        finally:
            torch.rand_like = orig
        out[f"out{bits}"] = x.numpy().copy()
    path = os.path.join(ROOT, "tests", "golden", "pixel_preprocess.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}")


def run_planner(dreamer_mod, name: str, d: synth.Dims, B: int, horizon: int, iters: int, candidates: int, top: int,
                seed: int, full: bool):
    """MPCPlanner.forward (src/planner.py:28-90) with the reference's transition / reward modules on synthetic
    weights; per-iteration candidate returns are captured with a forward hook on the reward model."""
    import planner as ref_planner
    P = synth.make_params(d, seed)
    agent = build_agent(dreamer_mod, d, P)
    mpc = ref_planner.MPCPlanner(d.A, horizon, iters, candidates, top, agent.transition_model, agent.reward_model)
    rng = np.random.Generator(np.random.PCG64(seed + 500))
    belief = rng.standard_normal((B, d.Be), dtype=np.float32) * 0.5
    state = rng.standard_normal((B, d.S), dtype=np.float32)
    rets = []
    hook = agent.reward_model.register_forward_hook(
        lambda m, i, o: rets.append(o.detach().view(horizon, -1).sum(dim=0).numpy().copy()))
    ns = synth.NoiseStream(seed)
    with torch.no_grad(), Inject(ns):
        action = mpc(torch.from_numpy(belief), torch.from_numpy(state))
    hook.remove()
    ns2 = synth.NoiseStream(seed)
    for it in range(iters):
        ns2.normal((horizon, B, candidates, d.A))
        for t in range(horizon):
            ns2.normal((B * candidates, d.S))
    assert ns.calls == ns2.calls, "reference planner RNG order differs from synth.make_planner_noise"
    out = {"belief": belief, "state": state, "action": t2n(action),
           "meta": np.array([B, horizon, iters, candidates, top, seed])}
    for it, r in enumerate(rets):
        store(out, f"returns{it}", r, full)
    path = os.path.join(ROOT, "tests", "golden", f"{name}.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path) / 1024:.1f} KiB")


def run_planet(name: str, d: synth.Dims, seed: int):
    """Planet.train_step (src/planet.py:310-368) x2: dynamics learning with the summed free-nats KL."""
    import planet as ref_planet
    P = synth.make_params(d, seed)
    batch = synth.make_batch(d, seed)
    tb = {k: torch.from_numpy(v) for k, v in batch.items()}
    params = ref_params(d, free_nats=0.05)
    params["algorithm"] = "planet"
    agent = ref_planet.Planet(params, FakeEnv(d))
    for mod in ("transition_model", "observation_model", "reward_model", "encoder"):
        getattr(agent, mod).load_state_dict({k: torch.from_numpy(v.copy()) for k, v in P[mod].items()})
    agent.buffer.sample = lambda n, L: [tb["observations"], tb["actions"], tb["rewards"], tb["nonterminals"]]
    out = {}
    for step in range(2):
        with Inject(synth.NoiseStream(seed + step)):
            logs = agent.train_step()
        for k, v in logs.items():
            out[f"step{step}.log.{k}"] = np.array(v, dtype=np.float64)
        for mod in ("transition_model", "observation_model", "reward_model", "encoder"):
            for k, p in getattr(agent, mod).state_dict().items():
                out[f"step{step}.param.{mod}.{k}"] = t2n(p)
            for k, p in getattr(agent, mod).named_parameters():
                out[f"step{step}.grad.{mod}.{k}"] = t2n(p.grad)
    path = os.path.join(ROOT, "tests", "golden", f"{name}.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path) / 1024:.1f} KiB")


def run_categorical(dreamer_mod):
    """R2c pieces that are callable in the reference: CategoricalBeliefModel.forward
    (src/models.py:101-117) with its autograd gradients, and Dreamer._kl_loss / _get_dist, Categorical branch
    (src/dreamer.py:92-146), with gradients w.r.t. both logits."""
    import models as ref_models
    out = {}
    for name, (rows, inp, hid, D, C, seed) in synth.CATEGORICAL_CASES.items():
        c = synth.make_categorical_case(rows, inp, hid, D, C, seed)
        m = ref_models.CategoricalBeliefModel(inp, hid, D, C, "ELU")
        m.load_state_dict({k: torch.from_numpy(c[k].copy()) for k in ("model.0.weight", "model.0.bias", "model.2.weight",
                                                                      "model.2.bias")})
        x = torch.from_numpy(c["x"].copy()).requires_grad_(True)
        # The sampling noise is drawn inside ATen (torch.multinomial's single-draw path: q ~ Exp(1) per class,
        # argmax(probs / q)), out of reach of a Python patch: draw the same stream first, store it, re-seed, run.
        torch.manual_seed(seed)
        q = torch.empty(rows * D, C).exponential_(1)
        torch.manual_seed(seed)
        state, (logits,) = m(x)
        with torch.no_grad():
            pr = torch.softmax(logits - logits.logsumexp(-1, keepdim=True), -1).reshape(-1, C)
            assert torch.equal((pr / q).argmax(-1), state.detach().reshape(-1, C).argmax(-1)), \
                "the stored draws are not the ones the reference's sampler consumed"
        out[f"{name}.q"] = q.numpy().reshape(rows, D, C).copy()
        loss = (state * torch.from_numpy(c["g_state"])).sum() + (logits * torch.from_numpy(c["g_logits"])).sum()
        loss.backward()
        out[f"{name}.state"] = t2n(state)
        out[f"{name}.logits"] = t2n(logits)
        out[f"{name}.dx"] = t2n(x.grad)
        for k, p in m.named_parameters():
            store(out, f"{name}.grad.{k}", t2n(p.grad), full=p.numel() <= 20000)
        # KL between these logits (posterior role) and a second set (prior role), every branch of _kl_loss
        for tag, bal, fn in (("bal_clamped", 0.8, 3.0), ("bal_free", 0.8, 0.0), ("sum_mixed", -1, None)):
            ql = logits.detach().reshape(1, rows, D, C).clone().requires_grad_(True)
            pl = torch.from_numpy(c["other_logits"].copy()).reshape(1, rows, D, C).requires_grad_(True)
            if fn is None:      # sum form: threshold BETWEEN the two middle rows, so both sides of the max occur
                from torch.distributions import OneHotCategoricalStraightThrough as OH, kl_divergence
                srt = kl_divergence(OH(logits=ql.detach()), OH(logits=pl.detach())).sum(dim=2).reshape(-1).sort().values
                fn = float(0.5 * (srt[rows // 2 - 1] + srt[rows // 2]))
            ns = types.SimpleNamespace(latent_distribution="Categorical", kl_balance=bal,
                                       free_nats=torch.full((1,), fn))
            ns._get_dist = types.MethodType(dreamer_mod.Dreamer._get_dist, ns)
            kl = dreamer_mod.Dreamer._kl_loss(ns, (ql,), (pl,))
            kl.sum().backward()
            out[f"{name}.kl.{tag}"] = t2n(kl)
            out[f"{name}.kl.{tag}.free_nats"] = np.array(fn, dtype=np.float64)
            out[f"{name}.kl.{tag}.dpost"] = t2n(ql.grad)
            out[f"{name}.kl.{tag}.dprior"] = t2n(pl.grad)
    path = os.path.join(ROOT, "tests", "golden", "categorical.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path) / 1024:.1f} KiB")


def run_action_mode(dreamer_mod, d: synth.Dims, seed: int):
    """ActorModel.forward (src/models.py:506-517) and Dreamer.get_action(deterministic=True) (src/dreamer.py:429-444)."""
    P = synth.make_params(d, seed)
    agent = build_agent(dreamer_mod, d, P)
    rng = np.random.Generator(np.random.PCG64(seed + 900))
    N = 9
    belief = (0.5 * rng.standard_normal((N, d.Be), dtype=np.float32)).astype(np.float32)
    state = rng.standard_normal((N, d.S), dtype=np.float32)
    ns = synth.NoiseStream(seed)
    with torch.no_grad(), Inject(ns):
        mean, std = agent.actor(torch.from_numpy(belief), torch.from_numpy(state))
        action, entropy = agent.get_action(torch.from_numpy(belief), torch.from_numpy(state), deterministic=True)
    assert ns.calls == [(d.n_entropy, N, d.A), (d.n_entropy, N, d.A)], ns.calls
    out = {"belief": belief, "state": state, "mean": t2n(mean), "std": t2n(std), "action": t2n(action),
           "entropy": t2n(entropy)}
    path = os.path.join(ROOT, "tests", "golden", "action_mode.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}")


def run_act(dreamer_mod):
    """Planet.update_belief_and_act (src/planet.py:370-403) through Dreamer.get_action (src/dreamer.py:429-444): three
    consecutive calls (the carried belief / posterior state / action feed the next call) at B=1 (collection, explore=True)
    and B=10 (evaluation with an EnvBatcher-like env, explore=False), config-2 model size, injected noise.
    RNG order per call: prior (B,S), posterior (B,S), action (B,A), entropy (100,B,A), [explore: (B,A)]."""
    import env as ref_env
    d = synth.CONFIG2
    out = {}
    for name, B, explore, seed in (("b1_explore", 1, True, 31), ("b10_eval", 10, False, 32)):
        P = synth.make_params(d, seed)
        agent = build_agent(dreamer_mod, d, P)
        rng = np.random.Generator(np.random.PCG64(seed + 300))
        obs_seq = rng.standard_normal((3, B, d.O), dtype=np.float32)

        class Env(ref_env.EnvBatcher if B > 1 else object):      # isinstance(env, EnvBatcher) picks action vs action[0]
            def __init__(self):
                self.got = []

            def step(self, a):
                self.got.append(t2n(a))
                return None, 0.0, False

        env = Env()
        belief, state = torch.zeros(B, d.Be), torch.zeros(B, d.S)
        action = torch.zeros(B, d.A)
        ns = synth.NoiseStream(seed)
        with torch.no_grad(), Inject(ns):
            for i in range(3):
                belief, state, action, _, _, _ = agent.update_belief_and_act(env, belief, state, action,
                                                                            torch.from_numpy(obs_seq[i]), explore=explore)
                out[f"{name}.belief{i}"], out[f"{name}.state{i}"], out[f"{name}.action{i}"] = t2n(belief), t2n(state), t2n(action)
                out[f"{name}.env_action{i}"] = env.got[-1]
        want = []
        for i in range(3):
            want += [(B, d.S), (B, d.S), (B, d.A), (d.n_entropy, B, d.A)] + ([(B, d.A)] if explore else [])
        assert ns.calls == want, ns.calls
        out[f"{name}.obs"] = obs_seq
        out[f"{name}.meta"] = np.array([B, int(explore), seed])
    path = os.path.join(ROOT, "tests", "golden", "act.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path) / 1024:.1f} KiB")


def main_planner(dreamer_mod):
    run_planner(dreamer_mod, "planner_tiny", synth.TINY, B=2, horizon=5, iters=4, candidates=64, top=8, seed=6, full=True)
    # the reference's defaults (conf/config.yaml:31,63-66) at the config-2 model size, one environment
    run_planner(dreamer_mod, "planner_config2", synth.CONFIG2, B=1, horizon=15, iters=10, candidates=1000, top=100,
                seed=7, full=False)
    run_planet("tiny_planet", synth.TINY, seed=8)
    run_categorical(dreamer_mod)
    run_action_mode(dreamer_mod, synth.SMALL, seed=12)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    dreamer_mod, memory_mod = _import_reference()
    run_replay(memory_mod)
    run_config(dreamer_mod, "tiny", synth.TINY, full=True)
    run_config(dreamer_mod, "small", synth.SMALL, full=True, seed=1)
    # the sum-KL branch end to end (kl_balance = -1) and an unsaturated free-nats setting
    run_config(dreamer_mod, "tiny_klsum", synth.TINY, full=True, seed=2, kl_balance=-1, free_nats=0.05)
    run_config(dreamer_mod, "tiny_freenats0", synth.TINY, full=True, seed=3, free_nats=0.0)
    run_config(dreamer_mod, "tiny_pixel", synth.TINY_PIXEL, full=False, seed=4)
    run_config(dreamer_mod, "tiny_pixel_lin", synth.TINY_PIXEL_LIN, full=False, seed=5)
    run_pixel_preprocess()
    run_config(dreamer_mod, "config1", synth.CONFIG1, full=False)
    run_config(dreamer_mod, "config2", synth.CONFIG2, full=False)
    run_config(dreamer_mod, "config3", synth.CONFIG3, full=False, seed=6)      # BASELINE configs[2], full size (~1 min)
    run_config(dreamer_mod, "tiny_discount", synth.TINY_DISCOUNT, full=True, seed=9)   # use_discount=True
    main_planner(dreamer_mod)
    run_act(dreamer_mod)
    for name, (dd, sd_, ov) in CATEGORICAL_RUNS.items():
        run_config_categorical(dreamer_mod, name, dd, sd_, **ov)


if __name__ == "__main__":
    if "--only" in sys.argv:               # regenerate single files: --only config3 act ...
        torch.manual_seed(0)
        torch.set_num_threads(8)
        dm, mm = _import_reference()
        for what in sys.argv[sys.argv.index("--only") + 1:]:
            if what in CATEGORICAL_RUNS:
                dd, sd_, ov = CATEGORICAL_RUNS[what]
                run_config_categorical(dm, what, dd, sd_, **ov)
                continue
            if what == "categorical_scan":
                for name, (dd, sd_, ov) in CATEGORICAL_RUNS.items():
                    run_config_categorical(dm, name, dd, sd_, **ov)
                continue
            {"config3": lambda: run_config(dm, "config3", synth.CONFIG3, full=False, seed=6),
             "discount": lambda: run_config(dm, "tiny_discount", synth.TINY_DISCOUNT, full=True, seed=9),
             "act": lambda: run_act(dm)}[what]()
    elif "--planner-only" in sys.argv:       # regenerate only the planner / PlaNet vectors
        torch.manual_seed(0)
        torch.set_num_threads(8)
        main_planner(_import_reference()[0])
    else:
        main()
