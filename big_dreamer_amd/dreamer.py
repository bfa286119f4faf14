"""Dreamer agent with the reference's call surface (src/dreamer.py, src/planet.py) on the HIP engine.

Kept surface: ``Dreamer(params, env)``, ``train_step() -> Dict[str, float]``,
``imagine_ahead``, ``get_action``, ``update_critic``, ``update_belief_and_act``, ``buffer.append/sample``,
``eval()/train()``, ``observation_model(h, s)``, module attributes with reference ``state_dict`` names, ``load``
of the reference's checkpoint dict (src/planet.py:103-114), and the module function ``lambda_return``.
"""
from __future__ import annotations

import os
from typing import Any, Dict, Optional, Tuple

import torch
from torch import Tensor

from . import _cabi as cabi
from .engine import DreamerEngine
from .memory import ExperienceReplay
from .models import ActorModel, CnnImageEncoder, DenseModel, ObservationModel, TransitionModel, encoder_for
from .synth import Dims


def _hp_from_params(params: Dict[str, Any]) -> Dict[str, float]:
    ac = params["ActorCritic"]
    return dict(
        kl_balance=params["kl_balance"], kl_loss_weight=params["kl_loss_weight"], free_nats=params["free_nats"],
        grad_clip_norm=params["grad_clip_norm"], discount=params["discount"], disclam=params["disclam"],
        model_learning_rate=params["model_learning_rate"], actor_learning_rate=ac["actor_learning_rate"],
        value_learning_rate=ac["value_learning_rate"], adam_epsilon=params["adam_epsilon"],
        weight_decay=params["weight_decay"], entropy_weight=ac["entropy_weight"], polyak_avg=ac["polyak_avg"],
        discount_weight=params.get("discount_weight", 5.0))


class Dreamer:
    """Drop-in for the reference's ``Dreamer``: state or 64x64 pixel observations, Gaussian or Categorical latents
    (``latent_distribution``; Categorical: state_size = dimensions * classes, src/planet.py:56-57)."""

    def __init__(self, params: Dict[str, Any], env, device: Optional[str] = None, world_size: int = 1,
                 process_group=None):
        self.latent_distribution = params.get("latent_distribution", "Gaussian")
        if self.latent_distribution not in ("Gaussian", "Categorical"):
            raise NotImplementedError(f"{self.latent_distribution}  is yet yet implemented")     # as src/dreamer.py:108
        if params["ActorCritic"]["gradient_mixing"] != -1:
            raise NotImplementedError("gradient_mixing not yet implemented ")      # as src/dreamer.py:339
        self.use_discount = bool(params.get("use_discount", False))
        if params.get("disable_cuda", False) or not torch.cuda.is_available():
            raise RuntimeError("big_dreamer_amd runs on MI355X only: there is no CPU path (disable_cuda=True is the "
                               "reference's own CPU mode)")
        self.env = env
        self.params = params
        self.device = torch.device(device or f"cuda:{torch.cuda.current_device()}")
        self.belief_size, self.state_size = params["belief_size"], params["state_size"]
        cat_D = cat_C = 0
        if self.latent_distribution == "Categorical":
            cat_D, cat_C = int(params["discrete_latent_dimensions"]), int(params["discrete_latent_classes"])
            self.state_size = cat_D * cat_C                                                    # src/planet.py:56-57
        self.action_size, self.hidden_size = env.action_size, params["hidden_size"]
        self.embedding_size = params["embedding_size"]
        self.batch_size, self.seq_len = params["batch_size"], params["seq_len"]
        self.planning_horizon = params["planning_horizon"]
        self.action_noise = params["action_noise"]
        self.action_repeat = params["action_repeat"]
        self.seed_steps = params["seed_steps"]
        self.pixel_observation = bool(params.get("pixel_observation", False))
        obs_size = 3 * 64 * 64 if self.pixel_observation else env.observation_size
        self.dims = Dims(B=self.batch_size, L=self.seq_len, H=self.planning_horizon, Be=self.belief_size,
                         S=self.state_size, Hd=self.hidden_size, E=self.embedding_size, A=self.action_size,
                         O=obs_size, pixel=self.pixel_observation, cat_D=cat_D, cat_C=cat_C,
                         use_discount=self.use_discount)
        self.engine = DreamerEngine(self.dims, _hp_from_params(params), self.device, world_size=world_size,
                                    process_group=process_group)
        e = self.engine
        # PyTorch-default initialisation happens in the nn.Modules below (their parameters alias the engine's
        # flat buffers), so seeding torch before construction reproduces a run (src/main.py:56-58).
        init = {}
        feat = self.belief_size + self.state_size
        px, E = self.pixel_observation, self.embedding_size
        for mod, build in (
            ("transition_model", lambda: _ref_init_transition(self.dims)),
            ("observation_model", (lambda: _ref_init_decoder(feat, E)) if px else
             (lambda: _ref_init_dense(feat, self.hidden_size, env.observation_size))),
            ("reward_model", lambda: _ref_init_dense(feat, self.hidden_size, 1)),
            ("encoder", (lambda: _ref_init_cnn(E)) if px else
             (lambda: _ref_init_dense(env.observation_size, self.hidden_size, self.embedding_size))),
            ("actor", lambda: _ref_init_dense(feat, self.hidden_size, 2 * self.action_size)),
            ("critic", lambda: _ref_init_dense(feat, self.hidden_size, 1)),
        ) + ((("discount_model", lambda: _ref_init_dense(feat, self.hidden_size, 1)),) if self.use_discount else ()):
            init[mod] = {k: v.detach().numpy() for k, v in build().state_dict().items()}
        init["critic_target"] = init["critic"]                # copy.deepcopy(self.critic), src/dreamer.py:50
        e.load_params(init)
        for g in ("model", "actor", "critic", "critic_target"):
            e.pack(g)
        self.transition_model = TransitionModel(self.belief_size, self.state_size, self.action_size, self.hidden_size,
                                                self.embedding_size, latent_distribution=self.latent_distribution,
                                                discrete_latent_dimensions=cat_D or 32, discrete_latent_classes=cat_C or 32,
                                                engine=e)
        if px:
            self.observation_model = ObservationModel(self.belief_size, self.state_size, E, engine=e)
            self.encoder = CnnImageEncoder(E, engine=e)
        else:
            self.observation_model = DenseModel(feat, self.hidden_size, env.observation_size, engine=e,
                                                module="observation_model", prefix="obs")
            self.encoder = encoder_for(e, env.observation_size, self.hidden_size, self.embedding_size)
        self.reward_model = DenseModel(feat, self.hidden_size, 1, engine=e, module="reward_model", prefix="rew")
        self.actor = ActorModel(self.belief_size, self.state_size, self.hidden_size, self.action_size, engine=e)
        self.critic = DenseModel(feat, self.hidden_size, 1, engine=e, module="critic", prefix="cri")
        self.critic_target = DenseModel(feat, self.hidden_size, 1, engine=e, module="critic_target", prefix="tgt")
        if self.use_discount:       # src/dreamer.py:80-85
            self.discount_model = DenseModel(feat, self.hidden_size, 1, engine=e, module="discount_model", prefix="dsc")
        self.buffer = ExperienceReplay(params["experience_size"], env.action_size, params["bit_depth"], px,
                                       env.observation_size, self.device)
        self.load(params)

    # ---------------------------------------------------------------------------------------- checkpoints
    def load(self, params: Dict[str, Any]) -> None:
        """src/planet.py:103-114: load the reference's checkpoint dict -- the four world-model state dicts AND
        ``model_optimizer`` (Adam moments and step count, so a resumed run continues the reference's bias correction) --
        with a loader that executes nothing from the file.  Checkpoints written by ``save`` below also restore the
        actor, the critic and its target and their optimisers (the reference never saves those)."""
        path = params.get("models", "")
        if path and os.path.exists(path):
            d = torch.load(path, map_location="cpu", weights_only=True)
            for key in ("transition_model", "observation_model", "reward_model", "encoder"):
                getattr(self, key).load_state_dict(d[key])
            if "model_optimizer" in d:
                self.engine.load_optimizer_state_dict("model", d["model_optimizer"])
            for key in ("actor", "critic", "critic_target") + (("discount_model",) if self.use_discount else ()):
                if key in d:
                    getattr(self, key).load_state_dict(d[key])
            if self.use_discount and "discount_model" not in d and "model_optimizer" in d and d["model_optimizer"]["state"]:
                import warnings      # (a checkpoint in the reference's five-key layout: that is what the reference does too)
                warnings.warn("use_discount=True: the checkpoint carries Adam moments for the discount head (it is part of "
                              "the world-model optimiser, src/dreamer.py:167-169) but not its weights, which stay freshly "
                              "initialised")
            for key, group in (("actor_optimizer", "actor"), ("value_optimizer", "critic")):
                if key in d:
                    self.engine.load_optimizer_state_dict(group, d[key])

    def save(self, path: str) -> None:
        """Checkpoint in the layout ``Planet.load`` reads (src/planet.py:109-114: transition_model, observation_model,
        reward_model, encoder, model_optimizer -- the reference can resume from it), plus actor / critic /
        critic_target, the discount head when ``use_discount`` is set (its parameters are the tail of the world-model
        optimiser, src/dreamer.py:167-169) and the actor / value optimisers (src/dreamer.py:56-67 names) so that this
        framework resumes exactly.
        The reference declares ``save`` (src/base_agent.py:29-34) but never implements it (src/main.py:274 TODO)."""
        e = self.engine
        d = {key: {k: v.detach().cpu().clone().contiguous() for k, v in getattr(self, key).state_dict().items()}
             for key in ("transition_model", "observation_model", "reward_model", "encoder", "actor", "critic",
                         "critic_target") + (("discount_model",) if self.use_discount else ())}
        d["model_optimizer"] = e.optimizer_state_dict("model")
        d["actor_optimizer"] = e.optimizer_state_dict("actor")
        d["value_optimizer"] = e.optimizer_state_dict("critic")
        torch.save(d, path)

    def eval(self) -> None:      # no dropout / batch-norm anywhere on the path
        pass

    def train(self) -> None:
        pass

    # ---------------------------------------------------------------------------------------- replay
    def randomly_initialize_replay_buffer(self) -> Tuple[int, int]:
        """src/planet.py:136-159."""
        total_steps, s = 0, 0
        while total_steps < self.seed_steps:
            done, t = False, 0
            observation = self.env.reset()
            while not done:
                action = self.env.sample_random_action()
                next_observation, reward, done = self.env.step(action)
                self.buffer.append(observation, action, reward, done)
                observation = next_observation
                t += 1
            s += 1
            total_steps += t * self.action_repeat
        self.env.close()
        return total_steps, s

    # ---------------------------------------------------------------------------------------- training
    def train_step(self) -> Dict[str, float]:
        """src/dreamer.py:253-393.  Returns the reference's log dict."""
        obs, actions, rewards, nonterminals = self.buffer.sample(self.batch_size, self.seq_len)
        batch = {"observations": obs, "actions": actions, "rewards": rewards, "nonterminals": nonterminals}
        if os.environ.get("BD_LAZY_LOGS", "1") == "0":       # eager: every call waits for the GPU, like the reference's .item()
            logs = self.engine.train_step(batch)
            return {k: v for k, v in logs.items() if not k.startswith("grad_norm")}
        # Lazy mapping with the reference's keys: values are fetched when a key is first read, so the collect loop's burst
        # `for _ in range(collect_interval): logs = model.train_step()` (src/main.py:105-108) keeps the cross-step
        # pipeline full and only the dict it actually reads waits for the device.
        logs = self.engine.train_step(batch, sync_logs="lazy")
        logs._drop = ("grad_norm",)
        return logs

    def update_critic(self) -> None:
        self.engine.update_critic()

    # ---------------------------------------------------------------------------------------- acting / imagination
    @torch.no_grad()
    def imagine_ahead(self, prev_state: Tensor, prev_belief: Tensor, _noise: Optional[Dict[str, Tensor]] = None):
        """src/dreamer.py:179-237: (seq,batch,S), (seq,batch,Be) -> beliefs (H-1,N,Be), prior_states (H-1,N,S),
        (prior_means, prior_stds), action_entropy (H-1,N)."""
        e, d = self.engine, self.dims
        e.join()
        N = prev_state.shape[0] * prev_state.shape[1]
        start = torch.cat([prev_belief.reshape(N, d.Be), prev_state.reshape(N, d.S)], dim=1).contiguous().float()
        Hm = self.planning_horizon - 1
        noise = _noise or {"action": torch.randn(Hm, N, d.A, device=e.dev),
                           "entropy": torch.randn(Hm, d.n_entropy, N, d.A, device=e.dev),
                           "img_prior": self._state_draw(Hm, N)}
        ifeat, ent, _ = e.imagine(start, N, Hm, noise, save=False, tag="api_")
        f = ifeat.view(Hm, N, d.Be + d.S)
        if d.categorical:
            params = (e._buf["api_iprior_logits"].view(Hm, N, d.cat_D, d.cat_C).clone(),)
        else:
            params = (e._buf["api_iprior_mean"].view(Hm, N, d.S).clone(), e._buf["api_iprior_std"].view(Hm, N, d.S).clone())
        return f[..., :d.Be].clone(), f[..., d.Be:].clone(), params, ent.view(Hm, N).clone()

    def _state_draw(self, steps: int, rows: int) -> Tensor:
        """Noise of one state sample per row: standard normals (Gaussian latents) or the sampler's Exp(1) variates per
        class (Categorical; the one-step get_action path never looks at them)."""
        t = torch.empty(steps, rows, self.dims.S, device=self.engine.dev)
        return t.exponential_() if self.dims.categorical else t.normal_()

    @torch.no_grad()
    def get_action(self, belief: Tensor, state: Tensor, deterministic: bool = False,
                   _noise: Optional[Dict[str, Tensor]] = None) -> Tuple[Tensor, Tensor]:
        """src/dreamer.py:429-444: tanh-Normal sample and its 100-sample entropy estimate."""
        e, d = self.engine, self.dims
        e.join()
        N = belief.shape[0]
        if deterministic:
            # SampleDist.mode (src/models.py:709-723): of n_samples draws, the one with the highest log-density per row;
            # then the entropy estimate on fresh draws (RNG order: mode, entropy).  Never used by the reference loop --
            # a few elementwise torch ops on the actor's (mean, std), entropy from the same kernel as below.
            mean, std = self.actor(belief, state)
            nz = _noise or {}
            eps = nz["mode"].to(e.dev).float() if "mode" in nz else torch.randn(d.n_entropy, N, d.A, device=e.dev)
            sample = torch.tanh(mean.unsqueeze(0) + std.unsqueeze(0) * eps)
            idx = torch.argmax(_tanh_normal_log_prob(sample, mean.unsqueeze(0), std.unsqueeze(0)), dim=0)
            action = torch.gather(sample, 0, idx.reshape(1, N, 1).expand(1, N, d.A)).squeeze(0)
            start = torch.cat([belief, state], dim=1).contiguous().float()
            noise = {"action": torch.zeros(1, N, d.A, device=e.dev),
                     "entropy": (nz["entropy"].to(e.dev).float().reshape(1, d.n_entropy, N, d.A) if "entropy" in nz
                                 else torch.randn(1, d.n_entropy, N, d.A, device=e.dev)),
                     "img_prior": torch.ones(1, N, d.S, device=e.dev)}
            _, ent, _ = e.imagine(start, N, 1, noise, save=False, tag="act_")
            return action, ent.view(N).clone()
        start = torch.cat([belief, state], dim=1).contiguous().float()
        noise = _noise or {"action": torch.randn(1, N, d.A, device=e.dev),
                           "entropy": torch.randn(1, d.n_entropy, N, d.A, device=e.dev),
                           "img_prior": torch.ones(1, N, d.S, device=e.dev)}       # the unused prior sample's draws
        _, ent, act = e.imagine(start, N, 1, noise, save=False, tag="act_")   # one step: actor + sample (+ unused prior)
        return act.view(N, d.A).clone(), ent.view(N).clone()

    @torch.no_grad()
    def update_belief_and_act(self, env, belief, posterior_state, action, observation, explore=False,
                              _noise: Optional[Dict[str, Tensor]] = None):
        """src/planet.py:370-403.  (Data-parallel runs: issues any held-back actor update first -- a collective, call on
        every rank, as the collect loop does.)"""
        self.engine.flush_optimizers()
        # `_noise` (parity tests): the reference's draws in its RNG order -- "prior" (B,S), "post" (B,S), "action" (B,A),
        # "entropy" (100,B,A), and with explore "explore" (B,A)
        nz = _noise
        embedding = self.encoder(observation.to(self.device)).unsqueeze(dim=0)
        belief, _, _, posterior_state, _ = self.transition_model(
            posterior_state, action.unsqueeze(dim=0), belief, embedding,
            _noise=None if nz is None else (nz["prior"].unsqueeze(0), nz["post"].unsqueeze(0)))
        belief, posterior_state = belief.squeeze(dim=0), posterior_state.squeeze(dim=0)
        action, _ = self.get_action(belief, posterior_state, _noise=None if nz is None else {
            "action": nz["action"].unsqueeze(0).contiguous(), "entropy": nz["entropy"].unsqueeze(0).contiguous(),
            "img_prior": torch.ones(1, belief.shape[0], self.state_size, device=self.device)})
        if explore:
            eps = torch.randn_like(action) if nz is None else nz["explore"]
            action = torch.clamp(action + self.action_noise * eps, -1, 1)
        batched = hasattr(env, "n") and hasattr(env, "envs")          # EnvBatcher (src/env.py:343)
        next_observation, reward, done = env.step(action.cpu() if batched else action[0].cpu())
        return belief, posterior_state, action, next_observation, reward, done


class DreamerV2(Dreamer):
    """src/dreamerV2.py:13-25: Dreamer with the V2 KL settings (balanced KL; kl_loss_weight 1.0 is replaced by 0.1);
    ``latent_distribution=Categorical`` selects the 32 x 32 discrete latents (BASELINE configs[4])."""

    def __init__(self, params: Dict[str, Any], env, **kw):
        p = dict(params)
        if p.get("kl_loss_weight") == 1.0:
            p["kl_loss_weight"] = 0.1
        super().__init__(p, env, **kw)
        self.kl_balance = p["kl_balance"]


def _tanh_normal_log_prob(y: Tensor, mean: Tensor, std: Tensor) -> Tensor:
    """Independent(TransformedDistribution(Normal(mean, std), TanhBijector()), 1).log_prob(y) (src/models.py:630-673)."""
    import math
    yc = torch.where(torch.abs(y) <= 1.0, torch.clamp(y, -0.99999997, 0.99999997), y)
    x = 0.5 * torch.log((1 + yc) / (1 - yc))
    ladj = 2.0 * (math.log(2) - x - torch.nn.functional.softplus(-2.0 * x))
    base = -((x - mean) ** 2) / (2 * std ** 2) - torch.log(std) - math.log(math.sqrt(2 * math.pi))
    return (base - ladj).sum(-1)


def lambda_return(imged_reward: Tensor, value_pred: Tensor, bootstrap: Tensor, discount: float = 0.99,
                  lambda_: float = 0.95) -> Tensor:
    """src/dreamer.py:447-471 on the GPU (bd_lambda_return_forward).  The kernel takes the bootstrap from the
    last value row, as the only call site does (``bootstrap=value_pred[-1]``, src/dreamer.py:332)."""
    if not torch.equal(bootstrap, value_pred[-1]):
        raise NotImplementedError("bootstrap must be value_pred[-1] (the reference's only use)")
    Hm = imged_reward.shape[0]
    N = imged_reward[0].numel()
    r, v = imged_reward.contiguous().float(), value_pred.contiguous().float()
    out = torch.empty_like(r)
    cabi.check(cabi.lib.bd_lambda_return_forward(r.data_ptr(), v.data_ptr(), Hm, N, discount, lambda_, out.data_ptr(),
                                                 cabi.stream()))
    return out


# -------------------------------------------------------------------------------------------------
# PyTorch-default initialisation of the reference's modules (shapes: src/models.py, src/utils.py:368-404)
def _ref_init_dense(i: int, h: int, o: int) -> torch.nn.Module:
    from torch import nn
    layers, k = [], i
    for _ in range(4):
        layers += [nn.Linear(k, h), nn.ELU()]
        k = h
    layers += [nn.Linear(k, o), nn.Identity()]
    m = nn.Module()
    m.model = nn.Sequential(*layers)
    return m


def _ref_init_cnn(E: int) -> torch.nn.Module:
    from torch import nn
    m = nn.Module()
    m.model = nn.Sequential(nn.Conv2d(3, 32, 4, 2), nn.ELU(), nn.Conv2d(32, 64, 4, 2), nn.ELU(), nn.Conv2d(64, 128, 4, 2),
                            nn.ELU(), nn.Conv2d(128, 256, 4, 2), nn.ELU(), nn.Flatten(),
                            nn.Identity() if E == 1024 else nn.Linear(1024, E))
    return m


def _ref_init_decoder(feat: int, E: int) -> torch.nn.Module:
    from torch import nn
    m = nn.Module()
    m.decoder = nn.Sequential(nn.Linear(feat, E), nn.Identity(), nn.ConvTranspose2d(E, 128, 5, 2), nn.ELU(),
                              nn.ConvTranspose2d(128, 64, 5, 2), nn.ELU(), nn.ConvTranspose2d(64, 32, 6, 2), nn.ELU(),
                              nn.ConvTranspose2d(32, 3, 6, 2))
    return m


def _ref_init_transition(d: Dims) -> torch.nn.Module:
    from torch import nn
    from .models import CategoricalBeliefHolder, GaussianBeliefHolder
    m = nn.Module()
    m.rnn = nn.GRUCell(d.Be, d.Be)
    m.fc_embed_state_action = nn.Sequential(nn.Linear(d.S + d.A, d.Be), nn.ELU())
    if d.categorical:
        m.belief_prior = CategoricalBeliefHolder(d.Be, d.Hd, d.cat_D, d.cat_C)
        m.belief_posterior = CategoricalBeliefHolder(d.Be + d.E, d.Hd, d.cat_D, d.cat_C)
    else:
        m.belief_prior = GaussianBeliefHolder(d.Be, d.Hd, d.S, 0.1)
        m.belief_posterior = GaussianBeliefHolder(d.Be + d.E, d.Hd, d.S, 0.1)
    return m
