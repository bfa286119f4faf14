"""ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement (plain torch fp32 ops, autograd for the backward passes, exactly like the
reference) of the Dreamer world-model training step of jgsimard/big-dreamer, written from the
formulas of the reference files cited per function.  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import this module; the product package
``big_dreamer_amd`` never does and fails loudly without its HIP extension.

Parity pin: checked against golden vectors produced by importing the reference itself
(``oracle/gen_golden.py`` -> ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``).
The reference has no tests of its own for this path.

Conventions: every tensor is fp32, time-major ``(time, batch, feature)``.  ``P`` is a dict
``module -> {state_dict name -> tensor}`` (see ``big_dreamer_amd.synth.param_shapes``).
Noise is always an explicit input (standard-normal draws, reference order: SURVEY.md R-RNG).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
HALF_LOG_2PI = 0.5 * math.log(2.0 * math.pi)


# ----------------------------------------------------------------------------------------------
# building blocks
# ----------------------------------------------------------------------------------------------
def mlp(x: Tensor, sd: Dict[str, Tensor], n_hidden: int = 4) -> Tensor:
    """DenseModel.model = build_mlp(in, hid, out, n_layers): (Linear, ELU) x n_layers + Linear
    (src/utils.py:368-404, src/models.py:365-408)."""
    for i in range(n_hidden):
        x = F.elu(F.linear(x, sd[f"model.{2 * i}.weight"], sd[f"model.{2 * i}.bias"]))
    return F.linear(x, sd[f"model.{2 * n_hidden}.weight"], sd[f"model.{2 * n_hidden}.bias"])


def cnn_encoder(obs: Tensor, sd: Dict[str, Tensor]) -> Tensor:
    """CnnImageEncoder.forward (src/models.py:527-564): 4 x (Conv2d k4 s2 + ELU), Flatten, Identity | Linear(1024, E)."""
    lead = obs.shape[:-3]
    x = obs.reshape(-1, *obs.shape[-3:])
    for i in range(4):
        x = F.elu(F.conv2d(x, sd[f"model.{2 * i}.weight"], sd[f"model.{2 * i}.bias"], stride=2))
    x = x.flatten(1)
    if "model.9.weight" in sd:
        x = F.linear(x, sd["model.9.weight"], sd["model.9.bias"])
    return x.reshape(*lead, -1)


def cnn_decoder(belief: Tensor, state: Tensor, sd: Dict[str, Tensor]) -> Tensor:
    """ObservationModel.forward (src/models.py:350-362): Linear, reshape (E,1,1), 4 x ConvTranspose2d (ELU between)."""
    lead = belief.shape[:-1]
    x = F.linear(torch.cat([belief, state], dim=-1), sd["decoder.0.weight"], sd["decoder.0.bias"])
    x = x.reshape(-1, x.shape[-1], 1, 1)
    for idx in (2, 4, 6, 8):
        x = F.conv_transpose2d(x, sd[f"decoder.{idx}.weight"], sd[f"decoder.{idx}.bias"], stride=2)
        if idx != 8:
            x = F.elu(x)
    return x.reshape(*lead, 3, 64, 64)


def dense_on_features(belief: Tensor, state: Tensor, sd: Dict[str, Tensor]) -> Tensor:
    """DenseModel.forward(belief, state) = model(cat(belief, state, -1)) (src/models.py:393-408)."""
    return mlp(torch.cat([belief, state], dim=-1), sd)


def gru_cell(x: Tensor, h: Tensor, sd: Dict[str, Tensor]) -> Tensor:
    """nn.GRUCell (src/models.py:149,252): gate order r,z,n;
    n = tanh(W_in x + b_in + r*(W_hn h + b_hn)); h' = (1-z)*n + z*h."""
    gi = F.linear(x, sd["rnn.weight_ih"], sd["rnn.bias_ih"])
    gh = F.linear(h, sd["rnn.weight_hh"], sd["rnn.bias_hh"])
    i_r, i_z, i_n = gi.chunk(3, dim=1)
    h_r, h_z, h_n = gh.chunk(3, dim=1)
    r = torch.sigmoid(i_r + h_r)
    z = torch.sigmoid(i_z + h_z)
    n = torch.tanh(i_n + r * h_n)
    return (1.0 - z) * n + z * h


def embed_state_action(state: Tensor, action: Tensor, sd: Dict[str, Tensor]) -> Tensor:
    """fc_embed_state_action = Linear(S+A, Be) + ELU (src/models.py:152-154,251)."""
    return F.elu(F.linear(torch.cat([state, action], dim=1),
                          sd["fc_embed_state_action.0.weight"], sd["fc_embed_state_action.0.bias"]))


def gaussian_belief(inp: Tensor, sd: Dict[str, Tensor], which: str, eps: Tensor,
                    min_std: float = 0.1) -> Tuple[Tensor, Tensor, Tensor]:
    """GaussianBeliefModel.forward (src/models.py:60-73)."""
    hid = F.elu(F.linear(inp, sd[f"{which}.model.0.weight"], sd[f"{which}.model.0.bias"]))
    out = F.linear(hid, sd[f"{which}.model.2.weight"], sd[f"{which}.model.2.bias"])
    mean, raw = torch.chunk(out, 2, dim=1)
    std = F.softplus(raw) + min_std
    return mean + std * eps, mean, std


# ----------------------------------------------------------------------------------------------
# R1: TransitionModel.forward (src/models.py:191-299)
# ----------------------------------------------------------------------------------------------
def transition_forward(sd: Dict[str, Tensor], init_state: Tensor, actions: Tensor, init_belief: Tensor,
                       embeddings: Optional[Tensor], nonterminals: Optional[Tensor],
                       eps_prior: Tensor, eps_post: Optional[Tensor], cat: Optional[Tuple[int, int]] = None):
    """Returns beliefs, prior_states, (prior_means, prior_stds), posterior_states,
    (posterior_means, posterior_stds); posterior entries are None when embeddings is None
    (src/models.py:296-297).  ``actions``/``embeddings``/``nonterminals`` have T entries.
    ``cat=(D, C)``: latent_distribution="Categorical" -- the params are 1-tuples ``(logits (T,B,D,C),)`` and the noise
    holds the sampler's Exp(1) draws (see transition_forward_categorical)."""
    if cat is not None:
        return transition_forward_categorical(sd, init_state, actions, init_belief, embeddings, nonterminals, eps_prior,
                                              eps_post, cat[0], cat[1])
    T = actions.size(0)
    belief, prior_state, post_state = init_belief, init_state, init_state
    beliefs, priors, pmeans, pstds, posts, qmeans, qstds = [], [], [], [], [], [], []
    for t in range(T):
        state = prior_state if embeddings is None else post_state            # :241
        if nonterminals is not None:
            state = state * nonterminals[t]                                   # :247
        hidden = embed_state_action(state, actions[t], sd)                    # :251
        belief = gru_cell(hidden, belief, sd)                                 # :252
        prior_state, pm, ps = gaussian_belief(belief, sd, "belief_prior", eps_prior[t])   # :256
        beliefs.append(belief); priors.append(prior_state); pmeans.append(pm); pstds.append(ps)
        if embeddings is not None:
            # t_ = t - 1; embeddings[t_ + 1] == embeddings[t]                 # :265-266
            post_in = torch.cat([belief, embeddings[t]], dim=1)
            post_state, qm, qs = gaussian_belief(post_in, sd, "belief_posterior", eps_post[t])
            posts.append(post_state); qmeans.append(qm); qstds.append(qs)
    st = lambda xs: torch.stack(xs, dim=0)
    if embeddings is None:
        return st(beliefs), st(priors), (st(pmeans), st(pstds)), None, None
    return st(beliefs), st(priors), (st(pmeans), st(pstds)), st(posts), (st(qmeans), st(qstds))


def _sub(sd: Dict[str, Tensor], which: str) -> Dict[str, Tensor]:
    return {k[len(which) + 1:]: v for k, v in sd.items() if k.startswith(which + ".")}


def transition_forward_categorical(sd, init_state, actions, init_belief, embeddings, nonterminals, q_prior, q_post,
                                   D: int, C: int):
    """TransitionModel.forward with latent_distribution="Categorical" (src/models.py:191-299, Categorical branches
    :226-228,258-260,269-271,283-295) as the reference INTENDS it: at HEAD the loop stores the 1-tuple ``(logits,)`` in
    the per-step lists (:259-260,270-271), so ``stack`` raises, and ``posterior_params`` loses its tuple (:295).  Here
    the logits are stored and both params are 1-tuples.  The golden vectors for this path come from the reference's own
    code run with exactly those two repairs (oracle/gen_golden.py, ``_categorical_shims``).
    q_prior / q_post: (T, B, D*C) Exp(1) draws of the sampler (categorical_belief)."""
    T = actions.size(0)
    belief, prior_state, post_state = init_belief, init_state, init_state
    beliefs, priors, plog, posts, qlog = [], [], [], [], []
    prior_sd, post_sd = _sub(sd, "belief_prior"), _sub(sd, "belief_posterior")
    for t in range(T):
        state = prior_state if embeddings is None else post_state            # :241
        if nonterminals is not None:
            state = state * nonterminals[t]                                   # :247
        hidden = embed_state_action(state, actions[t], sd)                    # :251
        belief = gru_cell(hidden, belief, sd)                                 # :252
        prior_state, (pl,) = categorical_belief(belief, prior_sd, q_prior[t].reshape(-1, D, C), D, C)   # :256
        beliefs.append(belief); priors.append(prior_state); plog.append(pl)
        if embeddings is not None:
            post_in = torch.cat([belief, embeddings[t]], dim=1)               # :265-266
            post_state, (ql,) = categorical_belief(post_in, post_sd, q_post[t].reshape(-1, D, C), D, C)   # :267
            posts.append(post_state); qlog.append(ql)
    st = lambda xs: torch.stack(xs, dim=0)
    if embeddings is None:
        return st(beliefs), st(priors), (st(plog),), None, None
    return st(beliefs), st(priors), (st(plog),), st(posts), (st(qlog),)


# ----------------------------------------------------------------------------------------------
# R5/R6: actor, tanh-Normal sample, 100-sample entropy
# ----------------------------------------------------------------------------------------------
RAW_INIT_STD = float(torch.log(torch.exp(torch.tensor(5.0)) - 1))    # src/models.py:503
ACT_MIN_STD = 1e-4                                                    # src/models.py:479
ACT_MEAN_SCALE = 5.0                                                  # src/models.py:481
ATANH_CLAMP = 0.99999997                                              # src/models.py:663


def actor_forward(belief: Tensor, state: Tensor, sd: Dict[str, Tensor]) -> Tuple[Tensor, Tensor]:
    """ActorModel.forward, Gaussian branch (src/models.py:506-517)."""
    out = mlp(torch.cat([belief, state], dim=1), sd)
    m, r = torch.chunk(out, 2, dim=1)
    mean = ACT_MEAN_SCALE * torch.tanh(m / ACT_MEAN_SCALE)
    std = F.softplus(r + RAW_INIT_STD) + ACT_MIN_STD
    return mean, std


def tanh_normal_log_prob(y: Tensor, mean: Tensor, std: Tensor) -> Tensor:
    """Independent(TransformedDistribution(Normal, TanhBijector), 1).log_prob(y)
    (src/models.py:656-673; torch TransformedDistribution.log_prob)."""
    yc = torch.where(torch.abs(y) <= 1.0, torch.clamp(y, -ATANH_CLAMP, ATANH_CLAMP), y)
    x = 0.5 * torch.log((1 + yc) / (1 - yc))                           # atanh, :627
    ladj = 2.0 * (math.log(2) - x - F.softplus(-2.0 * x))              # :673
    base = -((x - mean) ** 2) / (2 * std ** 2) - torch.log(std) - math.log(math.sqrt(2 * math.pi))
    return (base - ladj).sum(-1)


def get_action(belief: Tensor, state: Tensor, sd: Dict[str, Tensor], eps_action: Tensor,
               eps_entropy: Tensor) -> Tuple[Tensor, Tensor]:
    """Dreamer.get_action(deterministic=False) (src/dreamer.py:429-444) + SampleDist.entropy
    (src/models.py:725-733).  eps_action (N,A); eps_entropy (n_samples,N,A)."""
    mean, std = actor_forward(belief, state, sd)
    action = torch.tanh(mean + std * eps_action)
    y = torch.tanh(mean.unsqueeze(0) + std.unsqueeze(0) * eps_entropy)
    logp = tanh_normal_log_prob(y, mean.unsqueeze(0), std.unsqueeze(0))
    return action, -torch.mean(logp, 0)


def get_action_mode(belief: Tensor, state: Tensor, sd: Dict[str, Tensor], eps_mode: Tensor,
                    eps_entropy: Tensor) -> Tuple[Tensor, Tensor]:
    """Dreamer.get_action(deterministic=True) (src/dreamer.py:440-444): SampleDist.mode (src/models.py:709-723) -- of
    n_samples draws the one with the highest log-density per row -- then SampleDist.entropy on fresh draws.
    eps_mode, eps_entropy (n_samples, N, A), consumed in that order."""
    mean, std = actor_forward(belief, state, sd)
    sample = torch.tanh(mean.unsqueeze(0) + std.unsqueeze(0) * eps_mode)                  # :713-714
    logprob = tanh_normal_log_prob(sample, mean.unsqueeze(0), std.unsqueeze(0))           # :715
    idx = torch.argmax(logprob, dim=0).reshape(1, -1, 1).expand(1, sample.size(1), sample.size(2))   # :718-722
    action = torch.gather(sample, 0, idx).squeeze(0)                                      # :723
    y = torch.tanh(mean.unsqueeze(0) + std.unsqueeze(0) * eps_entropy)
    return action, -torch.mean(tanh_normal_log_prob(y, mean.unsqueeze(0), std.unsqueeze(0)), 0)


# ----------------------------------------------------------------------------------------------
# R4: imagine_ahead (src/dreamer.py:179-237)
# ----------------------------------------------------------------------------------------------
def imagine_ahead(P, prev_state: Tensor, prev_belief: Tensor, horizon: int, eps_action: Tensor,
                  eps_entropy: Tensor, eps_prior: Tensor, cat: Optional[Tuple[int, int]] = None):
    """Returns beliefs (H',N,Be), prior_states (H',N,S), (means, stds), action_entropy (H',N).
    ``cat=(D, C)``: Categorical latents (src/dreamer.py:205-206,224-227,234-235 with the tuple repaired as in
    transition_forward_categorical): params = (logits (H',N,D,C),), eps_prior = the sampler's Exp(1) draws (H',N,D*C)."""
    tm = P["transition_model"]
    belief = prev_belief.reshape(-1, prev_belief.size(-1))
    state = prev_state.reshape(-1, prev_state.size(-1))
    bs, ss, ms, sds, ents = [], [], [], [], []
    prior_sd = _sub(tm, "belief_prior") if cat is not None else None
    for t in range(horizon - 1):
        action, ent = get_action(belief.detach(), state.detach(), P["actor"], eps_action[t], eps_entropy[t])
        hidden = embed_state_action(state, action, tm)
        belief = gru_cell(hidden, belief, tm)
        if cat is not None:
            state, (m,) = categorical_belief(belief, prior_sd, eps_prior[t].reshape(-1, cat[0], cat[1]), cat[0], cat[1])
            s = m
        else:
            state, m, s = gaussian_belief(belief, tm, "belief_prior", eps_prior[t])
        bs.append(belief); ss.append(state); ms.append(m); sds.append(s); ents.append(ent)
    st = lambda xs: torch.stack(xs, dim=0)
    if cat is not None:
        return st(bs), st(ss), (st(ms),), st(ents)
    return st(bs), st(ss), (st(ms), st(sds)), st(ents)


# ----------------------------------------------------------------------------------------------
# R8: lambda_return (src/dreamer.py:447-471)
# ----------------------------------------------------------------------------------------------
def lambda_return(imged_reward: Tensor, value_pred: Tensor, bootstrap: Tensor, discount: float = 0.99,
                  lambda_: float = 0.95) -> Tensor:
    next_values = torch.cat([value_pred[1:], bootstrap[None]], 0)
    disc = discount * torch.ones_like(imged_reward)
    inputs = imged_reward + disc * next_values * (1 - lambda_)
    last = bootstrap
    outs = []
    for t in reversed(range(inputs.size(0))):
        last = inputs[t] + disc[t] * lambda_ * last
        outs.append(last)
    return torch.stack(list(reversed(outs)), 0)


# ----------------------------------------------------------------------------------------------
# R3/R9: losses
# ----------------------------------------------------------------------------------------------
def normal_nll_mean(pred: Tensor, target: Tensor, event_dims: int = 1) -> Tensor:
    """-Independent(Normal(pred, 1), k).log_prob(target).mean() (src/planet.py:262-284)."""
    lp = -0.5 * (target - pred) ** 2 - HALF_LOG_2PI
    for _ in range(event_dims):
        lp = lp.sum(-1)
    return -lp.mean()


def kl_normal(qm: Tensor, qs: Tensor, pm: Tensor, ps: Tensor) -> Tensor:
    """torch.distributions.kl._kl_normal_normal(q, p), elementwise."""
    var_ratio = (qs / ps) ** 2
    t1 = ((qm - pm) / ps) ** 2
    return 0.5 * (var_ratio + t1 - 1 - var_ratio.log())


def kl_loss(post: Tuple[Tensor, Tensor], prior: Tuple[Tensor, Tensor], kl_balance: float,
            free_nats: float) -> Tensor:
    """Dreamer._kl_loss (src/dreamer.py:110-146); result has shape (1,) like the reference."""
    qm, qs = post
    pm, ps = prior
    fn = torch.full((1,), free_nats)
    if kl_balance == -1:
        div = kl_normal(qm, qs, pm, ps).sum(dim=2)
        return torch.max(div, fn).mean(dim=(0, 1))
    lhs = kl_normal(qm.detach(), qs.detach(), pm, ps).mean()
    rhs = kl_normal(qm, qs, pm.detach(), ps.detach()).mean()
    return kl_balance * torch.max(lhs, fn) + (1 - kl_balance) * torch.max(rhs, fn)


# ----------------------------------------------------------------------------------------------
# R10: clip_grad_norm_ + Adam (torch.optim.Adam, weight_decay = L2-in-gradient)
# ----------------------------------------------------------------------------------------------
class AdamState:
    def __init__(self, params):
        self.step = 0
        self.m = [torch.zeros_like(p) for p in params]
        self.v = [torch.zeros_like(p) for p in params]


def clip_grad_norm_(grads, max_norm: float) -> Tensor:
    """torch.nn.utils.clip_grad_norm_(norm_type=2): coef = clamp(max_norm/(norm+1e-6), max=1)."""
    total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g) for g in grads]))
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads:
        g.mul_(coef)
    return total


def adam_step(params, grads, st: AdamState, lr: float, eps: float, weight_decay: float,
              beta1: float = 0.9, beta2: float = 0.999) -> None:
    st.step += 1
    bc1 = 1 - beta1 ** st.step
    bc2 = 1 - beta2 ** st.step
    with torch.no_grad():
        for p, g, m, v in zip(params, grads, st.m, st.v):
            g = g.add(p, alpha=weight_decay)
            m.lerp_(g, 1 - beta1)
            v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
            denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
            p.addcdiv_(m, denom, value=-lr / bc1)


# ----------------------------------------------------------------------------------------------
# R2c: Categorical latents -- CategoricalBeliefModel.forward and the Categorical KL branch
# ----------------------------------------------------------------------------------------------
def categorical_belief(inp: Tensor, sd: Dict[str, Tensor], q_noise: Tensor, D: int, C: int):
    """CategoricalBeliefModel.forward (src/models.py:101-117): returns state (rows, D*C), (logits (rows, D, C),).

    ``OneHotCategoricalStraightThrough(logits).rsample()`` = one_hot(sample) + probs - probs.detach() with
    probs = softmax(logits - logsumexp(logits)); the sample is torch.multinomial(probs, 1, True), whose single-draw
    path is ``q = empty_like(probs).exponential_(1); argmax(probs / q)`` -- ``q_noise`` (rows, D, C) are those draws."""
    logits = F.linear(F.elu(F.linear(inp, sd["model.0.weight"], sd["model.0.bias"])), sd["model.2.weight"],
                      sd["model.2.bias"])                                                   # :108
    logits = logits.reshape(*logits.shape[:-1], D, C)                                       # :109-111
    norm = logits - logits.logsumexp(dim=-1, keepdim=True)
    probs = F.softmax(norm, dim=-1)
    idx = (probs / q_noise).argmax(dim=-1)
    sample = F.one_hot(idx, C).to(probs.dtype)
    state = sample + (probs - probs.detach())                                               # :114-115
    return state.reshape(*state.shape[:-2], D * C), (logits,)                               # :116-117


def kl_categorical(q_logits: Tensor, p_logits: Tensor) -> Tensor:
    """kl_divergence(OneHotCategorical(q), OneHotCategorical(p)) per factor (torch.distributions.kl
    ``_kl_categorical_categorical``): sum_c q (log q - log p), with the library's 0 / inf conventions."""
    lq = q_logits - q_logits.logsumexp(dim=-1, keepdim=True)
    lp = p_logits - p_logits.logsumexp(dim=-1, keepdim=True)
    qp, pp = F.softmax(lq, dim=-1), F.softmax(lp, dim=-1)
    t = qp * (lq - lp)
    t = torch.where(pp == 0, torch.full_like(t, float("inf")), t)
    t = torch.where(qp == 0, torch.zeros_like(t), t)
    return t.sum(-1)


def kl_loss_categorical(post_logits: Tensor, prior_logits: Tensor, kl_balance: float, free_nats: float) -> Tensor:
    """Dreamer._kl_loss, Categorical branch (src/dreamer.py:102-106,119-144); logits are (T, B, D, C)."""
    fn = torch.full((1,), free_nats)
    if kl_balance == -1:
        return torch.max(kl_categorical(post_logits, prior_logits).sum(dim=2), fn).mean(dim=(0, 1))   # :122-128
    lhs = kl_categorical(post_logits.detach(), prior_logits).mean()                          # :134
    rhs = kl_categorical(post_logits, prior_logits.detach()).mean()                          # :135
    return kl_balance * torch.max(lhs, fn) + (1 - kl_balance) * torch.max(rhs, fn)           # :140-144


# ----------------------------------------------------------------------------------------------
# MPCPlanner.forward (src/planner.py:28-90): cross-entropy method over prior-only rollouts
# ----------------------------------------------------------------------------------------------
def mpc_planner(P, belief: Tensor, state: Tensor, action_size: int, planning_horizon: int, optimisation_iters: int,
                candidates: int, top_candidates: int, eps_action: Tensor, eps_state: Tensor, trace: Optional[list] = None):
    """belief (B,Be), state (B,S) -> first action mean (B,A).

    eps_action (iters, H, B, candidates, A): the ``torch.randn`` draws of src/planner.py:53-59;
    eps_state (iters, H, B*candidates, S): the prior-state draws of the rollout (src/models.py:256 -> :72).
    ``trace`` (optional list) receives per iteration (returns (B*candidates,), action_mean, action_std)."""
    B, Hb, Z = belief.size(0), belief.size(1), state.size(1)
    belief = belief.unsqueeze(1).expand(B, candidates, Hb).reshape(-1, Hb)                        # :37
    state = state.unsqueeze(1).expand(B, candidates, Z).reshape(-1, Z)                            # :38
    mean = torch.zeros(planning_horizon, B, 1, action_size)                                       # :41-43
    std = torch.ones(planning_horizon, B, 1, action_size)                                         # :44-46
    for it in range(optimisation_iters):
        actions = (mean + std * eps_action[it]).view(planning_horizon, B * candidates, action_size)   # :60-62
        beliefs, states, _, _, _ = transition_forward(P["transition_model"], state, actions, belief, None, None,
                                                      eps_state[it], None)                        # :65
        returns = dense_on_features(beliefs.view(-1, Hb), states.view(-1, Z), P["reward_model"]) \
            .view(planning_horizon, -1).sum(dim=0)                                                # :68-72
        _, topk = returns.reshape(B, candidates).topk(top_candidates, dim=1, largest=True, sorted=False)   # :74-76
        topk = topk + candidates * torch.arange(0, B, dtype=torch.int64).unsqueeze(1)             # :78-80
        best = actions[:, topk.view(-1)].reshape(planning_horizon, B, top_candidates, action_size)   # :81-83
        mean = best.mean(dim=2, keepdim=True)                                                     # :86
        std = best.std(dim=2, unbiased=False, keepdim=True)                                       # :87
        if trace is not None:
            trace.append((returns.detach().clone(), mean.detach().clone(), std.detach().clone()))
    return mean[0].squeeze(dim=1)                                                                 # :90


# ----------------------------------------------------------------------------------------------
# the whole step: Dreamer.train_step (src/dreamer.py:253-393)
# ----------------------------------------------------------------------------------------------
DEFAULT_HP = dict(
    kl_balance=0.8, kl_loss_weight=0.1, free_nats=3.0, grad_clip_norm=100.0, discount=0.995, disclam=0.95,
    model_learning_rate=2e-4, actor_learning_rate=4e-5, value_learning_rate=1e-4, adam_epsilon=1e-5,
    weight_decay=1e-6, entropy_weight=1e-5, planning_horizon=15, discount_weight=5.0,
)

MODEL_MODULES = ("transition_model", "observation_model", "reward_model", "encoder")   # dreamer.py:160-165


class OracleDreamer:
    """Holds parameters + Adam state; ``train_step`` mirrors src/dreamer.py:253-393 line by line."""

    def __init__(self, P_numpy, hp=None):
        self.hp = dict(DEFAULT_HP)
        if hp:
            self.hp.update(hp)
        # latent_distribution="Categorical": hp["categorical"] = (discrete_latent_dimensions, discrete_latent_classes)
        self.cat = tuple(self.hp["categorical"]) if self.hp.get("categorical") else None
        self.P = {mod: {k: torch.tensor(v, dtype=torch.float32, requires_grad=(mod != "critic_target"))
                        for k, v in sd.items()} for mod, sd in P_numpy.items()}
        # use_discount=True: the discount head joins the model optimiser last (src/dreamer.py:167-169)
        self.use_discount = "discount_model" in self.P
        self.model_modules = MODEL_MODULES + (("discount_model",) if self.use_discount else ())
        self.model_params = [p for mod in self.model_modules for p in self.P[mod].values()]
        self.actor_params = list(self.P["actor"].values())
        self.critic_params = list(self.P["critic"].values())
        self.opt = {"model": AdamState(self.model_params), "actor": AdamState(self.actor_params),
                    "critic": AdamState(self.critic_params)}
        self.last = {}

    def update_critic(self, weight: float = 1.0):
        """polyak_update (src/utils.py:56-78)."""
        with torch.no_grad():
            for k, p in self.P["critic"].items():
                t = self.P["critic_target"][k]
                t.copy_(p * weight + t * (1.0 - weight))

    def world_model_forward(self, batch, noise):
        hp, P = self.hp, self.P
        obs, actions, rewards, nonterm = (batch[k] for k in ("observations", "actions", "rewards", "nonterminals"))
        B = obs.size(1)
        Be = P["transition_model"]["rnn.weight_hh"].size(1)
        S = P["transition_model"]["belief_prior.model.2.weight"].size(0) // (1 if self.cat else 2)
        init_belief = torch.zeros(B, Be)
        init_state = torch.zeros(B, S)
        pixel = obs.dim() == 5
        emb = cnn_encoder(obs[1:], P["encoder"]) if pixel else mlp(obs[1:], P["encoder"])      # :270
        beliefs, prior_states, prior_params, post_states, post_params = transition_forward(
            P["transition_model"], init_state, actions[:-1], init_belief, emb, nonterm[:-1],
            noise["obs_prior"], noise["obs_post"], self.cat)                                   # :272-278
        if pixel:   # Independent(Normal(means, 1), 3) (src/planet.py:264-265)
            obs_loss = normal_nll_mean(cnn_decoder(beliefs, post_states, P["observation_model"]), obs[1:], event_dims=3)
        else:
            obs_loss = normal_nll_mean(dense_on_features(beliefs, post_states, P["observation_model"]), obs[1:])
        rew_pred = dense_on_features(beliefs, post_states, P["reward_model"])
        rew_loss = normal_nll_mean(rew_pred, rewards[:-1].unsqueeze(-1))
        self._discount_loss = None
        if self.use_discount:   # _discount_loss (src/dreamer.py:239-251): -mean log Bernoulli(logits).prob(nonterminal)
            logits = dense_on_features(beliefs, post_states, P["discount_model"])
            self._discount_loss = F.binary_cross_entropy_with_logits(logits, nonterm[:-1].float(), reduction="none").sum(-1).mean()
        if self.cat:
            kl = kl_loss_categorical(post_params[0], prior_params[0], hp["kl_balance"], hp["free_nats"])
            model_loss = obs_loss + rew_loss + kl * hp["kl_loss_weight"]                       # :285
            if self._discount_loss is not None:
                model_loss = model_loss + self._discount_loss * hp["discount_weight"]          # :287-289
            inter = dict(embeddings=emb, beliefs=beliefs, prior_states=prior_states, prior_logits=prior_params[0],
                         posterior_states=post_states, posterior_logits=post_params[0], reward_pred=rew_pred)
            return model_loss, obs_loss, rew_loss, kl, inter
        kl = kl_loss(post_params, prior_params, hp["kl_balance"], hp["free_nats"])
        model_loss = obs_loss + rew_loss + kl * hp["kl_loss_weight"]                           # :285
        if self._discount_loss is not None:
            model_loss = model_loss + self._discount_loss * hp["discount_weight"]              # :287-289
        inter = dict(embeddings=emb, beliefs=beliefs, prior_states=prior_states, prior_means=prior_params[0],
                     prior_stds=prior_params[1], posterior_states=post_states, posterior_means=post_params[0],
                     posterior_stds=post_params[1], reward_pred=rew_pred)
        return model_loss, obs_loss, rew_loss, kl, inter

    def planet_train_step(self, batch_np, noise_np, keep: bool = True):
        """Planet.train_step (src/planet.py:310-368): dynamics learning only; its ``_kl_loss`` (src/planet.py:286-308)
        is always the summed form ``max(KL.sum(2), free_nats).mean()`` whatever ``kl_balance`` says."""
        hp = self.hp
        assert hp["kl_balance"] == -1, "Planet._kl_loss is the kl_balance == -1 form"
        batch = {k: torch.as_tensor(v) for k, v in batch_np.items()}
        noise = {k: torch.as_tensor(v) for k, v in noise_np.items()}
        model_loss, obs_loss, rew_loss, kl, inter = self.world_model_forward(batch, noise)
        logs = dict(observation_loss=obs_loss.item(), reward_loss=rew_loss.item(), kl_loss=kl.item(),
                    model_loss=model_loss.item())
        grads = torch.autograd.grad(model_loss, self.model_params, allow_unused=True)
        grads = [torch.zeros_like(p) if g is None else g.clone() for g, p in zip(grads, self.model_params)]
        model_grads = [g.clone() for g in grads] if keep else None
        gn_model = clip_grad_norm_(grads, hp["grad_clip_norm"])
        adam_step(self.model_params, grads, self.opt["model"], hp["model_learning_rate"], hp["adam_epsilon"],
                  hp["weight_decay"])
        if keep:
            self.last = dict(inter={k: t.detach() for k, t in inter.items()}, model_grads=model_grads,
                             grad_norms=dict(model=gn_model.item()))
        return logs

    def train_step(self, batch_np, noise_np, keep: bool = True):
        hp, P = self.hp, self.P
        batch = {k: torch.as_tensor(v) for k, v in batch_np.items()}
        noise = {k: torch.as_tensor(v) for k, v in noise_np.items()}
        logs = {}
        # ---------------- dynamics learning ----------------
        model_loss, obs_loss, rew_loss, kl, inter = self.world_model_forward(batch, noise)
        logs.update(observation_loss=obs_loss.item(), reward_loss=rew_loss.item(), kl_loss=kl.item(),
                    model_loss=model_loss.item())
        if self._discount_loss is not None:
            logs["discount_loss"] = self._discount_loss.item()                                 # :290
        grads = torch.autograd.grad(model_loss, self.model_params, allow_unused=True)
        grads = [torch.zeros_like(p) if g is None else g.clone() for g, p in zip(grads, self.model_params)]
        model_grads = [g.clone() for g in grads] if keep else None
        gn_model = clip_grad_norm_(grads, hp["grad_clip_norm"])
        adam_step(self.model_params, grads, self.opt["model"], hp["model_learning_rate"], hp["adam_epsilon"],
                  hp["weight_decay"])
        # ---------------- behaviour learning ----------------
        beliefs = inter["beliefs"].detach()
        post_states = inter["posterior_states"].detach()
        # FreezeParameters(model_modules): world-model weights are constants here (dreamer.py:313);
        # they are the *post-update* weights.
        Pf = dict(P)
        for mod in self.model_modules + ("critic_target",):
            Pf[mod] = {k: v.detach() for k, v in P[mod].items()}
        img_b, img_s, _, ent = imagine_ahead(Pf, post_states, beliefs, hp["planning_horizon"], noise["action"],
                                             noise["entropy"], noise["img_prior"], self.cat)
        img_reward = dense_on_features(img_b, img_s, Pf["reward_model"])                      # :321
        value_pred = dense_on_features(img_b, img_s, Pf["critic_target"])                     # :322
        returns = lambda_return(img_reward, value_pred, value_pred[-1], hp["discount"], hp["disclam"])
        objective = returns + hp["entropy_weight"] * ent.unsqueeze(-1)                         # :346
        wts = None
        if self.use_discount:
            # discount_arr = discount * round(Bernoulli(logits).probs) (:323-326; no gradient: round, frozen weights);
            # `discount_arr[:, 0, 0] = 1.0` sets trajectory 0 at EVERY step (the comment says "the first one of each
            # trajectory"; the indexing says otherwise -- reproduced as written); weights = cumprod over time (:349-351)
            with torch.no_grad():
                dl = dense_on_features(img_b, img_s, Pf["discount_model"])
                arr = hp["discount"] * torch.round(torch.sigmoid(dl))
                arr[:, 0, 0] = 1.0
                wts = torch.cumprod(arr, 0)
            objective = wts * objective
        actor_loss = -objective.mean()
        logs.update(actor_loss=actor_loss.item(), policy_entropy=ent.mean().item())
        agrads = [g.clone() for g in torch.autograd.grad(actor_loss, self.actor_params)]
        actor_grads = [g.clone() for g in agrads] if keep else None
        gn_actor = clip_grad_norm_(agrads, hp["grad_clip_norm"])
        adam_step(self.actor_params, agrads, self.opt["actor"], hp["actor_learning_rate"], hp["adam_epsilon"],
                  hp["weight_decay"])
        # critic (dreamer.py:370-391)
        v = dense_on_features(img_b.detach(), img_s.detach(), P["critic"])
        target = returns.detach()
        nll = 0.5 * (target - v) ** 2 + HALF_LOG_2PI
        value_loss = (wts * nll).mean() if wts is not None else nll.mean()                      # :378-381
        logs.update(value_loss=value_loss.item())
        cgrads = [g.clone() for g in torch.autograd.grad(value_loss, self.critic_params)]
        critic_grads = [g.clone() for g in cgrads] if keep else None
        gn_critic = clip_grad_norm_(cgrads, hp["grad_clip_norm"])
        adam_step(self.critic_params, cgrads, self.opt["critic"], hp["value_learning_rate"], hp["adam_epsilon"],
                  hp["weight_decay"])
        if keep:
            self.last = dict(inter={k: t.detach() for k, t in inter.items()}, imged_beliefs=img_b.detach(),
                             imged_states=img_s.detach(), action_entropy=ent.detach(),
                             imged_reward=img_reward.detach(), value_pred=value_pred.detach(),
                             returns=returns.detach(), critic_value=v.detach(),
                             model_grads=model_grads, actor_grads=actor_grads, critic_grads=critic_grads,
                             grad_norms=dict(model=gn_model.item(), actor=gn_actor.item(), critic=gn_critic.item()))
        return logs
