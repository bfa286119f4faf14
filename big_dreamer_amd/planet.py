"""PlaNet agent with the reference's call surface (src/planet.py) on the HIP engine.

``Planet.train_step`` is dynamics learning alone (src/planet.py:310-368) with the summed free-nats KL of
``Planet._kl_loss`` (src/planet.py:286-308); acting plans with the CEM planner (src/planet.py:405-410).

The reference's ``update_belief_and_act`` unpacks ``self.get_action(...)`` into ``action, _`` (src/planet.py:386)
although ``Planet.get_action`` returns the planner's bare tensor -- at HEAD PlaNet cannot act.  Here ``get_action``
returns the planned action and ``update_belief_and_act`` uses it directly (the evident intent).
"""
from __future__ import annotations

from typing import Any, Dict, Optional

import torch
from torch import Tensor

from .dreamer import Dreamer
from .planner import MPCPlanner


class Planet(Dreamer):
    def __init__(self, params: Dict[str, Any], env, **kw):
        p = dict(params)
        p["kl_balance"] = -1           # Planet._kl_loss ignores kl_balance: max(KL.sum(2), free_nats).mean()
        super().__init__(p, env, **kw)
        mpc = params["MPC"]
        self.optimisation_iters, self.candidates, self.top_candidates = (mpc["optimisation_iters"], mpc["candidates"],
                                                                         mpc["top_candidates"])
        self.planner = MPCPlanner(self.action_size, self.planning_horizon, self.optimisation_iters, self.candidates,
                                  self.top_candidates, self.transition_model, self.reward_model)

    def train_step(self) -> Dict[str, float]:
        """src/planet.py:310-368.  Returns the reference's log dict (four losses)."""
        obs, actions, rewards, nonterminals = self.buffer.sample(self.batch_size, self.seq_len)
        logs = self.engine.world_model_step({"observations": obs, "actions": actions, "rewards": rewards,
                                             "nonterminals": nonterminals})
        return {k: v for k, v in logs.items() if not k.startswith("grad_norm")}

    def update_critic(self) -> None:
        raise NotImplementedError("PlaNet has no critic (src/main.py:110 updates it for dreamer / dreamerV2 only)")

    @torch.no_grad()
    def get_action(self, belief: Tensor, state: Tensor, deterministic: bool = False,
                   _noise: Optional[Dict[str, Tensor]] = None) -> Tensor:
        """src/planet.py:405-410: the planner's first action mean, (B, A)."""
        return self.planner(belief, state, _noise=_noise)

    @torch.no_grad()
    def update_belief_and_act(self, env, belief, posterior_state, action, observation, explore=False):
        """src/planet.py:370-403 (see the module docstring for the one deviation)."""
        embedding = self.encoder(observation.to(self.device)).unsqueeze(dim=0)
        belief, _, _, posterior_state, _ = self.transition_model(posterior_state, action.unsqueeze(dim=0), belief,
                                                                 embedding)
        belief, posterior_state = belief.squeeze(dim=0), posterior_state.squeeze(dim=0)
        action = self.get_action(belief, posterior_state)
        if explore:
            action = torch.clamp(action + self.action_noise * torch.randn_like(action), -1, 1)
        batched = hasattr(env, "n") and hasattr(env, "envs")          # EnvBatcher (src/env.py:343)
        next_observation, reward, done = env.step(action.cpu() if batched else action[0].cpu())
        return belief, posterior_state, action, next_observation, reward, done
