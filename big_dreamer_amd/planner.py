"""CEM planner with the reference's call surface (src/planner.py) on the HIP engine.

``MPCPlanner(action_size, planning_horizon, optimisation_iters, candidates, top_candidates, transition_model,
reward_model)`` and ``forward(belief, state) -> (B, A)`` are the reference's (src/planner.py:10-35).  Each CEM
iteration is two launches: ``bd_plan_rollout`` (candidate actions, prior-only RSSM rollout and reward model fused
per planning step, returns summed in LDS) and ``bd_cem_refit`` (top-k selection + mean / std refit).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
from torch import Tensor, nn


class MPCPlanner(nn.Module):
    """Model-predictive control planner: cross-entropy method over the learned transition model."""

    def __init__(self, action_size: int, planning_horizon: int, optimisation_iters: int, candidates: int,
                 top_candidates: int, transition_model, reward_model):
        super().__init__()
        eng = getattr(transition_model, "_eng", None)
        if eng is None or getattr(reward_model, "_eng", None) is not eng:
            raise TypeError("MPCPlanner needs the engine-backed TransitionModel and reward DenseModel of one agent")
        if action_size != eng.d.A:
            raise ValueError(f"action_size {action_size} differs from the transition model's {eng.d.A}")
        if not 0 < top_candidates <= candidates:
            raise ValueError("top_candidates must be in 1..candidates")
        # plain attributes (not sub-modules): the planner owns no parameters of its own
        object.__setattr__(self, "transition_model", transition_model)
        object.__setattr__(self, "reward_model", reward_model)
        object.__setattr__(self, "_eng", eng)
        self.action_size = action_size
        self.planning_horizon = planning_horizon
        self.optimisation_iters = optimisation_iters
        self.candidates, self.top_candidates = candidates, top_candidates

    @torch.no_grad()
    def forward(self, belief: Tensor, state: Tensor, _noise: Optional[Dict[str, Tensor]] = None,
                _trace: Optional[list] = None) -> Tensor:
        """belief (B, Be), state (B, S) -> first action mean (B, A)  (src/planner.py:28-90).

        ``_noise``: {"action": (iters, H, B, candidates, A), "state": (iters, H, B*candidates, S)} standard-normal
        draws in the reference's order; drawn on the device when absent."""
        eng, d = self._eng, self._eng.d
        eng.join()                                   # order after queued pipeline work (engine.train_step)
        B = belief.shape[0]
        H, I, J = self.planning_horizon, self.optimisation_iters, self.candidates
        f = lambda t: t.to(eng.dev).contiguous().float()
        if _noise is None:
            eps_a = torch.randn(I, H, B, J, d.A, device=eng.dev)
            eps_s = torch.randn(I, H, B * J, d.S, device=eng.dev)
        else:
            eps_a, eps_s = f(_noise["action"]), f(_noise["state"])
            assert tuple(eps_a.shape) == (I, H, B, J, d.A) and tuple(eps_s.shape) == (I, H, B * J, d.S)
        mean = eng.plan(f(belief), f(state), H, I, J, self.top_candidates, eps_a, eps_s, _trace)
        return mean[0].clone()                       # first action mean (src/planner.py:90)
