"""Environment surface the agent needs (src/planet.py:36,91,147-158): ``action_size``, ``observation_size``,
``reset/step/sample_random_action/close``.  gym / mujoco are not installed in the build image, so the CLI
falls back to a small synthetic control task with the same interface; any object with this interface works."""
from __future__ import annotations

import numpy as np
import torch


class SyntheticEnv:
    """Damped linear system x' = A x + B u + noise, reward = -|x|^2 - 0.1|u|^2, fixed episode length.
    Observations are the state (state-observation mode), actions live in [-1, 1]."""

    def __init__(self, observation_size: int = 3, action_size: int = 1, max_episode_length: int = 1000,
                 action_repeat: int = 2, seed: int = 0):
        self.observation_size, self.action_size = observation_size, action_size
        self._nx = observation_size          # state dimension (the pixel subclass changes observation_size)
        self.max_episode_length, self.action_repeat = max_episode_length, action_repeat
        rng = np.random.default_rng(seed)
        q, _ = np.linalg.qr(rng.standard_normal((observation_size, observation_size)))
        self.A = (0.97 * q).astype(np.float32)
        self.B = (0.3 * rng.standard_normal((observation_size, action_size))).astype(np.float32)
        self.rng = rng
        self.t = 0
        self.x = np.zeros(observation_size, np.float32)

    def reset(self) -> torch.Tensor:
        self.t = 0
        self.x = self.rng.standard_normal(self._nx).astype(np.float32)
        return torch.from_numpy(self.x.copy()).unsqueeze(0)

    def step(self, action):
        u = np.asarray(action.detach().cpu().numpy() if isinstance(action, torch.Tensor) else action,
                       dtype=np.float32).reshape(-1)[: self.action_size]
        reward = 0.0
        for _ in range(self.action_repeat):
            self.x = self.A @ self.x + self.B @ u + 0.01 * self.rng.standard_normal(self._nx).astype(np.float32)
            reward += float(-(self.x ** 2).sum() - 0.1 * (u ** 2).sum())
            self.t += 1
            if self.t >= self.max_episode_length:
                break
        done = self.t >= self.max_episode_length
        return torch.from_numpy(self.x.copy()).unsqueeze(0), reward, done

    def sample_random_action(self) -> torch.Tensor:
        return torch.from_numpy(self.rng.uniform(-1, 1, self.action_size).astype(np.float32))

    def close(self) -> None:
        pass


class SyntheticPixelEnv(SyntheticEnv):
    """Same dynamics, observed as a 3x64x64 image in [-0.5, 0.5]: a fixed random linear rendering of the state
    squashed with tanh (stands in for the reference's rendered gym frames, src/env.py:235-317)."""

    def __init__(self, state_size: int = 3, action_size: int = 1, max_episode_length: int = 1000, action_repeat: int = 2,
                 seed: int = 0):
        super().__init__(state_size, action_size, max_episode_length, action_repeat, seed)
        self._render = (self.rng.standard_normal((3 * 64 * 64, state_size)) / np.sqrt(state_size)).astype(np.float32)
        self.state_size = state_size
        self.observation_size = (3, 64, 64)

    def _img(self):
        return torch.from_numpy((0.5 * np.tanh(self._render @ self.x)).astype(np.float32).reshape(1, 3, 64, 64))

    def reset(self):
        self.t = 0
        self.x = self.rng.standard_normal(self.state_size).astype(np.float32)
        return self._img()

    def step(self, action):
        _, reward, done = super().step(action)
        return self._img(), reward, done


def Env(params):
    """Factory with the reference's name (src/env.py:320-340)."""
    if params.get("pixel_observation", False):
        return SyntheticPixelEnv(int(params.get("synthetic_env_observation_size", 3)),
                                 int(params.get("synthetic_env_action_size", 1)), int(params["max_episode_length"]),
                                 int(params["action_repeat"]), int(params["seed"]))
    return SyntheticEnv(int(params.get("synthetic_env_observation_size", 3)),
                        int(params.get("synthetic_env_action_size", 1)), int(params["max_episode_length"]),
                        int(params["action_repeat"]), int(params["seed"]))
