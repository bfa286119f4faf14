#!/usr/bin/env python3
"""CLI with the reference's surface: ``python src/main.py key=value ...`` (reference src/main.py:20-285).

Collect-update loop: every ``environment_steps_per_update`` env steps run ``collect_interval`` train steps
(src/main.py:103-108), critic-target update cadence as at src/main.py:110-112, one env step with exploration
noise (src/main.py:129-143), append to the replay buffer (src/main.py:146), log every ``log_freq``.
Multi-GPU: launch with ``python -m torch.distributed.run --nproc-per-node N src/main.py ...``; each rank collects
its own experience and the gradients are all-reduced over RCCL (big_dreamer_amd/engine.py).
"""
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import big_dreamer_amd  # noqa: E402,F401  (first: sets the HIP runtime's queue count before HIP initialises)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from big_dreamer_amd.config import load_config  # noqa: E402


def my_app(argv):
    params = load_config(argv)
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    np.random.seed(params["seed"] + rank)
    torch.manual_seed(params["seed"])                 # identical initial weights on every rank
    random.seed(params["seed"] + rank)
    if params["algorithm"] not in ("planet", "dreamer", "dreamerV2"):     # as src/main.py:73-81
        raise NotImplementedError(f'algorithm {params["algorithm"]} is not yet implemented.')
    local = int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    if world > 1:      # (device_id: the communicator is built now, before the engine's streams -- DESIGN.md section 6)
        torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local))
    torch.cuda.set_stream(torch.cuda.Stream())        # stay off the legacy null stream
    from big_dreamer_amd.dreamer import Dreamer, DreamerV2
    from big_dreamer_amd.planet import Planet
    from big_dreamer_amd.env import Env
    env = Env(params)
    agent_cls = {"planet": Planet, "dreamer": Dreamer, "dreamerV2": DreamerV2}[params["algorithm"]]
    model = agent_cls(params, env, world_size=world)
    torch.manual_seed(params["seed"] + rank)
    env_steps, num_episodes = model.randomly_initialize_replay_buffer()
    if rank == 0:
        print(f"Initialized with {num_episodes} episodes and {env_steps} steps")
    dev = model.device
    observation = env.reset()
    belief = torch.zeros(1, params["belief_size"], device=dev)
    # (the reference sizes this with params["state_size"], src/main.py:94, which is wrong for Categorical latents, where
    # the agent's state_size is dimensions * classes, src/planet.py:56-57)
    posterior_state = torch.zeros(1, model.state_size, device=dev)
    action = torch.zeros(1, env.action_size, device=dev)
    logs, episode_reward, past = {}, 0.0, time.time()
    for step in range(env_steps, params["train_steps"]):
        if step % params["environment_steps_per_update"] == 0:
            t0 = time.time()
            for _ in range(params["collect_interval"]):
                logs = model.train_step()
            logs["weight_update_per_sec"] = params["collect_interval"] / (time.time() - t0)
        if params["algorithm"] != "planet" and step % params["ActorCritic"]["slow_critic_update_interval"]:
            model.update_critic()                                              # cadence as in the reference (:110-112)
        belief, posterior_state, action, next_observation, reward, done = model.update_belief_and_act(
            env, belief, posterior_state, action, observation, explore=True)
        model.buffer.append(observation, action.cpu()[0], reward, done)
        episode_reward += reward
        observation = next_observation
        if done:
            logs["episode_total_reward"] = episode_reward
            observation, episode_reward = env.reset(), 0.0
            belief.zero_(); posterior_state.zero_(); action.zero_()
        if step % params["log_freq"] == 0 and rank == 0:
            logs["env_update_per_sec"] = params["log_freq"] / max(time.time() - past, 1e-9)
            past = time.time()
            print(step, {k: (round(v, 5) if isinstance(v, float) else v) for k, v in logs.items()}, flush=True)
    env.close()


if __name__ == "__main__":
    my_app(sys.argv[1:])
