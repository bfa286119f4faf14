"""Latency of one collect-loop decision on the GPU box: Dreamer.update_belief_and_act (reference src/planet.py:370-403:
encoder -> one RSSM cell step -> actor sample + 100-sample entropy -> exploration noise -> action.cpu()) at B=1
(collection) and B=10 (evaluation), config-2 model size.  Writes profiles/<tag>_act_latency.json."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from big_dreamer_amd import synth  # noqa: E402
from big_dreamer_amd.config import load_config  # noqa: E402
from big_dreamer_amd.dreamer import Dreamer  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
d = synth.CONFIG2
out = {"what": "Dreamer.update_belief_and_act, host wall time per call incl. the action's D2H copy (the env.step input)",
       "model": "belief=200 state=30 hidden=200 embedding=1024 action=1 obs=3"}
for B in (1, 10):
    class Env:
        action_size, observation_size = d.A, d.O

        def __init__(self):
            if B > 1:
                self.n, self.envs = B, [None] * B

        def step(self, a):
            return torch.zeros(B, d.O), 0.0, False

    torch.manual_seed(0)
    agent = Dreamer(load_config(["experience_size=100"]), Env())
    env = Env()
    belief, state = torch.zeros(B, d.Be).cuda(), torch.zeros(B, d.S).cuda()
    action, obs = torch.zeros(B, d.A).cuda(), torch.zeros(B, d.O)
    for _ in range(20):
        belief, state, action, obs, _, _ = agent.update_belief_and_act(env, belief, state, action, obs, explore=True)
    torch.cuda.synchronize()
    n = 300
    t0 = time.perf_counter()
    for _ in range(n):
        belief, state, action, obs, _, _ = agent.update_belief_and_act(env, belief, state, action, obs, explore=True)
    torch.cuda.synchronize()
    out[f"B={B}"] = {"us_per_act": (time.perf_counter() - t0) / n * 1e6, "calls": n}
    del agent
print(json.dumps(out))
os.makedirs(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", f"{tag}_act_latency.json"), "w"),
          indent=1)
