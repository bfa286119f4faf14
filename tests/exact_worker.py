"""Worker of test_exact_math_build: runs in a process whose BD_LIB points at the -DBD_EXACT_MATH build of the library
(libm expm1f / log1pf / tanhf in the activation epilogues instead of v_exp_f32 / v_log_f32 / v_rcp_f32) -- two train
steps of the `small` golden case against the oracle, printing the worst errors."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from big_dreamer_amd import _cabi, synth  # noqa: E402
from big_dreamer_amd.engine import DreamerEngine  # noqa: E402
from oracle import dreamer_oracle as O  # noqa: E402

assert "exact" in _cabi.LIB_PATH, _cabi.LIB_PATH
d, seed = synth.SMALL, 1
P, batch = synth.make_params(d, seed), synth.make_batch(d, seed)
eng = DreamerEngine(d, None, "cuda", params=P)
od = O.OracleDreamer(P, dict(planning_horizon=d.H))
dev = lambda dct: {k: torch.as_tensor(v).cuda().contiguous() for k, v in dct.items()}
worst = {"log": 0.0, "weight": 0.0, "belief": 0.0}
for step in range(2):
    nz = synth.make_noise(d, seed + step)
    ologs = od.train_step(batch, nz)
    logs = eng.train_step(dev(batch), dev(nz))
    torch.cuda.synchronize()
    for k, v in ologs.items():
        if k not in ("policy_entropy", "actor_loss"):
            worst["log"] = max(worst["log"], abs(logs[k] - v) / (1.0 + abs(v)))
    if step == 0:
        f = eng._buf["p0_feat"].cpu().numpy().reshape(d.T, d.B, -1)[..., :d.Be]
        worst["belief"] = float(np.abs(f - od.last["inter"]["beliefs"].numpy()).max())
    for mod in list(O.MODEL_MODULES) + ["actor", "critic"]:
        for k, p in od.P[mod].items():
            worst["weight"] = max(worst["weight"], float(np.abs(eng.W(mod, k).cpu().numpy() - p.detach().numpy()).max()))
print("EXACT_RESULT " + json.dumps(worst))
