#!/usr/bin/env python3
"""Diagnostic: imagination forward kernel with and without the saved-activation stores (are the streaming stores /
their address translation on the phases' critical path?)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from big_dreamer_amd import synth  # noqa: E402
from big_dreamer_amd.engine import DreamerEngine  # noqa: E402

d = synth.CONFIG2
eng = DreamerEngine(d, None, "cuda", params=synth.make_params(d, 0))
eng.pipeline = False
N, Hm = d.T * d.B, d.Hm
feat = torch.randn(N, d.Be + d.S, device="cuda") * 0.1
noise = eng.make_noise(d.B)
for save in (True, False, True, False):
    for _ in range(3):
        eng.imagine(feat, N, Hm, noise, save=save)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        eng.imagine(feat, N, Hm, noise, save=save)
    e1.record()
    torch.cuda.synchronize()
    print(f"imagine_fwd save={save}: {e0.elapsed_time(e1) / 20:.4f} ms")
