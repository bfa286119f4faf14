#!/usr/bin/env python3
"""Head chain (230 -> 200 x4 -> 1 over 34 300 rows) through bd_mlp_forward / bd_mlp_backward alone on the GPU: the 16-row
per-tile form against the tall form (48-row workgroups, balanced pairs, in-place image); parity of every output."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from big_dreamer_amd import _cabi as cabi  # noqa: E402
from big_dreamer_amd import synth  # noqa: E402
from big_dreamer_amd.engine import DreamerEngine  # noqa: E402

lib = cabi.lib


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def main():
    torch.manual_seed(0)
    d = synth.CONFIG2
    eng = DreamerEngine(d, None, "cuda", params=synth.make_params(d, 0))
    Mi, F = d.Hm * d.N, d.Be + d.S
    ifeat = torch.randn(Mi, F, device="cuda")
    head_fl = 2.0 * Mi * (F * 200 + 3 * 200 * 200 + 200)
    d_r = torch.randn(Mi, device="cuda")
    keep = {}
    for mode, name in ((0, "per-tile (16 rows)"), (1, "tall (48 rows, balanced pairs)")):
        lib.bd_mlp_set_tall(mode)
        print(f"--- {name} ---")
        t = timed(lambda: eng.dense_forward("reward_model", "rew", "ir", ifeat, F, Mi, 1))
        print(f"forward  (saves): {t:7.1f} us = {head_fl / t / 1e6:6.1f} TFLOP/s = {head_fl / t / 1e6 / 157.3:.3f} of peak")
        r_out, r_acts, r_layers = eng.dense_forward("reward_model", "rew", "ir", ifeat, F, Mi, 1)
        difeat = torch.zeros(Mi, F, device="cuda")
        t = timed(lambda: eng.mlp_backward(Mi, d_r, 1, r_layers, r_acts + [None], None, din0=difeat, ld0=F, w0=F))
        print(f"backward (to features, no dpre): {t:7.1f} us = {head_fl / t / 1e6:6.1f} TFLOP/s")
        dpre = [torch.zeros(Mi, 200, device="cuda") for _ in range(4)]
        t = timed(lambda: eng.mlp_backward(Mi, d_r, 1, r_layers, r_acts + [None], dpre + [None], din0=difeat, ld0=F, w0=F))
        print(f"backward (features + dpre out):  {t:7.1f} us = {head_fl / t / 1e6:6.1f} TFLOP/s")
        torch.cuda.synchronize()
        keep[mode] = [r_out.clone()] + [x.clone() for x in r_acts] + [difeat.clone()] + [x.clone() for x in dpre]
    lib.bd_mlp_set_tall(-1)
    worst = 0.0
    for x, y in zip(keep[0], keep[1]):
        worst = max(worst, float((x - y).abs().max()) / max(1.0, float(x.abs().max())))
    print(f"max scaled difference between the two forms over out / saves / d features / dpre: {worst:.2e}")


if __name__ == "__main__":
    main()
