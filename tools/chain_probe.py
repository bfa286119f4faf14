#!/usr/bin/env python3
"""Dense-chain micro-benchmark at the imagination's 34 300 rows, alone on the GPU: the library GEMM (torch.mm -> rocBLAS /
hipBLASLt) as the incumbent for one 200x200 layer, the chain kernel (bd_mlp_forward / bd_mlp_backward) for 1 / 3 layers and
for the real head shape (230 -> 200 x4 -> 1, saved activations), and the weight-gradient GEMM."""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from big_dreamer_amd import _cabi as cabi  # noqa: E402
from big_dreamer_amd import synth  # noqa: E402
from big_dreamer_amd.engine import DreamerEngine  # noqa: E402

lib = cabi.lib


def timed(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def main():
    torch.manual_seed(0)
    M = 34300
    x = torch.randn(M, 200, device="cuda")
    w = torch.randn(200, 200, device="cuda") / 14
    out = torch.empty(M, 200, device="cuda")
    fl = 2.0 * M * 200 * 200
    t = timed(lambda: torch.mm(x, w.t(), out=out))
    print(f"torch.mm 34300x200x200 (library): {t:.1f} us = {fl / t / 1e6:.1f} TFLOP/s")
    xt = x.t().contiguous()
    t = timed(lambda: torch.mm(w, xt))
    print(f"torch.mm 200x200 @ 200x34300 (library, NN): {t:.1f} us = {fl / t / 1e6:.1f} TFLOP/s")
    d = synth.CONFIG2
    eng = DreamerEngine(d, None, "cuda", params=synth.make_params(d, 0))
    Mi, F = d.Hm * d.N, d.Be + d.S
    ifeat = torch.randn(Mi, F, device="cuda")
    head_fl = 2.0 * Mi * (F * 200 + 3 * 200 * 200 + 200)
    d_r = torch.randn(Mi, device="cuda")
    keep = {}
    for mode, name in ((0, "per-tile chain (16 rows)"), (1, "tall chain (48 rows)")):
        lib.bd_mlp_set_tall(mode)
        print(f"--- {name} ---")
        t = timed(lambda: eng.dense_forward("reward_model", "rew", "ir", ifeat, F, Mi, 1))
        print(f"head chain forward (230->200x4->1, saves): {t:.1f} us = {head_fl / t / 1e6:.1f} TFLOP/s")
        r_out, r_acts, r_layers = eng.dense_forward("reward_model", "rew", "ir", ifeat, F, Mi, 1)
        difeat = torch.empty(Mi, F, device="cuda")
        t = timed(lambda: eng.mlp_backward(Mi, d_r, 1, r_layers, r_acts + [None], None, din0=difeat, ld0=F, w0=F))
        print(f"head chain backward (dgrad to the features, no dpre out): {t:.1f} us = {head_fl / t / 1e6:.1f} TFLOP/s")
        dpre = [torch.zeros(Mi, 200, device="cuda") for _ in range(4)] + [d_r.view(Mi, 1)]
        t = timed(lambda: eng.mlp_backward(Mi, d_r, 1, r_layers, r_acts + [None], dpre[:-1] + [None]))
        print(f"head chain backward (critic form: dpre out, no din): {t:.1f} us")
        torch.cuda.synchronize()
        keep[mode] = [r_out.clone()] + [x.clone() for x in r_acts] + [difeat.clone()] + [x.clone() for x in dpre[:-1]]
    # what the epilogue stores cost: the same forward chain without saved activations
    layers = eng._dense_spec("reward_model", "rew", F, 1)
    out1 = torch.empty(Mi, 1, device="cuda")
    for mode in (0, 1):
        lib.bd_mlp_set_tall(mode)
        t = timed(lambda: eng.mlp_forward(Mi, ifeat, F, F, layers, None, out1, 1))
        print(f"mode {mode}: head chain forward WITHOUT saves: {t:.1f} us = {head_fl / t / 1e6:.1f} TFLOP/s")
    for i, (x, y) in enumerate(zip(keep[0], keep[1])):
        print(f"  tensor {i}: max |tall - per-tile| = {(x - y).abs().max().item():.3e} (scale {x.abs().max().item():.3e})")
    lib.bd_mlp_set_tall(-1)

    def wg():
        wc = eng._wbatch["critic"]
        eng._dense_wgrads(wc, "critic", Mi, dpre, ifeat, F, r_acts, [F] + [200] * 4 + [1])
        wc.run()
    t = timed(wg)
    print(f"critic weight gradients (grouped): {t:.1f} us = {head_fl / t / 1e6:.1f} TFLOP/s")


if __name__ == "__main__":
    main()
