#!/usr/bin/env python3
"""Diagnostic: forward time of every stride-2 layer of the two conv stacks at the training batch (2450 images), this
library's gather-GEMMs (NHWC) against torch / MIOpen (NCHW)."""
import os
import sys

import torch
import torch.nn.functional as Fnn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from big_dreamer_amd import _cabi as cabi, conv  # noqa: E402

ENC = [(3, 32, 4, 64), (32, 64, 4, 31), (64, 128, 4, 14), (128, 256, 4, 6)]
DEC = [(128, 64, 5, 5), (64, 32, 6, 13), (32, 3, 6, 30)]
imgs = int(sys.argv[1]) if len(sys.argv) > 1 else 2450


def timeit(fn, n=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for Cin, Cout, k, size in ENC:
    x = torch.randn(imgs, Cin, size, size, device="cuda")
    w = torch.randn(Cout, Cin, k, k, device="cuda") * 0.1
    b = torch.zeros(Cout, device="cuda")
    xs = conv.to_nhwc(x)
    K = k * k * Cin
    wp = torch.zeros(cabi.packed_floats(Cout, K), device="cuda")
    conv.pack_matrix(w.permute(0, 2, 3, 1).contiguous().view(Cout, K), wp, Cout, K)
    OH = conv.conv_out(size, k)
    out = torch.zeros(imgs, OH, OH, Cout, device="cuda")
    t_ref = timeit(lambda: Fnn.elu(Fnn.conv2d(x, w, b, stride=2)))
    t_own = timeit(lambda: conv.pattern_f(xs, out, wp, b, imgs, size, size, Cin, k, Cout, cabi.ACT_ELU))
    gf = 2.0 * imgs * OH * OH * K * Cout / 1e9
    print(f"conv {Cin:4d}->{Cout:4d} k{k} {size:2d}->{OH:2d}: torch {t_ref:7.3f} ms  own {t_own:7.3f} ms  ({gf / t_own:6.1f} TFLOP/s own, "
          f"{gf / t_ref:6.1f} torch)")
    # dgrad of this conv (T pattern) vs torch
    if Cin >= 4:
        gy = torch.randn(imgs, Cout, OH, OH, device="cuda")
        packs = [torch.zeros(n, device="cuda") for n in conv.class_pack_floats(Cout, Cin, k)]
        conv.pack_classes(w.permute(0, 2, 3, 1).contiguous(), packs, Cout, Cin, k)
        gys = conv.to_nhwc(gy)
        gx = torch.zeros(imgs, size, size, Cin, device="cuda")
        t_ref = timeit(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [2, 2], [0, 0], [1, 1], False, [0, 0], 1,
                                                                   [True, False, False]))
        t_own = timeit(lambda: conv.pattern_t(gys, gx, packs, None, imgs, OH, OH, Cout, k, Cin, size, size, cabi.ACT_NONE))
        print(f"     dgrad: torch {t_ref:7.3f} ms  own {t_own:7.3f} ms  ({gf / t_own:6.1f} TFLOP/s own)")
for Cin, Cout, k, size in DEC:
    x = torch.randn(imgs, Cin, size, size, device="cuda")
    w = torch.randn(Cin, Cout, k, k, device="cuda") * 0.1
    b = torch.zeros(Cout, device="cuda")
    packs = [torch.zeros(n, device="cuda") for n in conv.class_pack_floats(Cin, Cout, k)]
    conv.pack_classes(w.permute(0, 2, 3, 1).contiguous(), packs, Cin, Cout, k)
    OH = conv.convT_out(size, k)
    xs = conv.to_nhwc(x)
    out = torch.zeros(imgs, OH, OH, Cout, device="cuda")
    t_ref = timeit(lambda: Fnn.elu(Fnn.conv_transpose2d(x, w, b, stride=2)))
    t_own = timeit(lambda: conv.pattern_t(xs, out, packs, b, imgs, size, size, Cin, k, Cout, OH, OH, cabi.ACT_ELU))
    gf = 2.0 * imgs * size * size * k * k * Cin * Cout / 1e9
    print(f"convT {Cin:4d}->{Cout:4d} k{k} {size:2d}->{OH:2d}: torch {t_ref:7.3f} ms  own {t_own:7.3f} ms  ({gf / t_own:6.1f} TFLOP/s own, "
          f"{gf / t_ref:6.1f} torch)")
    gy = torch.randn(imgs, Cout, OH, OH, device="cuda")
    gys = conv.to_nhwc(gy)
    Kt = k * k * Cout
    wp = torch.zeros(cabi.packed_floats(Cin, Kt), device="cuda")
    conv.pack_matrix(w.permute(0, 2, 3, 1).contiguous().view(Cin, Kt), wp, Cin, Kt)
    gx = torch.zeros(imgs, size, size, Cin, device="cuda")
    if Cout >= 4:
        t_ref = timeit(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [2, 2], [0, 0], [1, 1], True, [0, 0], 1,
                                                                   [True, False, False]))
        t_own = timeit(lambda: conv.pattern_f(gys, gx, wp, None, imgs, OH, OH, Cout, k, Cin, cabi.ACT_NONE))
        print(f"     dgrad: torch {t_ref:7.3f} ms  own {t_own:7.3f} ms  ({gf / t_own:6.1f} TFLOP/s own)")


# ---- weight gradients (bd_wgrad_grouped with a gathered window operand), one GEMM at a time, against MIOpen's ----
print("weight gradients:")
for Cin, Cout, k, size in ENC:
    OH = conv.conv_out(size, k)
    x = torch.randn(imgs, Cin, size, size, device="cuda")
    w = torch.randn(Cout, Cin, k, k, device="cuda") * 0.1
    gy = torch.randn(imgs, Cout, OH, OH, device="cuda")
    xs, gys = conv.to_nhwc(x), conv.to_nhwc(gy)
    dW = torch.zeros(Cout, k, k, Cin, device="cuda")
    db = torch.zeros(Cout, device="cuda")
    desc = [conv.wgrad_desc(gys.view(imgs * OH * OH, Cout), Cout, xs, imgs, OH, OH, size, size, Cin, k, dW, db)]
    t_own = timeit(lambda: conv.run_wgrad(desc))
    t_ref = timeit(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [2, 2], [0, 0], [1, 1], False, [0, 0], 1,
                                                               [False, True, False]))
    gf = 2.0 * imgs * OH * OH * k * k * Cin * Cout / 1e9
    print(f"  conv  {Cin:4d}->{Cout:4d} k{k} (N={Cout:3d}, K={k*k*Cin:4d}, M={imgs*OH*OH:8d}): torch {t_ref:7.3f} ms  own {t_own:7.3f} ms "
          f"({gf / t_own:6.1f} TFLOP/s own, {gf / t_ref:6.1f} torch)")
for Cin, Cout, k, size in DEC:
    OH = conv.convT_out(size, k)
    x = torch.randn(imgs, Cin, size, size, device="cuda")
    w = torch.randn(Cin, Cout, k, k, device="cuda") * 0.1
    gy = torch.randn(imgs, Cout, OH, OH, device="cuda")
    xs, gys = conv.to_nhwc(x), conv.to_nhwc(gy)
    dW = torch.zeros(Cin, k, k, Cout, device="cuda")
    desc = [conv.wgrad_desc(xs.view(imgs * size * size, Cin), Cin, gys, imgs, size, size, OH, OH, Cout, k, dW, None)]
    t_own = timeit(lambda: conv.run_wgrad(desc))
    t_ref = timeit(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [2, 2], [0, 0], [1, 1], True, [0, 0], 1,
                                                               [False, True, False]))
    gf = 2.0 * imgs * size * size * k * k * Cin * Cout / 1e9
    print(f"  convT {Cin:4d}->{Cout:4d} k{k} (N={Cin:3d}, K={k*k*Cout:4d}, M={imgs*size*size:8d}): torch {t_ref:7.3f} ms  own {t_own:7.3f} ms "
          f"({gf / t_own:6.1f} TFLOP/s own, {gf / t_ref:6.1f} torch)")
