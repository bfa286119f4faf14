#!/usr/bin/env python3
"""Weight-stationary dense layer (tools/probes/dense_ws.hip) against torch and against the chain kernel (bd_mlp_forward):
correctness of the forward and dgrad forms, then time per 200x200 layer at the imagination's 34 300 rows."""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from big_dreamer_amd import _cabi as cabi  # noqa: E402
from big_dreamer_amd.categorical import _pack  # noqa: E402

lib = cabi.lib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "probes"))
import probes  # noqa: E402  (tools/probes/libbd_probes.so: `make -C tools/probes`)


def ws(x, pk, bias, saved, N, act, out):
    M, K = x.shape
    probes.check(probes.lib.bd_dense_ws(x.data_ptr(), K, pk.data_ptr(), bias.data_ptr() if bias is not None else None,
                               saved.data_ptr() if saved is not None else None, M, N, K, act, out.data_ptr(), N, cabi.stream()))


def chain(x, pks, biases, outs, act_last):
    M, K = x.shape
    a = cabi.MlpFwdArgs()
    a.M, a.in0, a.ld0, a.w0 = M, x.data_ptr(), K, K
    a.in1, a.ld1, a.w1 = None, 0, 0
    a.n_layers = len(pks)
    for i, (pk, b) in enumerate(zip(pks, biases)):
        last = i == len(pks) - 1
        a.layer[i] = cabi.Layer(pk.data_ptr(), b.data_ptr(), 200, 200, cabi.ACT_ELU if (not last or act_last) else cabi.ACT_NONE,
                                outs[i].data_ptr() if not last else None)
    a.out, a.ldo = outs[-1].data_ptr(), 200
    cabi.check(lib.bd_mlp_forward(C.byref(a), cabi.stream()))


def timed(fn, n=20):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def main():
    torch.manual_seed(0)
    M, K, N = 34300, 200, 200
    x = torch.randn(M, K, device="cuda")
    Ws = [torch.randn(N, K, device="cuda") / K ** 0.5 for _ in range(3)]
    bs = [0.1 * torch.randn(N, device="cuda") for _ in range(3)]
    pks = [_pack(w, False) for w in Ws]
    out = torch.empty(M, N, device="cuda")
    ws(x, pks[0], bs[0], None, N, cabi.ACT_ELU, out)
    ref = torch.nn.functional.elu(x @ Ws[0].t() + bs[0])
    print("forward max err", (out - ref).abs().max().item())
    # dgrad form: d pre_prev = (dpre W) * ELU'(saved_prev); packed transpose of W (N, K) -> out width K
    dpre = torch.randn(M, N, device="cuda")
    saved = torch.nn.functional.elu(torch.randn(M, K, device="cuda"))
    pkt = _pack(Ws[0], True)
    dout = torch.empty(M, K, device="cuda")
    ws(dpre, pkt, None, saved, K, 0, dout)
    refb = (dpre @ Ws[0]) * torch.where(saved > 0, torch.ones_like(saved), saved + 1)
    print("dgrad max err", (dout - refb).abs().max().item(), "scale", refb.abs().max().item())
    # ragged M
    Mr = 1000 + 7
    o2 = torch.empty(Mr, N, device="cuda")
    ws(x[:Mr], pks[0], bs[0], None, N, cabi.ACT_ELU, o2)
    print("ragged max err", (o2 - ref[:Mr]).abs().max().item())
    outs = [torch.empty(M, N, device="cuda") for _ in range(3)]
    t_ws1 = timed(lambda: ws(x, pks[0], bs[0], None, N, cabi.ACT_ELU, out))
    t_ch1 = timed(lambda: chain(x, pks[:1], bs[:1], outs[:1], True))

    def ws3():
        ws(x, pks[0], bs[0], None, N, cabi.ACT_ELU, outs[0])
        ws(outs[0], pks[1], bs[1], None, N, cabi.ACT_ELU, outs[1])
        ws(outs[1], pks[2], bs[2], None, N, cabi.ACT_ELU, outs[2])

    t_ws3 = timed(ws3)
    t_ch3 = timed(lambda: chain(x, pks, bs, outs, True))
    fl = 2.0 * M * K * N
    print(f"1 layer : weight-stationary {t_ws1:.1f} us ({fl / t_ws1 / 1e6:.1f} TFLOP/s) | chain kernel {t_ch1:.1f} us ({fl / t_ch1 / 1e6:.1f} TFLOP/s)")
    print(f"3 layers: weight-stationary {t_ws3:.1f} us ({3 * fl / t_ws3 / 1e6:.1f} TFLOP/s) | chain kernel {t_ch3:.1f} us ({3 * fl / t_ch3 / 1e6:.1f} TFLOP/s)")


def mfma_peak():
    out = torch.zeros(4, device="cuda")
    for blocks, iters in ((256, 20000), (256, 100000), (512, 50000)):
        f = lambda: probes.check(probes.lib.bd_mfma_probe(blocks, iters, out.data_ptr(), cabi.stream()))
        us = timed(f, 5)
        flops = blocks * 16 * iters * 32 * 2048.0
        print(f"MFMA-only loop, {blocks} workgroups x 16 waves, {iters} iterations: {us / 1e3:.2f} ms -> {flops / us / 1e6:.1f} TFLOP/s "
              f"(nominal 157.3 at 2.4 GHz => {flops / us / 1e6 / 157.3 * 2.4:.2f} GHz sustained)")


if __name__ == "__main__":
    mfma_peak()
    main()
