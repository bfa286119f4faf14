#!/usr/bin/env python3
"""bd_gemm_nt (csrc/gemm.hip) against torch.mm (rocBLAS / hipBLASLt) on the decoder's K = 3200 dgrad GEMM, alone on the GPU."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from big_dreamer_amd import _cabi as cabi
lib = cabi.lib
M, N, K = 2450, 1024, 3200
A = torch.randn(M, K, device="cuda"); Bbuf = torch.randn(N * K + 4, device="cuda")
C = torch.zeros(M, N, device="cuda"); C2 = torch.zeros(M, N, device="cuda")
def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for off in (0, 1):
    B = Bbuf[off:off + N * K].view(N, K)
    us = timed(lambda: cabi.check(lib.bd_gemm_nt(A.data_ptr(), K, B.data_ptr(), K, C.data_ptr(), N, M, N, K, 0, cabi.stream())))
    ut = timed(lambda: torch.mm(A, B.t(), out=C2))
    err = float((C - C2).abs().max())
    print(f"B float offset {off}: bd_gemm_nt {us:.1f} us = {2.0*M*N*K/us/1e6:.1f} TFLOP/s | torch.mm {ut:.1f} us = {2.0*M*N*K/ut/1e6:.1f} TFLOP/s | max diff {err:.2e}")
