import os, sys, time, torch
sys.path.insert(0, "/root/repo")
from big_dreamer_amd import _cabi as cabi
lib = cabi.lib
sys.path.insert(0, "/root/repo/tools/probes")
import probes  # tools/probes/libbd_probes.so: `make -C tools/probes`
out = torch.zeros(4, device="cuda")
def timed(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
waves = int(os.environ.get("BD_PROBE_THREADS", "1024")) // 64
blocks, iters = 256, 50000
us = timed(lambda: probes.check(probes.lib.bd_mfma_probe(blocks, iters, out.data_ptr(), cabi.stream())))
flops = blocks * waves * iters * 32 * 2048.0
print(f"{waves} waves/CU: {flops / us / 1e6:.1f} TFLOP/s")
