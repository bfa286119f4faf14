"""Diagnostic: stage timeline of the wide weight-gradient kernel (workgroup 0, wave 0, stages 10 and 11 of its row slab)
for the critic pass (5 GEMMs over 34 300 rows), alone on the GPU.  Needs the -DBD_STAMPS build (`make stamps`,
BD_LIB=big_dreamer_amd/libbd_stamps.so)."""
import ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from big_dreamer_amd import _cabi, synth
from big_dreamer_amd.engine import DreamerEngine
d = synth.CONFIG2
eng = DreamerEngine(d, None, "cuda", params=synth.make_params(d, 0))
Mi, F = d.Hm * d.N, d.Be + d.S
ifeat = torch.randn(Mi, F, device="cuda")
d_r = torch.randn(Mi, device="cuda")
r_out, r_acts, r_layers = eng.dense_forward("reward_model", "rew", "ir", ifeat, F, Mi, 1)
dpre = [torch.randn(Mi, 200, device="cuda") for _ in range(4)] + [d_r.view(Mi, 1)]
def wg():
    wc = eng._wbatch["critic"]
    eng._dense_wgrads(wc, "critic", Mi, dpre, ifeat, F, r_acts, [F] + [200] * 4 + [1])
    wc.run()
for _ in range(5): wg()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(30): wg()
torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 30 * 1e6
fl = 2.0 * Mi * (F * 200 + 3 * 200 * 200 + 200)
print(f"critic weight gradients (GEMM + reduce): {t:.1f} us = {fl / t / 1e6:.1f} TFLOP/s")
fn = getattr(_cabi.lib, "bd_debug_wstamps", None)
if fn is not None:
    fn.restype = ctypes.c_int
    out = (ctypes.c_ulonglong * 64)()
    assert fn(out) == 0
    st = np.array(out[:], dtype=np.int64)
    for k in (0, 8):
        print(f"stage {10 + k // 8}: issue DMA {st[k+1]-st[k]} | slices (MFMA) {st[k+2]-st[k+1]} | wait DMA {st[k+3]-st[k+2]} | barrier {st[k+4]-st[k+3]}   (cycles)")
    print("stage period:", st[8] - st[0])
