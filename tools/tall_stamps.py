"""Diagnostic: phase timeline of the tall chain kernel's workgroup 0 (needs the -DBD_STAMPS build: `make stamps`, run with
BD_LIB=big_dreamer_amd/libbd_stamps.so).  ROWS limits the rows (48 = one workgroup alone on the chip)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from big_dreamer_amd import _cabi, synth
from big_dreamer_amd.engine import DreamerEngine
d = synth.CONFIG2
eng = DreamerEngine(d, None, "cuda", params=synth.make_params(d, 0))
F = d.Be + d.S
Mi = int(os.environ.get("ROWS", d.Hm * d.N))
ifeat = torch.randn(d.Hm * d.N, F, device="cuda")
_cabi.lib.bd_mlp_set_tall(2)
for _ in range(3):
    eng.dense_forward("reward_model", "rew", "ir", ifeat, F, Mi, 1)
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 64)()
fn = _cabi.lib.bd_debug_tallstamps; fn.restype = ctypes.c_int
assert fn(out) == 0
st = np.array(out[:], dtype=np.int64)
u = lambda a, b: f"{(st[a] - st[b]) / 100:.2f}"
print(f"rows {Mi}: input load {u(1,0)} us, barrier {u(2,1)}")
for l in range(5):
    b = 4 * l
    print(f"  layer {l}: sweep {u(3+b, 2+b)}  barrier {u(4+b,3+b)}  epilogue {u(5+b,4+b)}  barrier {u(6+b,5+b)}")
print(f"  total {u(6+16, 0)} us")
