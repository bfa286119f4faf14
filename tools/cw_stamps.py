"""Diagnostic: phase timeline of the weight-stationary chain kernel (needs a -DBD_STAMPS build:
make -C big_dreamer_amd/csrc CXXFLAGS_EXTRA=-DBD_STAMPS LIB=../libbd_stamps.so; run with BD_LIB=...)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from big_dreamer_amd import _cabi, synth
from big_dreamer_amd.engine import DreamerEngine
d = synth.CONFIG2
eng = DreamerEngine(d, None, "cuda", params=synth.make_params(d, 0))
Mi, F = d.Hm * d.N, d.Be + d.S
ifeat = torch.randn(Mi, F, device="cuda")
_cabi.lib.bd_chain_ws_set_mode(1)
for _ in range(3):
    eng.dense_forward("reward_model", "rew", "ir", ifeat, F, Mi, 1)
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 64)()
fn = _cabi.lib.bd_debug_cwstamps; fn.restype = ctypes.c_int
assert fn(out) == 0
st = np.array(out[:], dtype=np.int64)
print("batch start -> input loaded:", st[1] - st[0], "| -> barrier:", st[2] - st[1], "(s_memtime ticks @100 MHz: x24 = core cycles)")
for l in range(5):
    b = 8 * l
    row = [f"layer {l}: since prev barrier -> sweep0 start {st[3+b]-st[2+b if l==0 else b]}"]
    row.append(f"sweep0 {st[4+b]-st[3+b]}")
    if st[5+b] > st[4+b]: row.append(f"epi0+gap {st[5+b]-st[4+b]} sweep1 {st[6+b]-st[5+b]}")
    last_sw = st[6+b] if st[6+b] > st[4+b] else st[4+b]
    row.append(f"last epilogue {st[7+b]-last_sw} barrier {st[8+b]-st[7+b]}")
    print("  ".join(row))
print("total batch:", st[8+32] - st[0], "ticks =", (st[8+32]-st[0])/100, "us")
