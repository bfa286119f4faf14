"""Diagnostic: phase timeline of one step of the K-split cluster observe scan (csrc/observe_ksplit.hip), member 0 of tile 0
(needs `make -C big_dreamer_amd/csrc stamps`; run with BD_LIB=big_dreamer_amd/libbd_stamps.so)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from big_dreamer_amd import _cabi, synth
from big_dreamer_amd.engine import DreamerEngine
d = synth.CONFIG2
eng = DreamerEngine(d, None, "cuda", params=synth.make_params(d, 0))
batch = {k: torch.as_tensor(v).cuda() for k, v in synth.make_batch(d, 0).items()}
eng.pipeline = False
for _ in range(3):
    eng.train_step(batch, None)
torch.cuda.synchronize()
fn = _cabi.lib.bd_debug_kstamps; fn.restype = ctypes.c_int
out = (ctypes.c_ulonglong * 64)()
assert fn(out) == 0
st = np.array(out[:12], dtype=np.int64)
names = ["A state/action frags + prefetches (+barrier)", "F1 x_c (wave 0) (+barrier)", "F2 gate partials + sc1 stores + publish",
         "wait_all #1", "F3 reduce + gates (+2 barriers)", "F4 q partials + publish", "wait_all #2", "F5 reduce + ELU (+2 barriers)",
         "F6 head partials + publish", "wait_all #3", "F7 reduce + sample (+2 barriers)"]
tot = st[11] - st[0]
print(f"K-split observe FORWARD step (member 0): {tot} cycles")
for i, n in enumerate(names):
    print(f"  {n:48s} {st[i+1]-st[i]:8d}  {100.0*(st[i+1]-st[i])/tot:5.1f} %")
st = np.array(out[16:28], dtype=np.int64)
names = ["B1 (dm, draw) + prefetches (+barrier)", "B2 dQ_c (wave 0) (+barrier)", "B3 dh partials + publish", "wait_all #1",
         "B4 reduce + gate grads (+2 barriers)", "B5 (DX, DH) partials + publish", "wait_all #2", "B6 reduce + dE, carry (+2 barriers)",
         "B7 ds partials + publish", "wait_all #3", "B8 reduce + mask (+barrier)"]
tot = st[11] - st[0]
print(f"K-split observe BACKWARD step (member 0): {tot} cycles")
for i, n in enumerate(names):
    print(f"  {n:48s} {st[i+1]-st[i]:8d}  {100.0*(st[i+1]-st[i])/tot:5.1f} %")
