"""Diagnostic: phase timeline of one step of the Categorical observe scan (needs `make -C big_dreamer_amd/csrc stamps`;
run with BD_LIB=big_dreamer_amd/libbd_stamps.so)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from big_dreamer_amd import _cabi, synth
from big_dreamer_amd.engine import DreamerEngine
d = synth.CONFIG5_STATE
eng = DreamerEngine(d, None, "cuda", params=synth.make_params(d, 0))
batch = {k: torch.as_tensor(v).cuda() for k, v in synth.make_batch(d, 0).items()}
os.environ["BD_PIPELINE"] = "0"
for _ in range(2):
    eng.train_step(batch, None)
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 64)()
fn = _cabi.lib.bd_debug_catstamps; fn.restype = ctypes.c_int
assert fn(out) == 0
st = np.array(out[:10], dtype=np.int64)
names = ["A mask+action frags (+barrier)", "state gather", "sv_s one-hot write (+barrier)", "B embed (+barrier)", "C GRU (+barrier)",
         "D posterior hidden (+barrier)", "E logits + sample", "indices out (+barrier)", "feat one-hot write"]
tot = st[9] - st[0]
print(f"one Categorical observe step (workgroup 0): {tot} cycles")
for i, n in enumerate(names):
    print(f"  {n:36s} {st[i+1]-st[i]:8d}  {100.0*(st[i+1]-st[i])/tot:5.1f} %")

out = (ctypes.c_ulonglong * 64)()
assert fn(out) == 0
st = np.array(out[16:22], dtype=np.int64)
names = ["mask (+barrier)", "1 head backward (4 chunks) -> d hidden acc", "dQ epilogue (+barrier)", "3 d belief, gate grads (+barrier)",
         "4 GRU dgrad (+barrier)"]
tot = st[5] - st[0]
print(f"one Categorical observe BACKWARD step (workgroup 0): {tot} cycles")
for i, n in enumerate(names):
    print(f"  {n:44s} {st[i+1]-st[i]:8d}  {100.0*(st[i+1]-st[i])/tot:5.1f} %")
