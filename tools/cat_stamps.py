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
eng.pipeline = False
for _ in range(2):
    eng.train_step(batch, None)
torch.cuda.synchronize()
eng1 = DreamerEngine(d, None, "cuda", params=synth.make_params(d, 0))
eng1.use_obs_cluster = False          # the one-workgroup-per-tile scan (csrc/scan_cat.hip)
eng1.pipeline = False
for _ in range(2):
    eng1.train_step(batch, None)
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

# ---- the cluster scan (csrc/observe_cat_cluster.hip), member 0 of tile 0 ----
fn = getattr(_cabi.lib, "bd_debug_ccstamps", None)
if fn is not None and eng._cat_cluster(d.B):
    fn.restype = ctypes.c_int
    out = (ctypes.c_ulonglong * 64)()
    assert fn(out) == 0
    st = np.array(out[:14], dtype=np.int64)
    names = ["A mask + action frags (+barrier)", "A2 state gather + sv_s one-hot (+barrier)", "B embed (+barrier)",
             "C GRU own blocks, split-K (+barrier)", "reduce + gates + sc1 stores + publish + plain stores", "wait_all #1",
             "gather h' (+barrier)", "D posterior hidden (+barrier)", "E own logits split-K + reduce (+2 barriers)",
             "sample + sc1 + publish", "wait_all #2", "gather indices (+barrier)", "feat one-hot write"]
    tot = st[13] - st[0]
    print(f"one Categorical CLUSTER observe step (member 0): {tot} cycles")
    for i, n in enumerate(names):
        print(f"  {n:52s} {st[i+1]-st[i]:8d}  {100.0*(st[i+1]-st[i])/tot:5.1f} %")
    st = np.array(out[16:26], dtype=np.int64)
    names = ["1a carry x W_es^T own cols + stage logits (+2 barriers)", "1b Jacobian own factors (+barrier)",
             "1c d hidden K-slice + sc1 + publish", "wait_all #1", "all-reduce of d hidden + ELU' (+barrier)",
             "3 d belief + gate grads, full (+barrier)", "4 GRU dgrad own blocks + reduce + publish", "wait_all #2",
             "gather [carry | dE] (+barrier)"]
    tot = st[9] - st[0]
    print(f"one Categorical CLUSTER observe BACKWARD step (member 0): {tot} cycles")
    for i, n in enumerate(names):
        print(f"  {n:52s} {st[i+1]-st[i]:8d}  {100.0*(st[i+1]-st[i])/tot:5.1f} %")
