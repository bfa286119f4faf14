"""Diagnostic: phase timeline of one imagination step from s_memtime stamps (needs a -DBD_STAMPS build:
make -C big_dreamer_amd/csrc CXXFLAGS_EXTRA=-DBD_STAMPS LIB=../libbd_stamps.so; run with BD_LIB=...)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from big_dreamer_amd import _cabi, synth  # noqa: E402
from big_dreamer_amd.engine import DreamerEngine  # noqa: E402

d = synth.CONFIG2
eng = DreamerEngine(d, None, "cuda", params=synth.make_params(d, 0))
batch = {k: torch.as_tensor(v).cuda() for k, v in synth.make_batch(d, 0).items()}
for _ in range(3):
    eng.train_step(batch, None)
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 64)()
fn = _cabi.lib.bd_debug_stamps
fn.restype = ctypes.c_int
assert fn(out) == 0
st = np.array(out[:15], dtype=np.int64)
names = ["actor L0", "barrier", "actor L1-3 (+3 barriers)", "actor out+sample", "barrier", "entropy (+2 barriers)",
         "embed", "barrier", "GRU", "barrier", "prior hidden", "barrier", "prior out+sample", "barrier"]
tot = st[14] - st[0]
print(f"one imagination step of workgroup {os.environ.get('BD_STAMP_WG', 0)}: {tot} ticks (s_memtime @100MHz => {tot/100:.1f} us)")
for i, n in enumerate(names):
    dt = st[i + 1] - st[i]
    print(f"  {n:28s} {dt:8d} ticks  {100.0 * dt / tot:5.1f} %")

fn3 = getattr(_cabi.lib, "bd_debug_dstamps", None)
if fn3 is not None:
    fn3.restype = ctypes.c_int
    out3 = (ctypes.c_ulonglong * 64)()
    assert fn3(out3) == 0
    for base, title in ((0, "actor out + sample (split-K dual head)"), (8, "prior out + sample (split-K dual head)")):
        st = np.array(out3[base:base + 6], dtype=np.int64)
        names3 = ["issue operand (noise) load", "contraction + reduce -> LDS", "barrier", "wait vmcnt(0)", "element math + stores"]
        print(f"{title}: {st[5] - st[0]} cycles   [drain older ops: {out3[base + 6] - st[0]}, operand load alone: "
              f"{out3[base + 7] - out3[base + 6]}]")
        for i, n in enumerate(names3):
            print(f"  {n:40s} {st[i + 1] - st[i]:8d}")
        si = np.array(out3[base + 16:base + 22], dtype=np.int64)
        print("    inside the contraction: to first stamp", si[0] - st[1], "| loads+MFMAs", si[1] - si[0], "| partials->LDS", si[2] - si[1],
              "| barrier", si[3] - si[2], "| reduce", si[4] - si[3], "| plain stores", si[5] - si[4], "| return", st[2] - si[5])

    sd = np.array(out3[32:38], dtype=np.int64)
    print("actor layer 2 (200x200), wave 0 (2 blocks): bias+pre", sd[1] - sd[0], "| K loop", sd[2] - sd[1], "| epilogue", sd[3] - sd[2],
          "| return", sd[4] - sd[3], "| barrier", sd[5] - sd[4])

# ---- cluster observe scan (forward), member 0 of tile 0, step 5 ----
fn2 = getattr(_cabi.lib, "bd_debug_cstamps", None)
if fn2 is not None:
    fn2.restype = ctypes.c_int
    out2 = (ctypes.c_ulonglong * 64)()
    assert fn2(out2) == 0
    st = np.array(out2[:11], dtype=np.int64)
    names = ["A masked state/action", "B embed (full)", "C GRU blocks, split-K", "reduce + gate math + sc1 stores",
             "publish (vmcnt0+barrier+flag)", "plain stores", "wait_all", "gather payload", "D posterior hidden (full)",
             "E posterior out (split-K)"]
    tot = st[10] - st[0]
    print(f"one observe step (cluster fwd, member 0): {tot} cycles")
    for i, n in enumerate(names):
        dt = st[i + 1] - st[i]
        print(f"  {n:34s} {dt:8d}  {100.0 * dt / tot:5.1f} %")

# ---- imagination backward, workgroup 0, step t = 3 ----
outb = (ctypes.c_ulonglong * 64)()
assert fn(outb) == 0
sb_ = np.array(outb[20:27], dtype=np.int64)
if sb_[6] > sb_[0] > 0:
    namesb = ["1 prior sample -> (mean, raw)", "2 prior hidden", "3 d belief, GRU gates", "4 through W_ih / W_hh", "5 embed layer, actor output grads",
              "6 actor MLP backward"]
    totb = sb_[6] - sb_[0]
    print(f"one imagination BACKWARD step of workgroup 0: {totb} cycles")
    for i, n in enumerate(namesb):
        print(f"  {n:36s} {sb_[i + 1] - sb_[i]:8d}  {100.0 * (sb_[i + 1] - sb_[i]) / totb:5.1f} %")

# ---- cluster observe scan (backward), member 0 of tile 0, step 5 ----
if fn2 is not None:
    outc = (ctypes.c_ulonglong * 64)()
    assert fn2(outc) == 0
    cb = np.array(outc[16:25], dtype=np.int64)
    if cb[8] > cb[0] > 0:
        namesc = ["1 sample / softplus -> (mean, raw)", "2 d q", "3 d belief, GRU gate grads (full)", "4 W_ih / W_hh, member's blocks (split-K)",
                  "publish", "wait_all", "gather payload", "5 d state through the embed layer"]
        totc = cb[8] - cb[0]
        print(f"one observe BACKWARD step (cluster, member 0): {totc} cycles")
        for i, n in enumerate(namesc):
            print(f"  {n:44s} {cb[i + 1] - cb[i]:8d}  {100.0 * (cb[i + 1] - cb[i]) / totc:5.1f} %")
