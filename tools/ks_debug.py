import sys, os, numpy as np, torch
sys.path.insert(0, "/root/repo")
from big_dreamer_amd import synth, _cabi as cabi
from big_dreamer_amd.engine import DreamerEngine
name = sys.argv[1] if len(sys.argv) > 1 else "TINY"
d = getattr(synth, name)
P, batch, noise = synth.make_params(d, 0), synth.make_batch(d, 0), synth.make_noise(d, 0)
dev = lambda x: torch.as_tensor(x).cuda().contiguous()
outs = {}
for mode in (3, 1, 0):
    cabi.check(cabi.lib.bd_observe_cluster_set_ksplit(mode))
    eng = DreamerEngine(d, None, "cuda", params=P)
    T, B, N = d.T, d.B, d.N
    obs = dev(batch["observations"])
    emb, pre = eng.encode(obs[1:].reshape(N, d.O), N)
    feat, qm, qs = eng.observe(dev(batch["actions"])[:-1], dev(batch["nonterminals"])[:-1], pre, dev(noise["obs_post"]),
                               torch.zeros(B, d.Be, device="cuda"), torch.zeros(B, d.S, device="cuda"), T, B)
    torch.cuda.synchronize()
    outs[mode] = (feat.cpu().numpy().reshape(T, B, -1).copy(), qm.cpu().numpy().copy(), {k: eng._buf[k].cpu().numpy().copy() for k in ("sv_x", "sv_gates", "sv_q", "sv_s")})
for m in (3, 1):
    f1, f0 = outs[m][0], outs[0][0]
    print(f"== mode {m} vs round-1 form")
    print("feat diff per t:", [round(float(np.abs(f1[t] - f0[t]).max()), 5) for t in range(d.T)])
    print("belief col diff t=0:", np.abs(f1[0][:, :d.Be] - f0[0][:, :d.Be]).max(0).round(4))
    print("row diff t=0:", np.abs(f1[0] - f0[0]).max(1).round(4))
    for k in ("sv_x", "sv_gates", "sv_q", "sv_s"):
        a, b = outs[m][2][k], outs[0][2][k]
        print(k, a.shape, "max diff", float(np.abs(a - b).max()))
