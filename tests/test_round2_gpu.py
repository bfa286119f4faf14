"""GPU: round-2 additions -- self-launching multi-rank bench, sticky cluster-scan error word, lazy log dicts on the
drop-in surface, update_belief_and_act pinned to the reference, checkpoint save / load with optimiser state."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from big_dreamer_amd import synth
from tests.helpers import assert_close, load_golden

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _dev(dct):
    return {k: torch.as_tensor(v).cuda().contiguous() for k, v in dct.items()}


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it: the parent spawns two rank processes (here both on the one
    GPU of the box, gloo transport -- RCCL refuses two ranks per device), rank 0 prints the one JSON line."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
           "--no-cpu-baseline", "--backend", "gloo", "--same-device"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["rccl_ranks"] == 2 and r["config"]["global_batch"] == 100
    assert r["steps"] == 4 and r["value"] > 0 and np.isfinite(r["losses"]["model_loss"])
    assert r["roofline"]["frac"] > 0 and "cpu_baseline" not in r


def test_cluster_timeout_is_sticky_until_read():
    """A peer time-out in the FORWARD cluster scan must still be reported after the BACKWARD launch (which re-zeroes the
    flags) and by the engine's log fetch; reading clears it."""
    from big_dreamer_amd import _cabi as cabi
    from big_dreamer_amd.engine import DreamerEngine
    d = synth.CONFIG2
    eng = DreamerEngine(d, None, "cuda", params=synth.make_params(d, 0))
    batch, noise = _dev(synth.make_batch(d, 0)), _dev(synth.make_noise(d, 0))
    assert eng._cluster_ok(d.B)
    eng.train_step(batch, noise)                      # healthy step: no error
    eng.cluster_status(d.B)
    # forward scan with a spin limit of one poll: some member is certain to find a peer's flag not yet at the epoch
    N, T, B = d.N, d.T, d.B
    emb, pre = eng.encode(batch["observations"][1:].reshape(N, d.O), N)
    cabi.lib.bd_observe_cluster_set_spin_limit(1)
    try:
        eng.observe(batch["actions"][:-1], batch["nonterminals"][:-1], pre, noise["obs_post"],
                    torch.zeros(B, d.Be, device="cuda"), torch.zeros(B, d.S, device="cuda"), T, B)
        torch.cuda.synchronize()
    finally:
        cabi.lib.bd_observe_cluster_set_spin_limit(0)
    # a whole healthy train step afterwards (forward + backward cluster launches): the word must survive it ...
    with pytest.raises(RuntimeError, match="timed out"):
        eng.train_step(batch, noise)                  # ... and the step's own log fetch reports it
    eng.train_step(batch, noise)                      # reported once, then clear
    eng.cluster_status(d.B)
    # the same through the C-ABI status call
    cabi.lib.bd_observe_cluster_set_spin_limit(1)
    try:
        eng.observe(batch["actions"][:-1], batch["nonterminals"][:-1], pre, noise["obs_post"],
                    torch.zeros(B, d.Be, device="cuda"), torch.zeros(B, d.S, device="cuda"), T, B)
        torch.cuda.synchronize()
    finally:
        cabi.lib.bd_observe_cluster_set_spin_limit(0)
    eng.observe(batch["actions"][:-1], batch["nonterminals"][:-1], pre, noise["obs_post"],
                torch.zeros(B, d.Be, device="cuda"), torch.zeros(B, d.S, device="cuda"), T, B)    # healthy launch after it
    with pytest.raises(RuntimeError, match="forward"):
        eng.cluster_status(d.B)
    eng.cluster_status(d.B)


def test_lazy_logs_equal_eager_logs_and_keep_the_pipeline_bit_identical(monkeypatch):
    """Dreamer.train_step returns a lazy mapping: six un-read pipelined steps, then every dict read back, must give
    exactly the values (and final weights) of six eagerly synchronised steps."""
    from big_dreamer_amd.engine import DreamerEngine, LazyLogs
    d = synth.SMALL
    P = synth.make_params(d, 3)
    batches = [_dev(synth.make_batch(d, 10 + i)) for i in range(6)]
    noises = [_dev(synth.make_noise(d, 10 + i)) for i in range(6)]
    eager = DreamerEngine(d, None, "cuda", params=P)
    want = [eager.train_step(b, n) for b, n in zip(batches, noises)]
    lazy = DreamerEngine(d, None, "cuda", params=P)
    got = [lazy.train_step(b, n, sync_logs="lazy") for b, n in zip(batches, noises)]
    assert all(isinstance(g, LazyLogs) and g._vals is None for g in got), "a lazy dict was resolved by queueing later steps"
    got[-1]["weight_update_per_sec"] = 1.0            # the loop writes into the dict (src/main.py:108)
    for i in (5, 0, 3, 1, 2, 4):                      # any read order
        for k, v in want[i].items():
            assert got[i][k] == v, (i, k, got[i][k], v)
    assert "weight_update_per_sec" in dict(got[-1]) and set(want[0]) <= set(got[0].keys())
    torch.cuda.synchronize()
    for g in ("model", "actor", "critic"):
        assert torch.equal(eager.groups[g].flat, lazy.groups[g].flat)
    # more un-read steps than the record ring holds: the oldest dicts are resolved before their record is recycled
    more = [lazy.train_step(batches[i % 6], noises[i % 6], sync_logs="lazy") for i in range(11)]
    ref = [eager.train_step(batches[i % 6], noises[i % 6]) for i in range(11)]
    for a, b in zip(more, ref):
        assert a["model_loss"] == b["model_loss"] and a["value_loss"] == b["value_loss"]


@pytest.mark.parametrize("case", ["b1_explore", "b10_eval"])
def test_update_belief_and_act_matches_the_reference(case):
    """Planet.update_belief_and_act (src/planet.py:370-403) x3 consecutive calls at B=1 (explore) and B=10 (EnvBatcher
    eval) on the reference's own draws: belief, posterior state and action of every call (tests/golden/act.npz)."""
    from big_dreamer_amd.config import load_config
    from big_dreamer_amd.dreamer import Dreamer
    g = load_golden("act")
    B, explore, seed = (int(x) for x in g[f"{case}.meta"])
    d = synth.CONFIG2

    class Env:
        action_size, observation_size = d.A, d.O

        def __init__(self):
            self.got = []
            if B > 1:
                self.n, self.envs = B, [None] * B          # what marks an EnvBatcher (src/env.py:343)

        def step(self, a):
            self.got.append(a.numpy().copy())
            return None, 0.0, False

    agent = Dreamer(load_config(["experience_size=100"]), Env())
    P = synth.make_params(d, seed)
    for mod in ("transition_model", "observation_model", "reward_model", "encoder", "actor", "critic", "critic_target"):
        getattr(agent, mod).load_state_dict({k: torch.from_numpy(v) for k, v in P[mod].items()})
    env = Env()
    ns = synth.NoiseStream(seed)
    belief, state = torch.zeros(B, d.Be).cuda(), torch.zeros(B, d.S).cuda()
    action = torch.zeros(B, d.A).cuda()
    for i in range(3):
        nz = {"prior": ns.normal((B, d.S)), "post": ns.normal((B, d.S)), "action": ns.normal((B, d.A)),
              "entropy": ns.normal((d.n_entropy, B, d.A))}
        if explore:
            nz["explore"] = ns.normal((B, d.A))
        belief, state, action, _, _, _ = agent.update_belief_and_act(env, belief, state, action,
                                                                    torch.from_numpy(g[f"{case}.obs"][i]),
                                                                    explore=bool(explore), _noise=_dev(nz))
        assert_close(f"{case}.belief{i}", belief.cpu().numpy(), g[f"{case}.belief{i}"], 2e-5, 2e-5)
        assert_close(f"{case}.state{i}", state.cpu().numpy(), g[f"{case}.state{i}"], 2e-5, 2e-5)
        assert_close(f"{case}.action{i}", action.cpu().numpy(), g[f"{case}.action{i}"], 2e-5, 2e-5)
        assert_close(f"{case}.env_action{i}", env.got[-1], g[f"{case}.env_action{i}"], 2e-5, 2e-5)


@pytest.mark.parametrize("discount", [False, True])
def test_checkpoint_roundtrip_restores_optimiser_state(tmp_path, discount):
    """save() -> load(): weights, Adam moments and step counts of all three optimisers; the resumed agent's next
    train step is bit-identical to the original's.  discount: use_discount=True -- the discount head is the tail of the
    world-model optimiser (src/dreamer.py:167-169), so its weights must travel with the moments."""
    from big_dreamer_amd.config import load_config
    from big_dreamer_amd.dreamer import Dreamer
    d = synth.TINY_DISCOUNT if discount else synth.SMALL

    class Env:
        action_size, observation_size = d.A, d.O

    ov = [f"belief_size={d.Be}", f"state_size={d.S}", f"hidden_size={d.Hd}", f"embedding_size={d.E}", f"batch_size={d.B}",
          f"seq_len={d.L}", f"planning_horizon={d.H}", "experience_size=100"] + (["use_discount=true"] if discount else [])
    torch.manual_seed(5)
    a = Dreamer(load_config(ov), Env())
    batches = [_dev(synth.make_batch(d, 20 + i)) for i in range(4)]
    noises = [_dev(synth.make_noise(d, 20 + i)) for i in range(4)]
    for i in range(3):
        a.engine.train_step(batches[i], noises[i])
    a.update_critic()
    path = str(tmp_path / "ckpt.pth")
    a.save(path)
    ck = torch.load(path, map_location="cpu", weights_only=True)
    assert set(ck) >= {"transition_model", "observation_model", "reward_model", "encoder", "model_optimizer"}
    # the reference's own optimiser accepts the saved state (src/planet.py:114)
    params = [torch.nn.Parameter(torch.zeros_like(v["exp_avg"])) for v in ck["model_optimizer"]["state"].values()]
    torch.optim.Adam(params, lr=1e-3).load_state_dict(ck["model_optimizer"])
    torch.manual_seed(99)                                   # different initial weights: everything must come from the file
    b = Dreamer(load_config(ov + [f"models={path}"]), Env())
    for grp in ("model", "actor", "critic"):
        ga, gb = a.engine.groups[grp], b.engine.groups[grp]
        assert gb.step == ga.step == 3
        assert torch.equal(ga.flat, gb.flat) and torch.equal(ga.m, gb.m) and torch.equal(ga.v, gb.v)
    assert torch.equal(a.engine.groups["critic_target"].flat, b.engine.groups["critic_target"].flat)
    if discount:
        assert "discount_model" in ck and any(m == "discount_model" for m, _, _ in a.engine.groups["model"].specs)
    la = a.engine.train_step(batches[3], noises[3])
    lb = b.engine.train_step(batches[3], noises[3])
    assert la == lb
    for grp in ("model", "actor", "critic"):
        assert torch.equal(a.engine.groups[grp].flat, b.engine.groups[grp].flat)
    if not discount:
        # torch's Optimizer.load_state_dict adopts the checkpoint's lr / eps / weight_decay (src/planet.py:114): so does
        # the engine -- an agent configured with another learning rate continues with the checkpoint's
        with pytest.warns(UserWarning, match="adopting the checkpoint's hyper-parameters"):
            c = Dreamer(load_config(ov + [f"models={path}", "model_learning_rate=0.5"]), Env())
        assert c.engine._opt_over["model"]["lr"] == a.engine.hp["model_learning_rate"]
        lc = c.engine.train_step(batches[3], noises[3])
        assert lc == la and torch.equal(c.engine.groups["model"].flat, a.engine.groups["model"].flat)


def test_non_contiguous_operands_are_rejected():
    from big_dreamer_amd import _cabi as cabi
    x = torch.zeros(8, 6, device="cuda")
    cabi.ptr(x)
    cabi.ptr(x[:, :3])                                  # row-strided view with unit inner stride: what the kernels take
    with pytest.raises(ValueError, match="non-contiguous"):
        cabi.ptr(x.t())
    with pytest.raises(ValueError, match="non-contiguous"):
        cabi.ptr(x[:, ::2])
    with pytest.raises(ValueError, match="non-contiguous"):
        cabi.ptr(torch.zeros(4, 5, 6, device="cuda").permute(1, 0, 2))


def test_tall_chain_matches_the_per_tile_chain():
    """csrc/mlp.hip, tall form of bd_mlp_forward / bd_mlp_backward (48-row workgroups, balanced (row tile, column block)
    pairs, transposed accumulators, in-place LDS image; the default for M >= 8192 rows): outputs, saved activations,
    d/d features (plain and accumulating) and pre-activation gradients of the head chain against the 16-row kernels, on a
    row count that leaves a ragged last workgroup.  Same fp32 MFMA products, same summation order within a dot product."""
    from big_dreamer_amd import _cabi as cabi
    from big_dreamer_amd.engine import DreamerEngine
    d = synth.CONFIG2
    eng = DreamerEngine(d, None, "cuda", params=synth.make_params(d, 4))
    Mi, F = d.Hm * d.N + 7, d.Be + d.S
    g = torch.Generator(device="cuda").manual_seed(2)
    ifeat = torch.randn(Mi, F, device="cuda", generator=g)
    d_r = torch.randn(Mi, device="cuda", generator=g)
    seed_din = torch.randn(Mi, F, device="cuda", generator=g)
    keep = {}
    try:
        for mode in (0, 1):
            cabi.lib.bd_mlp_set_tall(mode)
            r_out, r_acts, r_layers = eng.dense_forward("reward_model", "rew", f"tt{mode}", ifeat, F, Mi, 1)
            difeat = torch.zeros(Mi, F, device="cuda")
            dacc = seed_din.clone()
            dpre = [torch.zeros(Mi, d.Hd, device="cuda") for _ in range(4)]
            eng.mlp_backward(Mi, d_r, 1, r_layers, r_acts + [None], dpre + [None], din0=difeat, ld0=F, w0=F)
            eng.mlp_backward(Mi, d_r, 1, r_layers, r_acts + [None], None, din0=dacc, ld0=F, w0=F, accumulate=True)
            torch.cuda.synchronize()
            keep[mode] = [r_out.clone()] + [x.clone() for x in r_acts] + [difeat, dacc] + dpre
    finally:
        cabi.lib.bd_mlp_set_tall(-1)
    for i, (x, y) in enumerate(zip(keep[0], keep[1])):
        scale = float(x.abs().max())
        assert float((x - y).abs().max()) <= 2e-6 * max(1.0, scale), (i, float((x - y).abs().max()), scale)


def test_tall_chain_with_one_hot_segment_matches_the_per_tile_chain():
    """Categorical latents: the dense heads read [h; one-hot s] and bd_mlp_forward takes the state as class indices (layer
    0 = contraction over h + a gather of D rows of the transposed weights).  The tall form builds that gather sum in its
    LDS image between layer 0's sweep and epilogue; outputs and saved activations against the 16-row kernels."""
    from big_dreamer_amd import _cabi as cabi
    from big_dreamer_amd.engine import DreamerEngine
    d = synth.CONFIG5_STATE            # 32 x 32 latents, 200-wide heads
    eng = DreamerEngine(d, None, "cuda", params=synth.make_params(d, 5))
    Mi, F = 9000 + 5, d.Be + d.S
    g = torch.Generator(device="cuda").manual_seed(4)
    sidx = torch.randint(0, d.cat_C, (Mi, d.cat_D), device="cuda", generator=g).to(torch.uint8)
    feat = torch.zeros(Mi, F, device="cuda")
    feat[:, :d.Be] = torch.randn(Mi, d.Be, device="cuda", generator=g)
    keep = {}
    try:
        for mode in (0, 1):
            cabi.lib.bd_mlp_set_tall(mode)
            out, acts, _ = eng.dense_forward("reward_model", "rew", f"g{mode}", feat, F, Mi, 1, sidx=sidx)
            torch.cuda.synchronize()
            keep[mode] = [out.clone()] + [x.clone() for x in acts]
    finally:
        cabi.lib.bd_mlp_set_tall(-1)
    for i, (x, y) in enumerate(zip(keep[0], keep[1])):
        scale = float(x.abs().max())
        assert scale > 0 and float((x - y).abs().max()) <= 2e-6 * max(1.0, scale), (i, float((x - y).abs().max()), scale)


def test_exact_math_build():
    """The -DBD_EXACT_MATH build (libm-grade ELU / sigmoid / tanh / softplus in the epilogues; `make exact`, built by
    __graft_entry__.build()) loaded through BD_LIB in a fresh process: two train steps of the `small` case against the
    oracle.  With exact activations the beliefs agree to summation-order noise (1e-6), weights after Adam to 2e-6."""
    lib = os.path.join(ROOT, "big_dreamer_amd", "libbigdreamer_hip_exact.so")
    assert os.path.exists(lib), "libbigdreamer_hip_exact.so missing: run __graft_entry__.build() (make exact)"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "exact_worker.py")], capture_output=True, text=True,
                         timeout=600, cwd=ROOT, env=dict(os.environ, BD_LIB=lib))
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("EXACT_RESULT ")][-1][len("EXACT_RESULT "):])
    assert res["belief"] < 2e-6 and res["weight"] < 2e-6 and res["log"] < 1e-5, res


def test_use_discount_on_the_surface():
    """use_discount=True through Dreamer(params, env): the discount head exists with the reference's state_dict names, joins
    the model optimiser last (src/dreamer.py:167-169) and train_step logs discount_loss (src/dreamer.py:290)."""
    from big_dreamer_amd.config import load_config
    from big_dreamer_amd.dreamer import Dreamer
    d = synth.TINY_DISCOUNT

    class Env:
        action_size, observation_size = d.A, d.O

    agent = Dreamer(load_config([f"belief_size={d.Be}", f"state_size={d.S}", f"hidden_size={d.Hd}", f"embedding_size={d.E}",
                                 f"batch_size={d.B}", f"seq_len={d.L}", f"planning_horizon={d.H}", "experience_size=100",
                                 "use_discount=true"]), Env())
    assert [n for n, _ in agent.discount_model.named_parameters()] == [n for n, _ in synth.param_shapes(d)["discount_model"]]
    assert agent.engine.groups["model"].specs[-1][0] == "discount_model"
    rep = synth.make_replay(d, rows=100, seed=2)
    for k, v in rep.items():
        getattr(agent.buffer, k)[:] = v
    agent.buffer.idx, agent.buffer.full = 0, True
    np.random.seed(0)
    logs = agent.train_step()
    assert "discount_loss" in set(logs.keys()) and all(np.isfinite(float(v)) for v in logs.values())
    assert 0.0 < float(logs["discount_loss"]) < 5.0


def test_second_engine_in_a_process_is_as_fast_as_the_first():
    """Round 2 measured a second engine built in one process 10-15 % slow and did not find the cause.  It is the stream ->
    hardware-queue assignment: torch hands out pool streams and HIP maps them onto a few hardware queues in creation order,
    so the second engine's three hot streams shared queues (tools/two_engines.py: 3.29 / 3.94 / 3.46 ms per step for engines
    1 / 2 / 3 with one stream set per engine, 3.29 / 3.23 / 3.21 with the process-wide set).  Engines now share one stream
    per (device, role, priority): the second engine runs on the SAME streams and is not slower than the first (3 %)."""
    import gc
    import time
    from big_dreamer_amd.engine import DreamerEngine
    from big_dreamer_amd.memory import ExperienceReplay
    d = synth.CONFIG2
    dev = torch.device("cuda", torch.cuda.current_device())

    def run():
        eng = DreamerEngine(d, None, dev, params=synth.make_params(d, 0))
        rep = synth.make_replay(d, rows=2000, seed=0)
        buf = ExperienceReplay(2000, d.A, 5, False, d.O, dev)
        for k, v in rep.items():
            getattr(buf, k)[:] = v
        buf.idx, buf.full = 0, True
        buf.sync_device()

        def step():
            o, a, r, n = buf.sample(d.B, d.L)
            eng.train_step({"observations": o, "actions": a, "rewards": r, "nonterminals": n}, None, sync_logs=False)

        for _ in range(10):
            step()
        eng.join()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):                       # best of three bursts: the comparison is about a persistent 10-20 % effect
            t0 = time.perf_counter()
            for _ in range(30):
                step()
            eng.join()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 30 * 1e3)
        streams = (eng._s_wm.cuda_stream, eng._s_bh.cuda_stream, eng._side.cuda_stream)
        del eng, buf
        gc.collect()
        return best, streams

    prev = torch.cuda.current_stream()
    torch.cuda.set_stream(torch.cuda.Stream())       # not the legacy null stream
    try:
        np.random.seed(0)
        (t1, s1), (t2, s2) = run(), run()
    finally:
        torch.cuda.set_stream(prev)
    assert s1 == s2, "the second engine did not reuse the process-wide pipeline streams"
    assert t2 <= 1.03 * t1, f"second engine {t2:.3f} ms/step against {t1:.3f} for the first"
