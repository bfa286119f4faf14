"""CPU tests: the C-ABI library loads and exports every symbol include/bigdreamer_hip.h declares, and the
host-side replay index logic matches the reference's golden draws (no compute calls without a GPU)."""
import os
import re

import numpy as np
import pytest

from tests.helpers import load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cabi_exports_every_declared_symbol():
    from big_dreamer_amd import _cabi
    hdr = open(os.path.join(ROOT, "include", "bigdreamer_hip.h")).read()
    declared = set(re.findall(r"^(?:int|size_t|const char\*)\s+(bd_\w+)\s*\(", hdr, flags=re.M))
    assert declared, "no declarations parsed from the header"
    assert declared == set(_cabi.EXPORTED), declared ^ set(_cabi.EXPORTED)
    for name in declared:
        assert hasattr(_cabi.lib, name)
    assert _cabi.lib.bd_version() >= 1
    assert _cabi.lib.bd_packed_floats(200, 230) == 13 * 15 * 256
    assert _cabi.lib.bd_reduce_ws_floats() > 0


def test_cabi_rejects_bad_arguments_without_gpu():
    """Argument validation happens before any launch: errors come back as codes + text, never exceptions."""
    from big_dreamer_amd import _cabi
    assert _cabi.lib.bd_pack_weights(None, 0, None) != 0
    assert b"bd_pack_weights" in _cabi.lib.bd_last_error()
    assert _cabi.lib.bd_sum(None, 0, None, 0, None, None) != 0
    with pytest.raises(RuntimeError):
        _cabi.check(_cabi.lib.bd_lambda_return_forward(None, None, 0, 0, 0.99, 0.95, None, None))


def test_replay_sample_indices_match_reference():
    """R0: same np.random draws -> same chunk start indices and rejections as src/memory.py:51-68."""
    from big_dreamer_amd import synth
    from big_dreamer_amd.memory import ExperienceReplay
    g = load_golden("replay")
    d = synth.TINY
    rows = 64
    rep = synth.make_replay(d, rows=rows, seed=3)
    for case, (idx, full) in {"partial": (40, False), "wrapped": (17, True)}.items():
        buf = ExperienceReplay(rows, d.A, 5, False, d.O, "cpu")
        for k, v in rep.items():
            getattr(buf, k)[:] = v
        buf.idx, buf.full = idx, full
        np.random.seed(11)
        idxs = np.asarray([buf._sample_idx(7) for _ in range(6)])
        vec = idxs.transpose().reshape(-1)
        np.testing.assert_array_equal(buf.observations[vec].reshape(7, 6, -1), g[f"{case}.observations"])
        np.testing.assert_array_equal(buf.actions[vec].reshape(7, 6, -1), g[f"{case}.actions"])
        np.testing.assert_array_equal(buf.rewards[vec].reshape(7, 6), g[f"{case}.rewards"])
        np.testing.assert_array_equal(buf.nonterminals[vec].reshape(7, 6, 1), g[f"{case}.nonterminals"])
        assert not any(buf.idx in row[1:] for row in idxs)


def test_config_overrides_like_hydra():
    """key=value / group.key=value overrides on the reference's config keys (src/conf/config.yaml)."""
    from big_dreamer_amd.config import load_config
    cfg = load_config(["batch_size=7", "ActorCritic.entropy_weight=1e-4", "env=HumanoidStandup-v2", "models=''"])
    assert cfg["batch_size"] == 7 and cfg["ActorCritic"]["entropy_weight"] == 1e-4
    assert cfg["env"] == "HumanoidStandup-v2" and cfg["models"] in ("", None)
    assert cfg["model_learning_rate"] == 2e-4 and cfg["ActorCritic"]["actor_learning_rate"] == 4e-5
    assert cfg["planning_horizon"] == 15 and cfg["kl_balance"] == 0.8 and cfg["free_nats"] == 3.0
    with pytest.raises(KeyError):
        load_config(["not_a_key=1"])


def test_wgrad_plan_handles_conv_descriptors_on_the_host():
    """bd_wgrad_plan is host-only: a pass that mixes 2 450-row GEMMs with gathered conv weight gradients of up to
    2.4 M rows gets per-GEMM row splits (multiples of the 16-row stage, or whole images for the 3-channel-image layers; every
    GEMM covered), and inconsistent gather geometry is rejected with a message."""
    import ctypes as C
    from big_dreamer_amd import _cabi as cabi
    from big_dreamer_amd import conv

    def desc(M, N, K, bias, g=None):
        d = cabi.WgradDesc()
        d.dpre, d.ldp, d.act1, d.lda1, d.M1, d.act2, d.lda2 = 16, N, 16, K, M, None, 0
        d.M, d.N, d.K, d.dW, d.ldw, d.db = M, N, K, 16, K, (16 if bias else None)
        if g:
            d.g_nseg, d.g_seglen, d.g_gh, d.g_gw, d.g_IH, d.g_IW, d.g_C = g
        return d

    imgs = 2450
    descs = [desc(imgs, 600, 200, True), desc(imgs, 200, 1024, False), desc(imgs, 1, 200, True),
             desc(imgs * 961, 32, 48, True, (4, 12, 31, 31, 64, 64, 3)),            # Conv2d 3->32 k4 on 64x64
             desc(imgs * 169, 64, 1152, False, (6, 192, 13, 13, 30, 30, 32)),       # ConvT 64->32 k6, 13 -> 30
             desc(imgs * 4, 256, 2048, True, (4, 512, 2, 2, 6, 6, 128))]
    arr = (cabi.WgradDesc * len(descs))(*descs)
    tb, tr, wsf = C.c_int(0), C.c_int(0), C.c_size_t(0)
    assert cabi.lib.bd_wgrad_plan(arr, len(descs), C.byref(tb), C.byref(tr), C.byref(wsf)) == 0, cabi.lib.bd_last_error()
    assert tb.value == sum(d.tiles_n * d.tiles_k * d.splits for d in arr) and tr.value > 0 and wsf.value > 0
    for d in arr:
        assert d.splits * d.rows_per >= d.M and (d.splits - 1) * d.rows_per < d.M
        if d.g_pad == 1:        # a 3-channel-image layer on the wave-private bodies (wgrad.hip): whole images per workgroup
            assert d.g_C == 3 and d.rows_per % (d.g_gh * d.g_gw) == 0 and d.tiles_n == 1 and d.tiles_k == 1
        else:
            assert d.rows_per % 16 == 0
        # tiles are at most 13 blocks wide; narrow ones (<= 2 dpre blocks) may be 36 K blocks tall (wgrad.hip, deep tiles)
        assert d.tiles_n * 13 * 16 >= d.N and d.tiles_k * 36 * 16 >= d.K
    assert arr[3].g_pad == 1 and arr[4].g_pad == 0, "the thin-image body is for the C = 3 layers only"
    assert max(d.rows_per for d in arr) < 20000, "conv GEMMs must not leave millions of rows to one workgroup"
    bad = (cabi.WgradDesc * 1)(desc(imgs * 961, 32, 48, True, (4, 12, 31, 31, 60, 64, 3)))      # windows leave the image
    assert cabi.lib.bd_wgrad_plan(bad, 1, C.byref(tb), C.byref(tr), C.byref(wsf)) != 0
    assert b"gather geometry" in cabi.lib.bd_last_error()
    # geometry helpers of the conv host module
    assert [conv.conv_out(s, 4) for s in (64, 31, 14, 6)] == [31, 14, 6, 2]
    assert [conv.convT_out(s, k) for s, k in ((1, 5), (5, 5), (13, 6), (30, 6))] == [5, 13, 30, 64]
    assert (conv.taps(5, 0), conv.taps(5, 1), conv.taps(6, 0), conv.taps(6, 1), conv.taps(4, 1)) == (3, 2, 3, 3, 2)
