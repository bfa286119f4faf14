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
