"""Shared helpers for the parity tests (golden loading, fingerprint checks, comparisons)."""
import os

import numpy as np

from big_dreamer_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# name -> (Dims, seed, hyper-parameter overrides, stored in full?)
CASES = {
    "tiny": (synth.TINY, 0, {}, True),
    "small": (synth.SMALL, 1, {}, True),
    "tiny_klsum": (synth.TINY, 2, dict(kl_balance=-1, free_nats=0.05), True),
    "tiny_freenats0": (synth.TINY, 3, dict(free_nats=0.0), True),
    "tiny_pixel": (synth.TINY_PIXEL, 4, {}, False),
    "tiny_pixel_lin": (synth.TINY_PIXEL_LIN, 5, {}, False),
    "config1": (synth.CONFIG1, 0, {}, False),
    "config2": (synth.CONFIG2, 0, {}, False),
    "config3": (synth.CONFIG3, 6, {}, False),
    "tiny_discount": (synth.TINY_DISCOUNT, 9, {}, True),   # use_discount=True: discount head, Bernoulli loss, discounted actor objective       # BASELINE configs[2] at full size: 64x64 pixels, A=17, batch 50 x chunk 50
}


# latent_distribution="Categorical" (configs[4] latents): name -> (Dims, seed, overrides, stored in full?)
# golden = the reference's own Dreamer code under the two shims of oracle/gen_golden.py (CategoricalShims)
CAT_CASES = {
    "cat_tiny": (synth.CAT_TINY, 51, dict(free_nats=0.0), True),
    "cat_tiny_klsum": (synth.CAT_TINY, 52, dict(kl_balance=-1, free_nats=0.01), True),
    "cat_32": (synth.CAT_32, 53, dict(free_nats=0.0), False),
    "cat_32_v2": (synth.CAT_32, 54, dict(), False),
    # BASELINE configs[4] as stated: pixel observations + Categorical latents
    "cat_pixel_tiny": (synth.CAT_PIXEL_TINY, 55, dict(free_nats=0.0), False),
    "cat_pixel_32": (synth.CAT_PIXEL_32, 56, dict(), False),
}


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, f"{name}.npz"), allow_pickle=False))


def abssum(arrs):
    return sum(float(np.abs(np.asarray(v, dtype=np.float64)).sum()) for v in arrs)


def check_fingerprints(g, P, batch, noise):
    """The synthetic inputs are re-derived from seeds; make sure they are the ones the golden run saw."""
    assert np.isclose(abssum(v for sd in P.values() for v in sd.values()), float(g["fingerprint.params"]), rtol=1e-12)
    assert np.isclose(abssum(batch.values()), float(g["fingerprint.batch"]), rtol=1e-12)
    assert np.isclose(abssum(noise.values()), float(g["fingerprint.noise"]), rtol=1e-12)


def sample_of(a):
    a = np.asarray(a)
    return a.reshape(-1)[:: max(1, a.size // 257)][:257]


def assert_close(name, got, want, atol, rtol):
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, f"{name}: shape {got.shape} vs {want.shape}"
    err = np.abs(got - want)
    tol = atol + rtol * np.abs(want)
    bad = err > tol
    assert not bad.any(), (f"{name}: {bad.sum()}/{bad.size} elements out of tolerance; max abs err "
                           f"{err.max():.3e} (atol {atol}, rtol {rtol}); worst want={want.flat[err.argmax()]:.6g} "
                           f"got={got.flat[err.argmax()]:.6g}")


def compare_tensor(g, key, got, full, atol, rtol):
    """Compare against a golden tensor stored in full, or as (sum, abssum, strided sample)."""
    got = np.asarray(got)
    if key in g:
        assert_close(key, got.reshape(g[key].shape), g[key], atol, rtol)
    else:
        assert_close(key + ".sample", sample_of(got), g[key + ".sample"], atol, rtol)
        n = got.size
        assert_close(key + ".abssum", np.abs(got.astype(np.float64)).sum(), g[key + ".abssum"],
                     atol * n, rtol)
        # plain sums cancel, so bound their error by the abssum scale
        assert abs(got.astype(np.float64).sum() - float(g[key + ".sum"])) <= atol * n + rtol * float(g[key + ".abssum"])


# MPCPlanner golden cases: name -> (Dims, B, horizon, iters, candidates, top, seed, stored in full?)
PLANNER_CASES = {
    "planner_tiny": (synth.TINY, 2, 5, 4, 64, 8, 6, True),
    "planner_config2": (synth.CONFIG2, 1, 15, 10, 1000, 100, 7, False),
}
