"""GPU: the perf-mode noise generator (csrc/bd_rng.h, csrc/rng.hip: Philox4x32-10, counter-based).  Parity tests never use
it (they pass the reference's draws as explicit arrays); what is checked here is that it IS Philox4x32-10 (known-answer
vectors of the Random123 distribution), that its normals / exponentials have the right distribution, that streams and steps
are independent and reproducible, and that the in-kernel entropy estimator agrees with the explicit-noise estimator in
distribution."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _fill(tensors, seed, step):
    from big_dreamer_amd import _cabi as cabi
    a = cabi.RngFillArgs()
    a.n, a.seed, a.step = len(tensors), seed, step
    for i, (t, kind, sid) in enumerate(tensors):
        a.t[i] = cabi.RngTensor(t.data_ptr(), t.numel(), kind, sid)
    cabi.check(cabi.lib.bd_rng_fill(C.byref(a), cabi.stream()))
    torch.cuda.synchronize()


def _philox_ref(ctr, key):
    """Philox4x32-10 in Python integers (Salmon et al., SC'11)."""
    M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
    c, k = list(ctr), list(key)
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        c = [((p1 >> 32) ^ c[1] ^ k[0]) & 0xFFFFFFFF, p1 & 0xFFFFFFFF, ((p0 >> 32) ^ c[3] ^ k[1]) & 0xFFFFFFFF, p0 & 0xFFFFFFFF]
        k = [(k[0] + W0) & 0xFFFFFFFF, (k[1] + W1) & 0xFFFFFFFF]
    return c


def test_python_philox_matches_the_published_known_answers():
    # Random123 kat_vectors: philox4x32 10 rounds
    assert _philox_ref([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert _philox_ref([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert _philox_ref([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_device_generator_is_philox4x32_10():
    """Exp(1) variates are -log(u) with u = ((x >> 8) + 0.5) / 2^24 of the four Philox words: invert and compare the 24 used
    bits of every word with the Python implementation pinned by the published vectors above."""
    from big_dreamer_amd import _cabi as cabi
    seed, step, sid = 0x0123456789ABCDEF, 7, 3
    t = torch.zeros(4096, device="cuda")
    _fill([(t, cabi.BD_RNG_EXPONENTIAL, sid)], seed, step)
    got = np.rint(np.exp(-t.cpu().numpy().astype(np.float64)) * 16777216.0 - 0.5).astype(np.int64)
    for i4 in (0, 1, 17, 1023):
        want = [w >> 8 for w in _philox_ref([i4, 0, sid, step], [seed & 0xFFFFFFFF, seed >> 32])]
        assert np.abs(got[4 * i4:4 * i4 + 4] - np.array(want)).max() <= 2, (i4, got[4 * i4:4 * i4 + 4], want)     # float32 log/exp round trip


def test_distributions_streams_and_reproducibility():
    from scipy import stats
    from big_dreamer_amd import _cabi as cabi
    n = 1 << 21
    a, b, c, e = (torch.zeros(n, device="cuda") for _ in range(4))
    _fill([(a, cabi.BD_RNG_NORMAL, 1), (b, cabi.BD_RNG_NORMAL, 2), (e, cabi.BD_RNG_EXPONENTIAL, 3)], 1234, 0)
    _fill([(c, cabi.BD_RNG_NORMAL, 1)], 1234, 1)
    x, y, z, w = (t.cpu().numpy().astype(np.float64) for t in (a, b, c, e))
    for v in (x, y, z):
        assert abs(v.mean()) < 4.0 / np.sqrt(n) and abs(v.var() - 1.0) < 6.0 * np.sqrt(2.0 / n)
        assert abs(stats.skew(v)) < 0.01 and abs(stats.kurtosis(v)) < 0.02
        assert stats.kstest(v[:200000], "norm").pvalue > 1e-4
        assert v.max() > 4.5 and v.min() < -4.5                     # tails present (24-bit uniforms reach |z| ~ 5.7)
    assert abs(w.mean() - 1.0) < 5.0 / np.sqrt(n) and abs(w.var() - 1.0) < 0.02 and w.min() > 0
    assert stats.kstest(w[:200000], "expon").pvalue > 1e-4
    # different streams / different steps: uncorrelated; same (seed, stream, step): bit-identical
    for u, v in ((x, y), (x, z)):
        assert abs(np.corrcoef(u, v)[0, 1]) < 5.0 / np.sqrt(n)
    # consecutive values (Box-Muller pairs, neighbouring counters): uncorrelated
    assert abs(np.corrcoef(x[:-1], x[1:])[0, 1]) < 5.0 / np.sqrt(n)
    a2 = torch.zeros(n, device="cuda")
    _fill([(a2, cabi.BD_RNG_NORMAL, 1)], 1234, 0)
    assert torch.equal(a, a2)
    # ragged count / unaligned base: every element written, none beyond
    r = torch.full((1031,), -7.0, device="cuda")
    _fill([(r[1:1030], cabi.BD_RNG_NORMAL, 9)], 5, 5)
    assert float(r[0]) == -7.0 and float(r[1030]) == -7.0 and bool((r[1:1030] != -7.0).all())


def test_in_kernel_entropy_draws_match_the_explicit_noise_estimator_in_distribution():
    """bd_actor_entropy_rng (100 draws per (row, action dim) generated in the kernel) against bd_actor_entropy on torch
    normals: two independent 100-sample Monte-Carlo estimates of the same tanh-Normal entropies -- their means over 20 000
    rows agree to the estimator's own standard error, and so do the saved d/d mean, d/d std."""
    from big_dreamer_amd import _cabi as cabi
    Hm, N, A, ns = 5, 4000, 3, 100
    g = torch.Generator(device="cuda").manual_seed(3)
    mean = torch.randn(Hm * N, A, device="cuda", generator=g)
    std = torch.rand(Hm * N, A, device="cuda", generator=g) * 1.5 + 0.2
    out = []
    for mode in ("explicit", "rng", "rng2"):
        stats_ = torch.zeros(Hm * N, 4 * A, device="cuda")
        stats_[:, 2 * A:3 * A], stats_[:, 3 * A:] = mean, std
        ent = torch.zeros(Hm * N, device="cuda")
        if mode == "explicit":
            eps = torch.randn(Hm, ns, N, A, device="cuda", generator=g)
            cabi.check(cabi.lib.bd_actor_entropy(eps.data_ptr(), stats_.data_ptr(), ent.data_ptr(), Hm, N, A, ns, cabi.stream()))
        else:
            cabi.check(cabi.lib.bd_actor_entropy_rng(99, 4 if mode == "rng" else 5, 4, stats_.data_ptr(), ent.data_ptr(), Hm, N, A,
                                                     ns, cabi.stream()))
        torch.cuda.synchronize()
        out.append((ent.cpu().double().numpy(), stats_[:, 2 * A:].cpu().double().numpy()))
    (e0, s0), (e1, s1), (e2, s2) = out
    se = np.sqrt((e1 - e2).var() / 2.0 / len(e0))            # standard error of a mean of such estimates
    assert abs(e0.mean() - e1.mean()) < 6 * se * np.sqrt(2) and abs(e1.mean() - e2.mean()) < 6 * se * np.sqrt(2)
    assert not np.array_equal(e1, e2) and np.isfinite(e1).all() and np.isfinite(s1).all()
    # per-row: the two estimators scatter around each other like two independent draws of the same estimator
    assert abs(np.std(e0 - e1) / np.std(e1 - e2) - 1.0) < 0.1
    assert abs(s0.mean(0) - s1.mean(0)).max() < 6 * np.sqrt((s1 - s2).var(0).max() / len(e0))


def test_pixel_gather_with_in_kernel_noise_is_the_reference_dequantisation():
    """bd_replay_gather_pixels_rng: floor(u8 / 2^(8-bits)) / 2^bits - 0.5 + U[0,1) / 2^bits (preprocess_observation_,
    src/utils.py:299-317) with the uniform drawn in the kernel: the quantised part is exact, the noise part lies in [0, 1),
    is uniform, differs between steps and is reproducible."""
    from scipy import stats
    from big_dreamer_amd import _cabi as cabi
    rows, n_idx, pixels, bits = 9, 40, 3 * 64 * 64, 5
    g = torch.Generator().manual_seed(1)
    u8 = torch.randint(0, 256, (rows, pixels), dtype=torch.uint8, generator=g).cuda()
    idx = torch.randint(0, rows, (n_idx,), dtype=torch.int64, generator=g).cuda()
    outs = []
    for step in (0, 0, 1):
        dst = torch.zeros(n_idx, pixels, device="cuda")
        cabi.check(cabi.lib.bd_replay_gather_pixels_rng(u8.data_ptr(), idx.data_ptr(), n_idx, pixels, bits, 77, step,
                                                        dst.data_ptr(), cabi.stream()))
        torch.cuda.synchronize()
        outs.append(dst)
    assert torch.equal(outs[0], outs[1]) and not torch.equal(outs[0], outs[2])
    q = torch.floor(u8[idx].float() / 2 ** (8 - bits))
    noise = ((outs[0] + 0.5) * 2 ** bits - q).cpu().double().numpy().reshape(-1)
    assert noise.min() >= -1e-5 and noise.max() < 1.0 + 1e-5
    assert abs(noise.mean() - 0.5) < 1e-3 and abs(noise.var() - 1.0 / 12.0) < 1e-3
    assert stats.kstest(np.clip(noise[:200000], 0.0, 1.0), "uniform").pvalue > 1e-4
