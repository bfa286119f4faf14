"""CPU: the perf-mode generator's core (csrc/bd_rng.h) is Philox4x32-10: the published known-answer vectors of the Random123
distribution (kat_vectors, philox4x32 10 rounds) through the host build of the same source, and through the Python restatement
the GPU tests use to check the device output."""
import ctypes as C

from tests.test_rng_gpu import _philox_ref

KAT = [([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
       ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
       ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0], [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1])]


def test_python_restatement_matches_known_answers():
    for ctr, key, want in KAT:
        assert _philox_ref(ctr, key) == want


def test_library_core_matches_known_answers():
    from big_dreamer_amd import _cabi as cabi
    for ctr, key, want in KAT:
        out = (C.c_uint * 4)()
        cabi.check(cabi.lib.bd_philox4x32_10((C.c_uint * 4)(*ctr), (C.c_uint * 2)(*key), out))
        assert list(out) == want
