// rng.hip -- bd_rng_fill: the noise tensors of one phase of a training step (reference draw sites: src/models.py:72 randn_like,
// src/models.py:114-115 OneHotCategorical sample, src/dreamer.py:443-444 rsample) filled by ONE launch of a Philox4x32-10
// generator (bd_rng.h), in place of one library RNG launch per tensor.  Perf mode only: parity tests pass explicit arrays.
#include "bd_device.h"
#include "bd_host.h"
#include "bd_rng.h"

namespace bd {

__global__ __launch_bounds__(256) void rng_fill_kernel(bd_rng_fill_args a) {
    // block -> (tensor, first group of four): tensors laid end to end in units of 1024 floats (256 threads x 4)
    size_t blk = blockIdx.x;
    int t = 0;
#pragma unroll 1
    for (; t < a.n - 1; ++t) {
        const size_t nb = (a.t[t].count + 1023) / 1024;
        if (blk < nb) break;
        blk -= nb;
    }
    const size_t i4 = blk * 256 + threadIdx.x;          // group of four floats inside tensor t
    const size_t i = i4 * 4;
    if (i >= a.t[t].count) return;
    const Rng g{(uint32_t)a.seed, (uint32_t)(a.seed >> 32), a.t[t].stream_id, (uint32_t)a.step};
    float v[4];
    if (a.t[t].kind == BD_RNG_EXPONENTIAL) rng_exp4(g, i4, v);
    else rng_normal4(g, i4, v);
    float* p = a.t[t].p + i;
    if (i + 4 <= a.t[t].count && ((uintptr_t)p & 15) == 0) {
        *reinterpret_cast<floatx4*>(p) = floatx4{v[0], v[1], v[2], v[3]};
    } else {
        for (int j = 0; j < 4 && i + j < a.t[t].count; ++j) p[j] = v[j];
    }
}

}  // namespace bd

extern "C" {
using namespace bd;

// the generator's core on the HOST (same source as the device code): known-answer tests run without a GPU
int bd_philox4x32_10(const unsigned* ctr4, const unsigned* key2, unsigned* out4) {
    BD_REQUIRE(ctr4 && key2 && out4, "bd_philox4x32_10: null pointer");
    const Philox4 p = philox4x32_10(ctr4[0], ctr4[1], ctr4[2], ctr4[3], key2[0], key2[1]);
    for (int i = 0; i < 4; ++i) out4[i] = p.x[i];
    return 0;
}

int bd_rng_fill(const bd_rng_fill_args* a, void* stream) {
    BD_REQUIRE(a && a->n > 0 && a->n <= BD_RNG_MAX_TENSORS, "bd_rng_fill: 1..%d tensors", BD_RNG_MAX_TENSORS);
    size_t blocks = 0;
    for (int t = 0; t < a->n; ++t) {
        BD_REQUIRE(a->t[t].p && a->t[t].count > 0 && (a->t[t].kind == BD_RNG_NORMAL || a->t[t].kind == BD_RNG_EXPONENTIAL),
                   "bd_rng_fill: tensor %d: bad descriptor", t);
        blocks += (a->t[t].count + 1023) / 1024;
    }
    BD_REQUIRE(blocks < (1ull << 31), "bd_rng_fill: too many elements");
    hipLaunchKernelGGL(rng_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *a);
    BD_CHECK_LAUNCH("bd_rng_fill");
    return 0;
}

}  // extern "C"
