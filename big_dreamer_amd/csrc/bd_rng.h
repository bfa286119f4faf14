// bd_rng.h -- counter-based random numbers for the perf mode of the training step (north_star: "Philox only in perf mode";
// parity tests always pass explicit noise arrays).
//
// Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11): a keyed bijection of a
// 128-bit counter; no state, any element of any stream is computable on its own, so a kernel can (re)generate the draws it
// consumes instead of reading them from HBM.  Layout used here:
//     key     = (seed lo, seed hi)                       -- per process: torch.initial_seed()
//     counter = (index lo, index hi, stream id, step)    -- index: which group of 4 values; stream id: which noise tensor
//                                                            (observe / action / prior / entropy ...); step: train step
// One call yields four 32-bit words = four uniforms = four normals (two Box-Muller pairs) or four Exp(1) variates.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bd {

struct Philox4 {
    uint32_t x[4];
};

__host__ __device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                                          uint32_t k1) {
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    return Philox4{{c0, c1, c2, c3}};
}

// uniform in (0, 1): 24 random bits, never 0 or 1 (so that log(u) is finite)
__host__ __device__ __forceinline__ float u01(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }

struct Rng {
    uint32_t k0, k1, stream, step;
};

// four standard normals: two Box-Muller pairs (accurate logf / sincosf: the tails matter for the tanh-Normal entropy estimate)
__device__ __forceinline__ void rng_normal4(const Rng& g, uint64_t idx, float (&out)[4]) {
    const Philox4 p = philox4x32_10((uint32_t)idx, (uint32_t)(idx >> 32), g.stream, g.step, g.k0, g.k1);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float r = sqrtf(-2.f * logf(u01(p.x[2 * h])));
        float s, c;
        sincosf(6.28318530717958647692f * u01(p.x[2 * h + 1]), &s, &c);
        out[2 * h] = r * c;
        out[2 * h + 1] = r * s;
    }
}

// four Exp(1) variates (torch.multinomial's single-draw path consumes one per class)
__device__ __forceinline__ void rng_exp4(const Rng& g, uint64_t idx, float (&out)[4]) {
    const Philox4 p = philox4x32_10((uint32_t)idx, (uint32_t)(idx >> 32), g.stream, g.step, g.k0, g.k1);
#pragma unroll
    for (int h = 0; h < 4; ++h) out[h] = -logf(u01(p.x[h]));
}

}  // namespace bd
