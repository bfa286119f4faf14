// bd_scan.h -- device helpers shared by the imagination scans (imagine.hip: Gaussian latents; scan_cat.hip: Categorical
// latents): the tanh-Normal entropy sample, hidden-layer / pre-activation-gradient epilogues.
#pragma once
#include "bd_device.h"

namespace bd {

// One sample of the entropy estimate: log-density of y = tanh(mean + std*e) under the tanh-Normal, and its
// derivatives w.r.t. mean and std following the reference's autograd graph (rsample -> tanh -> clamp ->
// atanh -> Normal.log_prob - log|det J|).  Terms that depend only on (row, action dim) -- log std, 1/std^2,
// 1/std -- are hoisted by the caller; exp(-2x) is shared by softplus(-2x) and sigmoid(-2x).
struct EntConst {
    float mean, sd, inv_var, inv_sd, base0;     // base0 = -log(sd) - log(sqrt(2 pi)) - 2 ln 2
};
__device__ __forceinline__ EntConst entropy_const(float mean, float sd) {
    constexpr float kLogSqrt2Pi = 0.91893853320467274f, kLn2 = 0.69314718055994531f;
    return EntConst{mean, sd, 1.f / (sd * sd), 1.f / sd, -logf(sd) - kLogSqrt2Pi - 2.f * kLn2};
}
__device__ __forceinline__ void entropy_sample(const EntConst& c, float e, float& lp, float& dm, float& ds) {
    constexpr float kClamp = 0.99999994f;                // float32(0.99999997), src/models.py:663
    const float u = c.mean + c.sd * e;
    const float y = tanhf(u);
    const float yc = fminf(fmaxf(y, -kClamp), kClamp);
    const float a = 1.f + yc, b = 1.f - yc;
    const float xh = 0.5f * logf(a / b);                 // atanh (src/models.py:627)
    const float diff = xh - c.mean;
    const float t = -2.f * xh;
    const float ex = expf(t);                            // shared: softplus(t) = log1p(ex), sigmoid(t) = ex / (1 + ex)
    const float sp = t > 20.f ? t : log1pf(ex);          // F.softplus threshold
    const float sg = t > 20.f ? 1.f : ex / (1.f + ex);
    // log p = -(xh-mean)^2/(2 sd^2) - log sd - log sqrt(2pi) - 2 (ln 2 - xh - softplus(-2 xh))     (src/models.py:673)
    lp = c.base0 - 0.5f * diff * diff * c.inv_var + 2.f * (xh + sp);
    const float gx = -diff * c.inv_var + 2.f - 4.f * sg;                    // d lp / d xh
    const bool pass = (y >= -kClamp) && (y <= kClamp);                      // clamp backward mask
    const float J = pass ? (1.f - y * y) / (a * b) : 0.f;                   // d xh / d u = (1 - y^2) / ((1+yc)(1-yc))
    dm = diff * c.inv_var + gx * J;
    ds = diff * diff * c.inv_var * c.inv_sd - c.inv_sd + gx * J * e;
}

// The three epilogue helpers below take a branch-free path for a block that lies wholly inside the matrix (uniform
// test: every row of the tile valid, all 16 columns < width) with 32-bit lane offsets off a uniform base pointer; the
// per-element predicates and 64-bit index products of the general path cost ~250 cycles per value (s_memtime stamps of
// the dense-chain kernels, tools/tall_stamps.py).  rows * width of one time slice must stay below 2^31 (host-checked).

// hidden layer epilogue: ELU -> LDS fragment (+ optional save for the backward)
struct HiddenEpi {
    float* dst;
    float* save;
    size_t tn;
    int width, rows, row0, lane;
    __device__ __forceinline__ void operator()(int nb, floatx4 acc) const {
        const int c = lane & 15, q = lane >> 4;
        float* __restrict__ p = dst + acc_frag_off(nb, lane, 0);
        if (row0 + 16 <= rows && nb * 16 + 16 <= width) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = elu(acc[r]);
#pragma unroll
            for (int r = 0; r < 4; ++r) p[4 * r] = v[r];
            if (save) {
                float* __restrict__ s = save + tn * width + ((unsigned)(row0 + 4 * q) * (unsigned)width + (unsigned)(nb * 16 + c));
#pragma unroll
                for (int r = 0; r < 4; ++r) st_save(s + (unsigned)r * (unsigned)width, v[r]);
            }
            return;
        }
        const int col = nb * 16 + c;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int grow = row0 + 4 * q + r;
            const bool ok = grow < rows && col < width;
            const float v = ok ? elu(acc[r]) : 0.f;
            p[4 * r] = v;
            if (ok && save) save[(tn + grow) * width + col] = v;
        }
    }
};

// The same epilogue for TRANSPOSED accumulators (tile_linear_seg_tr): the lane's four values are four consecutive columns
// of ONE row -- its 16 bytes of the next layer's fragment tile and 16 contiguous bytes of the row-major save: one
// ds_write_b128 and one global_store_dwordx4 instead of four of each with four address computations.  In a scan the
// epilogue of one wave competes for issue slots with the MFMA stream of the other wave of its SIMD (DESIGN: how a SIMD
// shares its issue slot), so instruction count is what it costs.  Any width: a lane whose four columns straddle the edge
// stores them one by one.
struct HiddenEpiTR {
    float* dst;
    float* save;
    size_t tn;
    int width, rows, row0, lane;
    __device__ __forceinline__ void operator()(int nb, floatx4 acc) const {
        const int row = lane & 15, col0 = nb * 16 + 4 * (lane >> 4);
        const bool rok = row0 + row < rows;
        floatx4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (rok && col0 + r < width) ? elu(acc[r]) : 0.f;
        reinterpret_cast<floatx4*>(dst)[nb * 64 + lane] = v;
        if (save && rok && col0 < width) {
            float* __restrict__ s = save + (tn + row0 + row) * width + col0;
            if (col0 + 4 <= width) {
                __builtin_memcpy(s, &v, 16);          // (4-byte aligned address: one global_store_dwordx4)
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (col0 + r < width) s[r] = v[r];
            }
        }
    }
};

// "gradient w.r.t. a hidden ELU output" epilogue: multiply by ELU' (from the saved output), keep in LDS for the
// next contraction (dst may be null) and store for bd_wgrad (out may be null).
struct DpreEpi {
    float* dst;
    float* out;
    size_t tn;
    int width, rows, row0, lane;
    __device__ __forceinline__ void operator()(int, int nb, floatx4 acc, const Pre4& p) const {
        const int c = lane & 15, q = lane >> 4;
        if (row0 + 16 <= rows && nb * 16 + 16 <= width) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[r] * elu_grad_from_out(p.v[r]);
            if (out) {
                float* __restrict__ s = out + tn * width + ((unsigned)(row0 + 4 * q) * (unsigned)width + (unsigned)(nb * 16 + c));
#pragma unroll
                for (int r = 0; r < 4; ++r) s[(unsigned)r * (unsigned)width] = v[r];
            }
            if (dst) {
                float* __restrict__ w = dst + acc_frag_off(nb, lane, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) w[4 * r] = v[r];
            }
            return;
        }
        const int col = nb * 16 + c;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int grow = row0 + 4 * q + r;
            float v = 0.f;
            if (grow < rows && col < width) {
                v = acc[r] * elu_grad_from_out(p.v[r]);
                if (out) out[(tn + grow) * width + col] = v;
            }
            if (dst) dst[acc_frag_off(nb, lane, r)] = v;
        }
    }
};
// the saved ELU outputs that DpreEpi needs, fetched before the contraction
struct DprePre {
    const float* saved;
    size_t tn;
    int width, rows, row0, lane;
    __device__ __forceinline__ Pre4 operator()(int, int nb) const {
        Pre4 p;
        const int c = lane & 15, q = lane >> 4;
        if (row0 + 16 <= rows && nb * 16 + 16 <= width) {
            const float* __restrict__ s = saved + tn * width + ((unsigned)(row0 + 4 * q) * (unsigned)width + (unsigned)(nb * 16 + c));
#pragma unroll
            for (int r = 0; r < 4; ++r) p.v[r] = s[(unsigned)r * (unsigned)width];
            return p;
        }
        const int col = nb * 16 + c;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int grow = row0 + 4 * q + r;
            p.v[r] = (grow < rows && col < width) ? saved[(tn + grow) * width + col] : 1.f;
        }
        return p;
    }
};

struct PreAct {          // operands of the action-sample backward per accumulator row
    float act[4], th[4], sg[4], dm[4], ds[4], eps[4];
};

}  // namespace bd
