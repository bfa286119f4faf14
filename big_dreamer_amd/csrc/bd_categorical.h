// bd_categorical.h -- the CategoricalBeliefModel head (src/models.py:101-117) inside the persistent scan kernels of
// scan_cat.hip: hidden -> D*C logits -> per-factor softmax, sample, and the straight-through backward.
//
// The S = D*C logits of a 16-row tile do not fit LDS next to the recurrence's carries (64 KiB at 32 x 32), so the head
// runs in CHUNKS of CW <= 256 columns (whole factors): MFMA the chunk's column blocks into an LDS image, then one
// thread per (row, factor) walks the C classes.  The image is swizzled so that those walks are bank-conflict free:
//     addr(row, fl, c) = row * (CW + 8) + fl * C + (c + fl) % C        (fl = factor inside the chunk)
// -- consecutive lanes hold consecutive factors of a row, the rotation by fl spreads them over the banks, the 8-float
// row pad separates the rows of a 32-lane group.  Streamed per-class operands (the sampler's Exp(1) draws, upstream
// gradients) are staged through an image of the same shape with coalesced loads.
#pragma once
#include "bd_device.h"

namespace bd {

struct CatGeo {
    int D, C, S, CW, NCH, nF, ld;
    __host__ __device__ CatGeo(int D_, int C_) : D(D_), C(C_), S(D_ * C_) {
        if (S <= 256) {
            CW = cdiv(S, 16) * 16;
            NCH = 1;
            nF = D;
        } else {
            CW = 256;
            NCH = cdiv(S, 256);
            nF = 256 / C;
        }
        ld = CW + 8;
    }
    __host__ bool ok() const { return C >= 1 && C <= 256 && D >= 1 && (S <= 256 || (256 % C == 0 && S % 16 == 0)); }
    __host__ __device__ int image_floats() const { return 16 * ld; }
    // columns of chunk ch that exist
    __device__ __forceinline__ int cols(int ch) const { return (S - ch * CW) < CW ? (S - ch * CW) : CW; }
    __device__ __forceinline__ int addr(int row, int fl, int c) const {
        int r = c + fl % C;
        if (r >= C) r -= C;
        return row * ld + fl * C + r;
    }
    // chunk-relative column -> image address
    __device__ __forceinline__ int addr_col(int row, int colc) const {
        const int fl = colc / C;
        return addr(row, fl, colc - fl * C);
    }
};

// Coalesced copy of the chunk's columns of 16 global rows (row stride ld_g) into a swizzled image; rows >= rows_valid
// and columns >= S read `fill`.
__device__ __forceinline__ void cat_stage(const CatGeo& g, int ch, const float* __restrict__ src, size_t ld_g, int rows_valid,
                                          float fill, float* __restrict__ img) {
    const int n = g.cols(ch);
    for (int i = bd_tid(); i < 16 * g.CW; i += blockDim.x) {
        const int row = i / g.CW, colc = i - row * g.CW;
        if (colc >= n) continue;
        img[g.addr_col(row, colc)] = row < rows_valid ? src[(size_t)row * ld_g + ch * g.CW + colc] : fill;
    }
}

// One (row, factor): normalised log-probabilities as Categorical(logits=...) keeps them, probs = softmax of those,
// sample = argmax(probs / q) with the first maximum winning (torch.multinomial's single-draw path; same operation order
// as cat_head_fwd_kernel in reduce.hip).  lg / qs: swizzled images.  Returns the class.
__device__ __forceinline__ int cat_sample(const CatGeo& g, const float* __restrict__ lg, const float* __restrict__ qs, int row,
                                          int fl) {
    float m = -INFINITY;
    for (int c = 0; c < g.C; ++c) m = fmaxf(m, lg[g.addr(row, fl, c)]);
    float s = 0.f;
    for (int c = 0; c < g.C; ++c) s += expf(lg[g.addr(row, fl, c)] - m);
    const float lse = m + logf(s);
    const float m2 = m - lse;                   // max of the normalised log-probabilities
    float s2 = 0.f;
    for (int c = 0; c < g.C; ++c) s2 += expf((lg[g.addr(row, fl, c)] - lse) - m2);
    float best = -INFINITY;
    int arg = 0;
    for (int c = 0; c < g.C; ++c) {
        const int a = g.addr(row, fl, c);
        const float pr = expf((lg[a] - lse) - m2) / s2;
        const float r = pr / qs[a];
        if (r > best) { best = r; arg = c; }
    }
    return arg;
}

// Straight-through backward of one (row, factor), in place: gimg holds g = d loss / d state on entry and
// d logits = probs * (g - sum_c probs * g) on exit (probs from the logits image lg).
__device__ __forceinline__ void cat_jacobian(const CatGeo& g, const float* __restrict__ lg, float* __restrict__ gimg, int row,
                                             int fl) {
    float m = -INFINITY;
    for (int c = 0; c < g.C; ++c) m = fmaxf(m, lg[g.addr(row, fl, c)]);
    float s = 0.f;
    for (int c = 0; c < g.C; ++c) s += expf(lg[g.addr(row, fl, c)] - m);
    const float lse = m + logf(s);
    const float m2 = m - lse;
    float s2 = 0.f;
    for (int c = 0; c < g.C; ++c) s2 += expf((lg[g.addr(row, fl, c)] - lse) - m2);
    const float inv = 1.f / s2;
    float dot = 0.f;
    for (int c = 0; c < g.C; ++c) {
        const int a = g.addr(row, fl, c);
        dot += expf((lg[a] - lse) - m2) * inv * gimg[a];
    }
    for (int c = 0; c < g.C; ++c) {
        const int a = g.addr(row, fl, c);
        const float pr = expf((lg[a] - lse) - m2) * inv;
        gimg[a] = pr * (gimg[a] - dot);
    }
}

// The head, forward: logits = hid W2^T + b2 chunk by chunk; `logit_row(row)` gives the global row base of the logits
// output (or nullptr), `q_row0` the first of 16 consecutive global rows of the sampler's draws (row stride S).
// Fills sidx_l[16][D] (int) with the sampled classes.  Ends with a workgroup barrier.
__device__ __forceinline__ void cat_head_forward(const CatGeo& g, const float* __restrict__ hid, int Kb_hd,
                                                 const float* __restrict__ w2, const float* __restrict__ b2,
                                                 const float* __restrict__ q_row0, float* __restrict__ logits_row0,
                                                 int rows_valid, float* __restrict__ lg, float* __restrict__ qs,
                                                 int* __restrict__ sidx_l) {
    const int lane = bd_tid() & 63;
    for (int ch = 0; ch < g.NCH; ++ch) {
        const int n = g.cols(ch);
        cat_stage(g, ch, q_row0, (size_t)g.S, rows_valid, 1.f, qs);
        const Seg seg[1] = {{hid, w2 + (size_t)ch * (g.CW / 16) * Kb_hd * kFragFloats, Kb_hd}};
        tile_linear_g<1, 1>(seg, b2 + ch * g.CW, n, [&](int, int nb, floatx4 acc) {
            const int colc = nb * 16 + (lane & 15);
            if (colc >= n) return;
            const int fl = colc / g.C, c = colc - fl * g.C;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * (lane >> 4) + r;
                lg[g.addr(row, fl, c)] = acc[r];
                if (logits_row0 && row < rows_valid) logits_row0[(size_t)row * g.S + ch * g.CW + colc] = acc[r];
            }
        });
        lds_barrier();
        const int nf = n / g.C;
        for (int i = bd_tid(); i < 16 * nf; i += blockDim.x) {
            const int row = i / nf, fl = i - row * nf;
            sidx_l[row * g.D + ch * g.nF + fl] = cat_sample(g, lg, qs, row, fl);
        }
        lds_barrier();
    }
}

}  // namespace bd
