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
// and columns >= S read `fill`.  All of a thread's loads are issued before its first LDS write (16-byte loads where the
// layout allows): written as load-store pairs the loop paid one global round trip per element.
__device__ __forceinline__ void cat_stage(const CatGeo& g, int ch, const float* __restrict__ src, size_t ld_g, int rows_valid,
                                          float fill, float* __restrict__ img) {
    const int n = g.cols(ch);
    const float* __restrict__ base = src + ch * g.CW;
    if ((n & 3) == 0 && (ld_g & 3) == 0 && (g.C & 3) == 0 && (((uintptr_t)base) & 15) == 0) {
        const int n4 = n >> 2, total = 16 * n4;
        constexpr int NIT = 4;
        for (int i0 = bd_tid(); i0 < total; i0 += NIT * blockDim.x) {
            floatx4 v[NIT];
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int i = i0 + it * blockDim.x;
                v[it] = floatx4{fill, fill, fill, fill};
                if (i < total) {
                    const int row = i / n4, c4 = i - row * n4;
                    if (row < rows_valid) v[it] = *reinterpret_cast<const floatx4*>(base + (size_t)row * ld_g + 4 * c4);
                }
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int i = i0 + it * blockDim.x;
                if (i < total) {
                    const int row = i / n4, colc = 4 * (i - row * n4);
                    const int fl = colc / g.C, c = colc - fl * g.C;      // the four classes share a factor (C % 4 == 0)
#pragma unroll
                    for (int j = 0; j < 4; ++j) img[g.addr(row, fl, c + j)] = v[it][j];
                }
            }
        }
        return;
    }
    for (int i = bd_tid(); i < 16 * g.CW; i += blockDim.x) {
        const int row = i / g.CW, colc = i - row * g.CW;
        if (colc >= n) continue;
        img[g.addr_col(row, colc)] = row < rows_valid ? base[(size_t)row * ld_g + colc] : fill;
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

// The same with everything in REGISTERS (compile-time C, C % 4 == 0): logits and g from the chunk images, the direct
// logit gradient and the output as 16-byte global accesses (a thread's C values are contiguous), the result also into the
// fragment tile dLf of the chunk's K blocks.  dextra / dout: this (row, factor)'s first class, or null.
template <int CC>
__device__ __forceinline__ void cat_jacobian_reg(const CatGeo& g, const float* __restrict__ lg, const float* __restrict__ gimg,
                                                 int row, int fl, bool valid, const float* __restrict__ dextra,
                                                 float* __restrict__ dout, float* __restrict__ dLf) {
    float v[CC], gr[CC];
    floatx4 ex[CC / 4];
#pragma unroll
    for (int c = 0; c < CC; c += 4)
        ex[c / 4] = (dextra && valid) ? *reinterpret_cast<const floatx4*>(dextra + c) : floatx4{0.f, 0.f, 0.f, 0.f};
    const int rot = fl % CC;
#pragma unroll
    for (int c = 0; c < CC; ++c) {
        int r = c + rot;
        if (r >= CC) r -= CC;
        v[c] = lg[row * g.ld + fl * CC + r];
        gr[c] = gimg[row * g.ld + fl * CC + r];
    }
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < CC; ++c) m = fmaxf(m, v[c]);
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CC; ++c) s += expf(v[c] - m);
    const float lse = m + logf(s);
    const float m2 = m - lse;
    float s2 = 0.f;
#pragma unroll
    for (int c = 0; c < CC; ++c) {
        v[c] = expf((v[c] - lse) - m2);
        s2 += v[c];
    }
    const float inv = 1.f / s2;
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < CC; ++c) {
        v[c] *= inv;
        dot += v[c] * gr[c];
    }
#pragma unroll
    for (int c = 0; c < CC; c += 4) {
        floatx4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            o[j] = valid ? v[c + j] * (gr[c + j] - dot) + ex[c / 4][j] : 0.f;
            dLf[frag_idx(row, fl * CC + c + j)] = o[j];
        }
        if (dout && valid) *reinterpret_cast<floatx4*>(dout + c) = o;
    }
}

// C == 32, FOUR lanes per (row, factor), eight classes each: every thread of the workgroup works (16 rows x 8 factors of a
// 256-column chunk x 4 = 512), the three per-factor reductions (max, sum of exponentials, probs . g) are two quad
// shuffles each, the softmax needs ONE hardware exponential per class (p = e / sum e).  `quad` = lane index inside the
// quad (tid & 3); dextra / dout point at this (row, factor)'s first class.
__device__ __forceinline__ float quad_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 1, 64));
    return fmaxf(v, __shfl_xor(v, 2, 64));
}
__device__ __forceinline__ float quad_sum(float v) {
    v += __shfl_xor(v, 1, 64);
    return v + __shfl_xor(v, 2, 64);
}
__device__ __forceinline__ void cat_jacobian_quad32(const CatGeo& g, const float* __restrict__ lg, const float* __restrict__ gimg,
                                                    int row, int fl, int quad, bool valid, const float* __restrict__ dextra,
                                                    float* __restrict__ dout, float* __restrict__ dLf) {
    constexpr int CC = 32, PER = 8;
    const int c0 = quad * PER;
    floatx4 ex[2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
        ex[h] = (dextra && valid) ? *reinterpret_cast<const floatx4*>(dextra + c0 + 4 * h) : floatx4{0.f, 0.f, 0.f, 0.f};
    float v[PER], gr[PER];
    const int rot = fl % CC;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        int r = c0 + j + rot;
        if (r >= CC) r -= CC;
        v[j] = lg[row * g.ld + fl * CC + r];
        gr[j] = gimg[row * g.ld + fl * CC + r];
    }
    float m = v[0];
#pragma unroll
    for (int j = 1; j < PER; ++j) m = fmaxf(m, v[j]);
    m = quad_max(m);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        v[j] = __expf(v[j] - m);
        s += v[j];
    }
    const float inv = __builtin_amdgcn_rcpf(quad_sum(s));
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        v[j] *= inv;
        dot += v[j] * gr[j];
    }
    dot = quad_sum(dot);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        floatx4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            o[j] = valid ? v[4 * h + j] * (gr[4 * h + j] - dot) + ex[h][j] : 0.f;
            dLf[frag_idx(row, fl * CC + c0 + 4 * h + j)] = o[j];
        }
        if (dout && valid) *reinterpret_cast<floatx4*>(dout + c0 + 4 * h) = o;
    }
}

// Full-width image (forward heads: all S logits of the tile at once, row stride S + 8): same swizzle, fl = global factor.
struct CatFull {
    int D, C, S, ld;
    __host__ __device__ CatFull(int D_, int C_) : D(D_), C(C_), S(D_ * C_), ld(cdiv(D_ * C_, 16) * 16 + 8) {}
    __host__ __device__ int image_floats() const { return 16 * ld; }
    __device__ __forceinline__ int addr(int row, int f, int c) const {
        int r = c + f % C;
        if (r >= C) r -= C;
        return row * ld + f * C + r;
    }
};

// One (row, factor) with the C logits and the C draws in REGISTERS (compile-time C): same operations, same order as
// cat_sample; the draws come straight from global memory as 16-byte loads (a thread's C draws are contiguous).
template <int CC>
__device__ __forceinline__ int cat_sample_reg(const CatFull& g, const float* __restrict__ lg, const float* __restrict__ qrow,
                                              int row, int f) {
    float v[CC], q[CC];
    if constexpr (CC % 4 == 0) {
#pragma unroll
        for (int c = 0; c < CC; c += 4) {
            const floatx4 t = *reinterpret_cast<const floatx4*>(qrow + c);
            q[c] = t[0]; q[c + 1] = t[1]; q[c + 2] = t[2]; q[c + 3] = t[3];
        }
    } else {
#pragma unroll
        for (int c = 0; c < CC; ++c) q[c] = qrow[c];
    }
    int rot = f % CC;
#pragma unroll
    for (int c = 0; c < CC; ++c) {
        int r = c + rot;
        if (r >= CC) r -= CC;
        v[c] = lg[row * g.ld + f * CC + r];
    }
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < CC; ++c) m = fmaxf(m, v[c]);
#ifdef BD_EXACT_MATH
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CC; ++c) s += expf(v[c] - m);
    const float lse = m + logf(s);
    const float m2 = m - lse;
    float s2 = 0.f;
#pragma unroll
    for (int c = 0; c < CC; ++c) {
        v[c] = expf((v[c] - lse) - m2);
        s2 += v[c];
    }
    float best = -INFINITY;
    int arg = 0;
#pragma unroll
    for (int c = 0; c < CC; ++c) {
        const float r = (v[c] / s2) / q[c];
        if (r > best) { best = r; arg = c; }
    }
    return arg;
#else
    // argmax_c probs_c / q_c = argmax_c exp(v_c - m) / q_c: the two normalisations of the library's softmax are common
    // positive factors.  One hardware exponential and one reciprocal per class instead of three libm exponentials and two
    // divisions (the sample phase was a third of an observe step); the ratios differ from the library's by ~1e-7
    // relative, i.e. a sample can differ only where two classes tie to that precision.
    float best = -INFINITY;
    int arg = 0;
#pragma unroll
    for (int c = 0; c < CC; ++c) {
        const float r = __expf(v[c] - m) * __builtin_amdgcn_rcpf(q[c]);
        if (r > best) { best = r; arg = c; }
    }
    return arg;
#endif
}

// generic C: logits from the image, draws from global memory
__device__ __forceinline__ int cat_sample_any(const CatFull& g, const float* __restrict__ lg, const float* __restrict__ qrow,
                                              int row, int f) {
    float m = -INFINITY;
    for (int c = 0; c < g.C; ++c) m = fmaxf(m, lg[g.addr(row, f, c)]);
    float s = 0.f;
    for (int c = 0; c < g.C; ++c) s += expf(lg[g.addr(row, f, c)] - m);
    const float lse = m + logf(s);
    const float m2 = m - lse;
    float s2 = 0.f;
    for (int c = 0; c < g.C; ++c) s2 += expf((lg[g.addr(row, f, c)] - lse) - m2);
    float best = -INFINITY;
    int arg = 0;
    for (int c = 0; c < g.C; ++c) {
        const float r = (expf((lg[g.addr(row, f, c)] - lse) - m2) / s2) / qrow[c];
        if (r > best) { best = r; arg = c; }
    }
    return arg;
}

// The head, forward, whole width at once: ONE contraction over all S/16 column blocks into the image (8 blocks per wave
// at 32 x 32: a deep software pipeline, no per-chunk barriers), then every thread samples one (row, factor).
// q_row0: first of 16 consecutive global rows of the sampler's draws (row stride S).  Ends with a workgroup barrier.
__device__ __forceinline__ void cat_head_forward_full(const CatFull& g, const float* __restrict__ hid, int Kb_hd,
                                                      const float* __restrict__ w2, const float* __restrict__ b2,
                                                      const float* __restrict__ q_row0, float* __restrict__ logits_row0,
                                                      int rows_valid, float* __restrict__ lg, int* __restrict__ sidx_l) {
    const int lane = bd_tid() & 63;
    const Seg seg[1] = {{hid, w2, Kb_hd}};
    tile_linear_g<1, 1>(seg, b2, g.S, [&](int, int nb, floatx4 acc) {
        const int col = nb * 16 + (lane & 15);
        if (col >= g.S) return;
        const int f = col / g.C, c = col - f * g.C;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * (lane >> 4) + r;
            lg[g.addr(row, f, c)] = acc[r];
            if (logits_row0 && row < rows_valid) logits_row0[(size_t)row * g.S + col] = acc[r];
        }
    });
    lds_barrier();
    for (int i = bd_tid(); i < 16 * g.D; i += blockDim.x) {
        const int row = i / g.D, f = i - row * g.D;
        int arg = 0;
        if (row < rows_valid) {
            const float* qrow = q_row0 + (size_t)row * g.S + f * g.C;
            arg = g.C == 32 ? cat_sample_reg<32>(g, lg, qrow, row, f) : cat_sample_any(g, lg, qrow, row, f);
        }
        sidx_l[i] = arg;
    }
    lds_barrier();
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

// ---- the one-hot state as class indices (shared by scan_cat.hip and observe_cat_cluster.hip) ----------------------------
// gather-sum of the state columns of a first layer: out[row][col] = scale[row] * sum_f w[row][f] * WT[(f*C + idx[row][f]) * N + col]
__device__ __forceinline__ void state_gather(const CatGeo& g, const float* __restrict__ WT, int N, const int* __restrict__ sidx_l,
                                             const float* __restrict__ sw_l, const float* __restrict__ scale_l,
                                             float* __restrict__ out) {
    if ((N & 3) == 0) {     // 16-byte loads, eight rows of the gather in flight per thread
        const int N4 = N >> 2;
        for (int i = bd_tid(); i < 16 * N4; i += blockDim.x) {
            const int row = i / N4, c4 = i - row * N4;
            const floatx4* __restrict__ W4 = reinterpret_cast<const floatx4*>(WT) + c4;
            floatx4 s = floatx4{0.f, 0.f, 0.f, 0.f};
            int f = 0;
            for (; f + 8 <= g.D; f += 8) {
                floatx4 t[8];
                float w[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    w[j] = sw_l[row * g.D + f + j];
                    t[j] = W4[(size_t)((f + j) * g.C + sidx_l[row * g.D + f + j]) * N4];
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) s += w[j] * t[j];
            }
            for (; f < g.D; ++f) s += sw_l[row * g.D + f] * W4[(size_t)(f * g.C + sidx_l[row * g.D + f]) * N4];
            if (scale_l) s *= scale_l[row];
            reinterpret_cast<floatx4*>(out)[i] = s;
        }
        return;
    }
    for (int i = bd_tid(); i < 16 * N; i += blockDim.x) {
        const int row = i / N, col = i - row * N;
        float s = 0.f;
        for (int f = 0; f < g.D; ++f) {
            const float w = sw_l[row * g.D + f];
            const int k = f * g.C + sidx_l[row * g.D + f];
            s += w * WT[(size_t)k * N + col];
        }
        out[i] = scale_l ? s * scale_l[row] : s;
    }
}

// class indices / weights of a dense [rows x S] state that is zero or one-hot per factor
__device__ __forceinline__ void state_to_indices(const CatGeo& g, const float* __restrict__ dense, size_t ld, int row0, int rows,
                                                 int* __restrict__ sidx_l, float* __restrict__ sw_l) {
    for (int i = bd_tid(); i < 16 * g.D; i += blockDim.x) {
        const int row = i / g.D, f = i - row * g.D;
        float best = 0.f;
        int arg = 0;
        if (row0 + row < rows) {
            const float* p = dense + (size_t)(row0 + row) * ld + f * g.C;
            for (int c = 0; c < g.C; ++c)
                if (fabsf(p[c]) > fabsf(best)) { best = p[c]; arg = c; }
        }
        sidx_l[i] = arg;
        sw_l[i] = best;
    }
}

// dense one-hot rows (scaled) from the indices: dst rows are global, row stride ld.  One thread per (row, factor) writes
// its C floats (16-byte stores when C and the row stride allow): no per-element index arithmetic.
__device__ __forceinline__ void write_onehot(const CatGeo& g, const int* __restrict__ sidx_l, const float* __restrict__ sw_l,
                                             const float* __restrict__ scale_l, float* __restrict__ dst_row0, size_t ld,
                                             int rows_valid) {
    const bool vec = (g.C & 3) == 0 && (ld & 3) == 0 && (((uintptr_t)dst_row0) & 15) == 0;
    for (int i = bd_tid(); i < 16 * g.D; i += blockDim.x) {
        const int row = i / g.D, f = i - row * g.D;
        if (row >= rows_valid) continue;
        const int hot = sidx_l[i];
        float v = sw_l[i];
        if (scale_l) v *= scale_l[row];
        float* p = dst_row0 + (size_t)row * ld + f * g.C;
        if (vec) {
            for (int c = 0; c < g.C; c += 4) {
                floatx4 o = floatx4{0.f, 0.f, 0.f, 0.f};
                if ((hot & ~3) == c) o[hot & 3] = v;
                *reinterpret_cast<floatx4*>(p + c) = o;
            }
        } else {
            for (int c = 0; c < g.C; ++c) p[c] = (c == hot) ? v : 0.f;
        }
    }
}

}  // namespace bd
