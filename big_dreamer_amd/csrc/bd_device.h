// bd_device.h -- device-side building blocks shared by every kernel of the Dreamer hot path (gfx950).
//
// Design:
//  * A workgroup owns a tile of 16*RT rows (batch elements / imagined trajectories).  Rows are independent
//    through the whole recurrence, so a persistent workgroup walks all time steps without any
//    inter-workgroup synchronisation.
//  * Activations of the tile live in LDS in *MFMA fragment order*: for a [16 x K] tile
//        frag[kb][lane][i] = X[row = lane&15][k = kb*16 + 4*(lane>>4) + i]
//    so the A operand of four consecutive v_mfma_f32_16x16x4_f32 is one conflict-free ds_read_b128.
//  * Weights are streamed from L2 in the matching packed order (bigdreamer_hip.h), one coalesced
//    global_load_dwordx4 (1 KiB per wave) per 16x16 block; they are never staged in LDS (each wave
//    owns different output columns, nothing would be shared).
//  * fp32 in, fp32 accumulate: v_mfma_f32_16x16x4_f32 is an exact fp32 FMA chain, so results differ
//    from the CPU reference only by summation order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bd {

typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int kWaves = 4;          // waves per workgroup (one per SIMD)
constexpr int kThreads = kWaves * 64;
constexpr int kFragFloats = 256;   // floats per [16 rows x 16 k] fragment block

__device__ __forceinline__ floatx4 mfma16(float a, float b, floatx4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__host__ __device__ __forceinline__ int cdiv(int a, int b) { return (a + b - 1) / b; }

// float index of element (row, k) of a [16 x 16*Kb] fragment tile
__device__ __forceinline__ int frag_idx(int row, int k) {
    return ((k >> 4) * 64 + ((k >> 2) & 3) * 16 + row) * 4 + (k & 3);
}

// ---- math (matching torch CPU fp32 semantics) ---------------------------------------------------
__device__ __forceinline__ float elu(float x) { return x > 0.f ? x : expm1f(x); }
// derivative of ELU expressed through its output y (y<=0 <=> x<=0): 1 or exp(x) = y+1
__device__ __forceinline__ float elu_grad_from_out(float y) { return y > 0.f ? 1.f : y + 1.f; }
__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }
// F.softplus(beta=1, threshold=20)
__device__ __forceinline__ float softplusf(float x) { return x > 20.f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float act_apply(int act, float x) { return act ? elu(x) : x; }

// ---- reductions ------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
// sum over the workgroup; result valid in thread 0.  `red` = kWaves doubles of LDS.
__device__ __forceinline__ double block_sum_d(double v, double* red) {
    v = wave_sum_d(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double s = 0.0;
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x + 63) >> 6;
        for (int w = 0; w < nw; ++w) s += red[w];
    }
    return s;
}

// ---- tile loads ------------------------------------------------------------------------------------
// Fill a fragment tile [RT][Kb] from up to two row-major sources concatenated along k
// (torch.cat([in0, in1], -1)); rows >= M and columns >= w0+w1 are zero.
template <int RT>
__device__ __forceinline__ void load_tile_concat(float* __restrict__ X, int Kb, int row0, int M,
                                                 const float* __restrict__ in0, int ld0, int w0,
                                                 const float* __restrict__ in1, int ld1, int w1,
                                                 float scale = 1.f) {
    const int Kp = Kb * 16;
    for (int idx = threadIdx.x; idx < RT * 16 * Kp; idx += blockDim.x) {
        const int r = idx / Kp, k = idx - r * Kp;
        const int grow = row0 + r;
        float v = 0.f;
        if (grow < M) {
            if (k < w0) v = in0[(size_t)grow * ld0 + k];
            else if (k < w0 + w1) v = in1[(size_t)grow * ld1 + (k - w0)];
        }
        X[(r >> 4) * Kb * kFragFloats + frag_idx(r & 15, k)] = v * scale;
    }
}

// ---- the core: [16*RT x K] (LDS, fragment order) times packed W^T -> accumulators ------------------
// Each wave owns output column blocks nb = wave, wave+4, ...; NI of them are kept in flight to give
// the MFMA pipe independent accumulation chains (16x16x4 f32: 32-cycle issue, 40-cycle dependent).
// epi(rt, nb, acc): lane holds out[row = 4*(lane>>4) + r][col = nb*16 + (lane&15)], r = 0..3.
template <int RT, int NI, class Epi>
__device__ __forceinline__ void tile_linear(const float* __restrict__ X, int Kb, const float* __restrict__ Wp,
                                            const float* __restrict__ bias, int N, Epi&& epi) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int Nb = (N + 15) >> 4;
    const floatx4* __restrict__ X4 = reinterpret_cast<const floatx4*>(X) + lane;
    const floatx4* __restrict__ W4 = reinterpret_cast<const floatx4*>(Wp) + lane;
    for (int nb0 = wave; nb0 < Nb; nb0 += kWaves * NI) {
        floatx4 acc[NI][RT];
        const floatx4* wp[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int nb = nb0 + i * kWaves;
            const int nbc = nb < Nb ? nb : Nb - 1;          // clamp: stay in bounds, result discarded
            wp[i] = W4 + (size_t)nbc * Kb * 64;
            const int col = nb * 16 + (lane & 15);
            const float b = (bias != nullptr && col < N) ? bias[col] : 0.f;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[i][rt] = floatx4{b, b, b, b};
        }
        for (int kb = 0; kb < Kb; ++kb) {
            floatx4 b4[NI], a4[RT];
#pragma unroll
            for (int i = 0; i < NI; ++i) b4[i] = wp[i][kb * 64];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) a4[rt] = X4[(rt * Kb + kb) * 64];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) acc[i][rt] = mfma16(a4[rt][j], b4[i][j], acc[i][rt]);
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int nb = nb0 + i * kWaves;
            if (nb < Nb) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) epi(rt, nb, acc[i][rt]);
            }
        }
    }
}

// Same contraction accumulated on top of caller-provided accumulators for ONE column block; used where
// several weight matrices feed the same output element (GRU gates, split mean/std heads).
template <int RT>
__device__ __forceinline__ void tile_accum(const float* __restrict__ X, int Kb, const float* __restrict__ Wp,
                                           int nb, floatx4 (&acc)[RT]) {
    const int lane = threadIdx.x & 63;
    const floatx4* __restrict__ X4 = reinterpret_cast<const floatx4*>(X) + lane;
    const floatx4* __restrict__ W4 = reinterpret_cast<const floatx4*>(Wp) + lane + (size_t)nb * Kb * 64;
    for (int kb = 0; kb < Kb; ++kb) {
        const floatx4 b4 = W4[kb * 64];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const floatx4 a4 = X4[(rt * Kb + kb) * 64];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[rt] = mfma16(a4[j], b4[j], acc[rt]);
        }
    }
}

// ---- multi-segment contraction ------------------------------------------------------------------------
// out = sum_s X_s * W_s^T + bias: the concatenated inputs of the reference (torch.cat([belief, state]),
// cat(state, action)) are kept as separate LDS fragments with separately packed weight column blocks.
struct Seg {
    const float* X;   // LDS fragment tile [Kb][64][4]
    const float* W;   // packed weights for this column block (out = N, in = 16*Kb)
    int Kb;
};

template <int NI, int NSEG, class Epi>
__device__ __forceinline__ void tile_linear_seg(const Seg (&seg)[NSEG], const float* __restrict__ bias, int N,
                                                Epi&& epi) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int Nb = (N + 15) >> 4;
    for (int nb0 = wave; nb0 < Nb; nb0 += kWaves * NI) {
        floatx4 acc[NI];
        int nbc[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int nb = nb0 + i * kWaves;
            nbc[i] = nb < Nb ? nb : Nb - 1;
            const int col = nb * 16 + (lane & 15);
            const float b = (bias != nullptr && col < N) ? bias[col] : 0.f;
            acc[i] = floatx4{b, b, b, b};
        }
#pragma unroll
        for (int s = 0; s < NSEG; ++s) {
            const floatx4* __restrict__ X4 = reinterpret_cast<const floatx4*>(seg[s].X) + lane;
            const floatx4* __restrict__ W4 = reinterpret_cast<const floatx4*>(seg[s].W) + lane;
            const int Kb = seg[s].Kb;
            for (int kb = 0; kb < Kb; ++kb) {
                const floatx4 a4 = X4[kb * 64];
                floatx4 b4[NI];
#pragma unroll
                for (int i = 0; i < NI; ++i) b4[i] = W4[((size_t)nbc[i] * Kb + kb) * 64];
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < NI; ++i) acc[i] = mfma16(a4[j], b4[i][j], acc[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int nb = nb0 + i * kWaves;
            if (nb < Nb) epi(nb, acc[i]);
        }
    }
}

// Two outputs sharing one column index (mean / raw-std rows of a Gaussian head, or d/dx and d/dh of the
// GRU): out0 = sum_s X_s W0_s^T + bias0, out1 = sum_s X_s W1_s^T + bias1.  The two chains are
// independent, which also hides the MFMA dependent-issue latency.
struct Seg2 {
    const float* X;
    const float* W0;  // may be nullptr: segment does not feed output 0
    const float* W1;  // may be nullptr: segment does not feed output 1
    int Kb;
};

template <int NSEG, class Epi>
__device__ __forceinline__ void tile_linear_dual(const Seg2 (&seg)[NSEG], const float* __restrict__ bias0,
                                                 const float* __restrict__ bias1, int N, Epi&& epi) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int Nb = (N + 15) >> 4;
    for (int nb = wave; nb < Nb; nb += kWaves) {
        const int col = nb * 16 + (lane & 15);
        const float b0 = (bias0 != nullptr && col < N) ? bias0[col] : 0.f;
        const float b1 = (bias1 != nullptr && col < N) ? bias1[col] : 0.f;
        floatx4 acc0 = floatx4{b0, b0, b0, b0}, acc1 = floatx4{b1, b1, b1, b1};
#pragma unroll
        for (int s = 0; s < NSEG; ++s) {
            const floatx4* __restrict__ X4 = reinterpret_cast<const floatx4*>(seg[s].X) + lane;
            const int Kb = seg[s].Kb;
            const size_t off = (size_t)nb * Kb * 64 + lane;
            if (seg[s].W0 != nullptr && seg[s].W1 != nullptr) {
                const floatx4* __restrict__ W0 = reinterpret_cast<const floatx4*>(seg[s].W0) + off;
                const floatx4* __restrict__ W1 = reinterpret_cast<const floatx4*>(seg[s].W1) + off;
                for (int kb = 0; kb < Kb; ++kb) {
                    const floatx4 a4 = X4[kb * 64], p4 = W0[kb * 64], q4 = W1[kb * 64];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc0 = mfma16(a4[j], p4[j], acc0);
                        acc1 = mfma16(a4[j], q4[j], acc1);
                    }
                }
            } else if (seg[s].W0 != nullptr) {
                const floatx4* __restrict__ W0 = reinterpret_cast<const floatx4*>(seg[s].W0) + off;
                for (int kb = 0; kb < Kb; ++kb) {
                    const floatx4 a4 = X4[kb * 64], p4 = W0[kb * 64];
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc0 = mfma16(a4[j], p4[j], acc0);
                }
            } else if (seg[s].W1 != nullptr) {
                const floatx4* __restrict__ W1 = reinterpret_cast<const floatx4*>(seg[s].W1) + off;
                for (int kb = 0; kb < Kb; ++kb) {
                    const floatx4 a4 = X4[kb * 64], q4 = W1[kb * 64];
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc1 = mfma16(a4[j], q4[j], acc1);
                }
            }
        }
        epi(nb, acc0, acc1);
    }
}

// ---- GRU cell (nn.GRUCell, src/models.py:149,252): four accumulators per output column block ----------
//   R  = W_ir x + W_hr h + b_ir + b_hr      Z  = W_iz x + W_hz h + b_iz + b_hz
//   NI = W_in x + b_in                      NH = W_hn h + b_hn
// epi(nb, R, Z, NI, NH): lane holds rows 4*(lane>>4)+r, column nb*16+(lane&15).
struct GruW {
    const float *w_ir, *w_iz, *w_in, *w_hr, *w_hz, *w_hn;   // packed (Be, Be) each
    const float *b_ih, *b_hh;                               // [3*Be]
};

template <class Epi>
__device__ __forceinline__ void gru_tile(const float* __restrict__ X, const float* __restrict__ Hf, int Kb, int Be,
                                         const GruW& w, Epi&& epi) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int Nb = (Be + 15) >> 4;
    const floatx4* __restrict__ X4 = reinterpret_cast<const floatx4*>(X) + lane;
    const floatx4* __restrict__ H4 = reinterpret_cast<const floatx4*>(Hf) + lane;
    for (int nb = wave; nb < Nb; nb += kWaves) {
        const int col = nb * 16 + (lane & 15);
        const bool ok = col < Be;
        const float br = ok ? w.b_ih[col] + w.b_hh[col] : 0.f;
        const float bz = ok ? w.b_ih[Be + col] + w.b_hh[Be + col] : 0.f;
        const float bni = ok ? w.b_ih[2 * Be + col] : 0.f;
        const float bnh = ok ? w.b_hh[2 * Be + col] : 0.f;
        floatx4 R = floatx4{br, br, br, br}, Z = floatx4{bz, bz, bz, bz};
        floatx4 NI = floatx4{bni, bni, bni, bni}, NH = floatx4{bnh, bnh, bnh, bnh};
        const size_t off = (size_t)nb * Kb * 64 + lane;
        const floatx4* __restrict__ Wir = reinterpret_cast<const floatx4*>(w.w_ir) + off;
        const floatx4* __restrict__ Wiz = reinterpret_cast<const floatx4*>(w.w_iz) + off;
        const floatx4* __restrict__ Win = reinterpret_cast<const floatx4*>(w.w_in) + off;
        const floatx4* __restrict__ Whr = reinterpret_cast<const floatx4*>(w.w_hr) + off;
        const floatx4* __restrict__ Whz = reinterpret_cast<const floatx4*>(w.w_hz) + off;
        const floatx4* __restrict__ Whn = reinterpret_cast<const floatx4*>(w.w_hn) + off;
        for (int kb = 0; kb < Kb; ++kb) {
            const floatx4 ax = X4[kb * 64], ah = H4[kb * 64];
            const floatx4 bir = Wir[kb * 64], biz = Wiz[kb * 64], bin = Win[kb * 64];
            const floatx4 bhr = Whr[kb * 64], bhz = Whz[kb * 64], bhn = Whn[kb * 64];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                R = mfma16(ax[j], bir[j], R);
                Z = mfma16(ax[j], biz[j], Z);
                NI = mfma16(ax[j], bin[j], NI);
                NH = mfma16(ah[j], bhn[j], NH);
                R = mfma16(ah[j], bhr[j], R);
                Z = mfma16(ah[j], bhz[j], Z);
            }
        }
        epi(nb, R, Z, NI, NH);
    }
}

// Backward of the two GRU matmuls: DX = W_ir^T dR + W_iz^T dZ + W_in^T dNI, DH = W_hr^T dR + W_hz^T dZ + W_hn^T dNH
struct GruWT {
    const float *wt_ir, *wt_iz, *wt_in, *wt_hr, *wt_hz, *wt_hn;   // packed transposes (Be, Be)
};

template <class Epi>
__device__ __forceinline__ void gru_tile_bwd(const float* __restrict__ dR, const float* __restrict__ dZ,
                                             const float* __restrict__ dNI, const float* __restrict__ dNH, int Kb,
                                             int Be, const GruWT& w, Epi&& epi) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int Nb = (Be + 15) >> 4;
    const floatx4* __restrict__ R4 = reinterpret_cast<const floatx4*>(dR) + lane;
    const floatx4* __restrict__ Z4 = reinterpret_cast<const floatx4*>(dZ) + lane;
    const floatx4* __restrict__ I4 = reinterpret_cast<const floatx4*>(dNI) + lane;
    const floatx4* __restrict__ H4 = reinterpret_cast<const floatx4*>(dNH) + lane;
    for (int nb = wave; nb < Nb; nb += kWaves) {
        floatx4 DX = floatx4{0.f, 0.f, 0.f, 0.f}, DH = floatx4{0.f, 0.f, 0.f, 0.f};
        const size_t off = (size_t)nb * Kb * 64 + lane;
        const floatx4* __restrict__ Wir = reinterpret_cast<const floatx4*>(w.wt_ir) + off;
        const floatx4* __restrict__ Wiz = reinterpret_cast<const floatx4*>(w.wt_iz) + off;
        const floatx4* __restrict__ Win = reinterpret_cast<const floatx4*>(w.wt_in) + off;
        const floatx4* __restrict__ Whr = reinterpret_cast<const floatx4*>(w.wt_hr) + off;
        const floatx4* __restrict__ Whz = reinterpret_cast<const floatx4*>(w.wt_hz) + off;
        const floatx4* __restrict__ Whn = reinterpret_cast<const floatx4*>(w.wt_hn) + off;
        for (int kb = 0; kb < Kb; ++kb) {
            const floatx4 ar = R4[kb * 64], az = Z4[kb * 64], ai = I4[kb * 64], ah = H4[kb * 64];
            const floatx4 bir = Wir[kb * 64], biz = Wiz[kb * 64], bin = Win[kb * 64];
            const floatx4 bhr = Whr[kb * 64], bhz = Whz[kb * 64], bhn = Whn[kb * 64];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                DX = mfma16(ar[j], bir[j], DX);
                DH = mfma16(ar[j], bhr[j], DH);
                DX = mfma16(az[j], biz[j], DX);
                DH = mfma16(az[j], bhz[j], DH);
                DX = mfma16(ai[j], bin[j], DX);
                DH = mfma16(ah[j], bhn[j], DH);
            }
        }
        epi(nb, DX, DH);
    }
}

// element offset inside a fragment tile for the accumulator element (row = 4*(lane>>4)+r, col = nb*16+(lane&15))
__device__ __forceinline__ int acc_frag_off(int nb, int lane, int r) {
    const int c = lane & 15, row = 4 * (lane >> 4) + r;
    return nb * kFragFloats + ((c >> 2) * 16 + row) * 4 + (c & 3);
}

}  // namespace bd
