// conv.hip -- the 64x64 conv stacks of the pixel configurations (CnnImageEncoder src/models.py:527-564,
// ObservationModel src/models.py:319-362) as gather-GEMMs on the fp32 MFMA tile machinery of this library.
//
// Activations are NHWC fp32 images.  Every stride-2 layer of the two stacks, forward and backward, is one of two
// gather patterns in front of the same contraction out[m][n] = sum_k A(m, k) W[n][k] (+ bias, ELU):
//   F  rows = output pixels of a stride-2 VALID convolution; A(m, .) = the k x k x C window, read as k segments of
//      k*C contiguous floats (NHWC makes (kx, ci) contiguous).  Conv2d forward; ConvTranspose2d dgrad.
//   T  rows = one parity class (oy%2, ox%2) of the output pixels of a stride-2 transposed convolution (sub-pixel
//      decomposition: each class is a dense stride-1 convolution with the taps ky = py + 2a, kx = px + 2b);
//      A(m, .) = Ta segments of Tb*C contiguous floats, masked at the image border.  ConvTranspose2d forward; Conv2d
//      dgrad.  The class results are scattered to the stride-2 positions of the output image.
// Weights live in the parameter buffer as (co, ky, kx, ci) for Conv2d and (ci, ky, kx, co) for ConvTranspose2d, so the
// F contractions and both weight-gradient GEMMs (wgrad.hip, gathered `act` operand) use them as plain [N][K] matrices
// and only the T classes need a re-pack (conv_pack_class_kernel).
//
// A workgroup (8 waves) owns 16*RT rows: the gathered A tile goes to LDS in MFMA fragment order once, then the
// RT * Nb (row tile, column block) units are dealt out contiguously over the waves and each run of units on one column
// block streams its packed weight fragments from L2 once (software-pipelined K loop, v_mfma_f32_16x16x4_f32).
#include "bd_device.h"
#include "bd_host.h"
#include <stdlib.h>

namespace bd {

template <int RTC>
struct ConvFrag {
    floatx4 a[RTC];
    floatx4 b;
};

// One segment: column block nb, row tiles rt0 .. rt0+RTC-1 of the LDS tile X[rt][Kb] (fragment order).
// `pre(rt, nb)` returns the epilogue's per-element multipliers (BD_ACT_ELU_GRAD: ELU' from the saved output; 1 otherwise):
// requested BEFORE the contraction, consumed after it.
template <int RTC, class Epi, class Pre>
__device__ __forceinline__ void conv_segment(const float* __restrict__ X, int Kb, const float* __restrict__ Wp,
                                             const float* __restrict__ bias, int N, int bmod, int nb, int rt0, Epi&& epi, Pre&& pre) {
    const int lane = bd_tid() & 63;
    const int col = nb * 16 + (lane & 15);
    const float b = (bias != nullptr && col < N) ? bias[bmod > 0 ? col % bmod : col] : 0.f;
    floatx4 acc[RTC], acc2[RTC];      // two chains per row tile: a lone dependent chain pays 40 cycles per 32-cycle MFMA
    floatx4 mul[RTC];
#pragma unroll
    for (int r = 0; r < RTC; ++r) {
        acc[r] = floatx4{b, b, b, b};
        acc2[r] = floatx4{0.f, 0.f, 0.f, 0.f};
        mul[r] = pre(rt0 + r, nb);
    }
    const floatx4* __restrict__ X4 = reinterpret_cast<const floatx4*>(X) + lane;
    const floatx4* __restrict__ W4 = reinterpret_cast<const floatx4*>(Wp) + lane + (size_t)nb * Kb * 64;
    pipelined_k<2>(
        Kb,
        [&](int kb) {
            ConvFrag<RTC> f;
            f.b = W4[(size_t)kb * 64];
#pragma unroll
            for (int r = 0; r < RTC; ++r) f.a[r] = X4[((rt0 + r) * Kb + kb) * 64];
            return f;
        },
        [&](const ConvFrag<RTC>& f) {
#pragma unroll
            for (int r = 0; r < RTC; ++r) {
                acc[r] = mfma16(f.a[r][0], f.b[0], acc[r]);
                acc2[r] = mfma16(f.a[r][1], f.b[1], acc2[r]);
            }
#pragma unroll
            for (int r = 0; r < RTC; ++r) {
                acc[r] = mfma16(f.a[r][2], f.b[2], acc[r]);
                acc2[r] = mfma16(f.a[r][3], f.b[3], acc2[r]);
            }
        });
#pragma unroll
    for (int r = 0; r < RTC; ++r) epi(rt0 + r, nb, acc[r] + acc2[r], mul[r]);
}

template <int RT>
__global__ __launch_bounds__(kThreads) void conv_gemm_kernel(bd_conv_args a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int Kb = cdiv(a.K, 16), Nb = cdiv(a.N, 16);
    float* X = smem;                                            // [RT][Kb] fragment tiles
    int* rowoff = reinterpret_cast<int*>(smem + (size_t)RT * Kb * kFragFloats);   // output element offset per row, -1: none
    int* rowflag = rowoff + 16 * RT;
    const int M = a.imgs * a.gh * a.gw;
    const int row0 = blockIdx.x * 16 * RT;
    constexpr int kRows = 16 * RT;
    constexpr int kTpr = kThreads / kRows;                      // threads per row of the gather
    // ---- gather the A tile ----
    {
        const int tid = threadIdx.x;
        const int r = tid / kTpr, j0 = tid - r * kTpr;
        const int m = row0 + r;
        const bool rok = m < M;
        int img = 0, y = 0, x = 0;
        if (rok) {
            img = m / (a.gh * a.gw);
            const int rem = m - img * a.gh * a.gw;
            y = rem / a.gw;
            x = rem - y * a.gw;
        }
        if (j0 == 0) {
            rowoff[r] = rok ? ((img * a.OH + y * a.osy + a.oy0) * a.OW + x * a.osx + a.ox0) * a.ldo : -1;
            rowflag[r] = (2 * y + 1 < a.OH ? 1 : 0) | (2 * x + 1 < a.OW ? 2 : 0);     // fused classes: odd row / column exist
        }
        const int iy0 = y * a.sy + a.y0, ix0 = x * a.sx + a.x0;
        const float* __restrict__ base = a.in + ((size_t)img * a.IH * a.IW) * a.C;
        float* __restrict__ Xr = X + (size_t)(r >> 4) * Kb * kFragFloats;
        const int rr = r & 15;
        const int Kp = Kb * 16;
        // (segment, offset) of k advance incrementally: one division per thread, not one per element
        if (a.vec4) {      // C % 4 == 0: segments are 16-byte aligned runs
            int s = (4 * j0) / a.seglen, off = 4 * j0 - s * a.seglen;
            for (int k = 4 * j0; k < Kp; k += 4 * kTpr) {
                floatx4 v = floatx4{0.f, 0.f, 0.f, 0.f};
                if (rok && k < a.K) {
                    const int iy = iy0 + s * a.ss, ix = ix0 + (off >> a.cshift);
                    if (!a.mask || (iy >= 0 && iy < a.IH && ix >= 0 && ix < a.IW))
                        v = *reinterpret_cast<const floatx4*>(base + ((size_t)iy * a.IW + ix0) * a.C + off);
                }
                *reinterpret_cast<floatx4*>(Xr + frag_idx(rr, k)) = v;
                off += 4 * kTpr;
                while (off >= a.seglen) {
                    off -= a.seglen;
                    ++s;
                }
            }
        } else {
            int s = j0 / a.seglen, off = j0 - s * a.seglen;
            for (int k = j0; k < Kp; k += kTpr) {
                float v = 0.f;
                if (rok && k < a.K) {
                    const int iy = iy0 + s * a.ss;
                    v = base[((size_t)iy * a.IW + ix0) * a.C + off];       // (unmasked pattern only: host checks)
                }
                Xr[frag_idx(rr, k)] = v;
                off += kTpr;
                while (off >= a.seglen) {
                    off -= a.seglen;
                    ++s;
                }
            }
        }
    }
    lds_barrier();
    // ---- contraction: RT * Nb units dealt out contiguously over the waves ----
    const int nrt = min(RT, cdiv(M - row0, 16));
    const int tidc = bd_tid();
    const int lane = tidc & 63, wave = bd_wave(tidc);
    const int U = nrt * Nb;
    const int ub = U / kWaves, urem = U - ub * kWaves;
    const int u0 = wave * ub + min(wave, urem), u1 = u0 + ub + (wave < urem ? 1 : 0);
    auto outpos = [&](int nb, int& coff, int& need) {
        const int col = nb * 16 + (lane & 15);
        coff = col;
        need = 0;
        if (a.fuse_cq > 0) {            // column = class * Cq + channel: pixel (2y + py, 2x + px)
            const int cls = col / a.fuse_cq, py = cls >> 1, px = cls & 1;
            coff = (py * a.OW + px) * a.ldo + (col - cls * a.fuse_cq);
            need = py | (px << 1);
        }
        return col < a.N;
    };
    auto pre = [&](int rt, int nb) {          // BD_ACT_ELU_GRAD: ELU'(saved output) of this lane's four elements
        floatx4 m = floatx4{1.f, 1.f, 1.f, 1.f};
        int coff, need;
        if (a.act != BD_ACT_ELU_GRAD || !outpos(nb, coff, need)) return m;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = rt * 16 + 4 * (lane >> 4) + r;
            const int off = rowoff[row];
            if (off >= 0 && (rowflag[row] & need) == need) m[r] = elu_grad_from_out(a.aux[(size_t)off + coff]);
        }
        return m;
    };
    auto epi = [&](int rt, int nb, floatx4 acc, floatx4 mul) {
        int coff, need;
        if (!outpos(nb, coff, need)) return;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = rt * 16 + 4 * (lane >> 4) + r;
            const int off = rowoff[row];
            if (off >= 0 && (rowflag[row] & need) == need)
                a.out[(size_t)off + coff] = a.act == BD_ACT_ELU ? elu(acc[r]) : acc[r] * mul[r];
        }
    };
    int u = u0;
    while (u < u1) {
        const int nb = u / nrt, rt0 = u - nb * nrt;
        const int cnt = min(min(u1 - u, nrt - rt0), 4);
        if (cnt == 4) conv_segment<4>(X, Kb, a.w, a.bias, a.N, a.fuse_cq, nb, rt0, epi, pre);
        else if (cnt == 3) conv_segment<3>(X, Kb, a.w, a.bias, a.N, a.fuse_cq, nb, rt0, epi, pre);
        else if (cnt == 2) conv_segment<2>(X, Kb, a.w, a.bias, a.N, a.fuse_cq, nb, rt0, epi, pre);
        else conv_segment<1>(X, Kb, a.w, a.bias, a.N, a.fuse_cq, nb, rt0, epi, pre);
        u += cnt;
    }
}

// Packed weights of one parity class of a T pattern from the stored (outer, ky, kx, inner) tensor:
//   dst[n = inner][k = (a, b', outer)] = src[outer][py + 2a][px + 2(Tb-1-b')][inner]        (zero padded blocks)
__global__ __launch_bounds__(256) void conv_pack_class_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                              int Couter, int Cinner, int ksz, int py, int px, int Ta,
                                                              int Tb) {
    const int K = Ta * Tb * Couter, N = Cinner;
    const int Nb = (N + 15) >> 4, Kb = (K + 15) >> 4;
    const int total = Nb * Kb * 64;
    floatx4* __restrict__ d4 = reinterpret_cast<floatx4*>(dst);
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int lane = e & 63, blk = e >> 6;
        const int nb = blk / Kb, kb = blk - nb * Kb;
        const int n = nb * 16 + (lane & 15);
        const int k0 = kb * 16 + 4 * (lane >> 4);
        floatx4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = k0 + i;
            float xv = 0.f;
            if (n < N && k < K) {
                const int c = k % Couter, t = k / Couter;
                const int bp = t % Tb, ta = t / Tb;
                const int ky = py + 2 * ta, kx = px + 2 * (Tb - 1 - bp);
                xv = src[(((size_t)c * ksz + ky) * ksz + kx) * Cinner + n];
            }
            v[i] = xv;
        }
        d4[e] = v;
    }
}

// All four parity classes in one weight matrix (bd_conv_args.fuse_cq): they read the same T x T input window, T = (ksz+1)/2:
//   dst[n = cls*Cinner + c][k = (a, b', outer)] = src[outer][py + 2a][px + 2(T-1-b')][c], 0 where ky or kx >= ksz
__global__ __launch_bounds__(256) void conv_pack_fused_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                              int Couter, int Cinner, int ksz) {
    const int T = (ksz + 1) >> 1;
    const int K = T * T * Couter, N = 4 * Cinner;
    const int Nb = (N + 15) >> 4, Kb = (K + 15) >> 4;
    const int total = Nb * Kb * 64;
    floatx4* __restrict__ d4 = reinterpret_cast<floatx4*>(dst);
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int lane = e & 63, blk = e >> 6;
        const int nb = blk / Kb, kb = blk - nb * Kb;
        const int n = nb * 16 + (lane & 15);
        const int k0 = kb * 16 + 4 * (lane >> 4);
        const int cls = n / Cinner, c = n - cls * Cinner, py = cls >> 1, px = cls & 1;
        floatx4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = k0 + i;
            float xv = 0.f;
            if (n < N && k < K) {
                const int co = k % Couter, t = k / Couter;
                const int bp = t % T, ta = t / T;
                const int ky = py + 2 * ta, kx = px + 2 * (T - 1 - bp);
                if (ky < ksz && kx < ksz) xv = src[(((size_t)co * ksz + ky) * ksz + kx) * Cinner + c];
            }
            v[i] = xv;
        }
        d4[e] = v;
    }
}

// g <- g * ELU'(y) from the saved ELU outputs (in place): the pre-activation gradient of a conv layer
__global__ __launch_bounds__(256) void elu_backward_kernel(float* __restrict__ g, const float* __restrict__ y, size_t n4) {
    floatx4* __restrict__ g4 = reinterpret_cast<floatx4*>(g);
    const floatx4* __restrict__ y4 = reinterpret_cast<const floatx4*>(y);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        floatx4 gv = g4[i];
        const floatx4 yv = y4[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) gv[j] *= elu_grad_from_out(yv[j]);
        g4[i] = gv;
    }
}

// (imgs, C, H*W) <-> (imgs, H*W, C)
__global__ __launch_bounds__(256) void layout_kernel(const float* __restrict__ src, float* __restrict__ dst, int imgs, int C,
                                                     int HW, int to_nhwc) {
    const size_t total = (size_t)imgs * C * HW;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        // e indexes the DESTINATION (coalesced writes)
        const size_t img = e / ((size_t)C * HW), rem = e - img * (size_t)C * HW;
        if (to_nhwc) {
            const int p = (int)(rem / C), c = (int)(rem - (size_t)p * C);
            dst[e] = src[(img * C + c) * HW + p];
        } else {
            const int c = (int)(rem / HW), p = (int)(rem - (size_t)c * HW);
            dst[e] = src[(img * HW + p) * C + c];
        }
    }
}

// Column sums of an [M x N] row-major matrix (bias gradient of a transposed-conv layer: sum over all output pixels),
// N <= 256: stage 1 = per-workgroup partial sums over a contiguous slab of rows, stage 2 = fixed-order sum of the
// partials (bitwise reproducible).
constexpr int kColsumBlocks = 1024;
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ rows, size_t M, int N, int Np,
                                                             float* __restrict__ partial) {
    __shared__ float red[256];
    const int c = threadIdx.x & (Np - 1), r0 = threadIdx.x / Np, rstep = 256 / Np;
    const size_t per = (M + gridDim.x - 1) / gridDim.x;
    const size_t m0 = (size_t)blockIdx.x * per, m1 = m0 + per < M ? m0 + per : M;
    float acc = 0.f;
    if (c < N)
        for (size_t m = m0 + r0; m < m1; m += rstep) acc += rows[m * N + c];
    red[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < Np) {
        float t = 0.f;
        for (int r = 0; r < rstep; ++r) t += red[r * Np + threadIdx.x];
        if (threadIdx.x < N) partial[(size_t)blockIdx.x * N + threadIdx.x] = t;
    }
}
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ partial, int nblocks, int N,
                                                           float* __restrict__ out) {
    // one workgroup per column; 256 threads take every 256th partial, then a fixed-order tree (reproducible)
    __shared__ float red[256];
    const int c = blockIdx.x;
    float t = 0.f;
    for (int b = threadIdx.x; b < nblocks; b += 256) t += partial[(size_t)b * N + c];
    red[threadIdx.x] = t;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[c] = red[0];
}

// ---- patch form: the input patch of a 2-D output tile goes to LDS ONCE ---------------------------------------------
// The gather form above builds im2col rows: every input element is fetched k*k/4 (F) or Ta*Tb (T) times from L2, and
// with few output channels (N = 3 / 32 / 64) those fetches, not the MFMAs, are the time (ConvT 32->3: 5 TFLOP/s).
// Here a workgroup owns RT x 16 output pixels of ONE image (RT grid rows x 16 consecutive grid columns): it copies the
// (RT*sy + span_y) x (16*sx + span_x) x C input patch to LDS (pixel stride C + 4 floats: the 16 lanes of a fragment read
// hit distinct bank groups), and the MFMA A fragment of (row tile, K block = (tap, 16 channels)) is one ds_read_b128 at
//     rt_base[rt] + koff[kb] + (lane & 15) * sx * (C + 4) + 4 * (lane >> 4)
// -- no im2col image at all.  Needs C % 16 == 0.  Same packed weights, same unit distribution, same epilogue.
template <int RTC, class Epi, class Pre>
__device__ __forceinline__ void patch_segment(const float* __restrict__ Pt, const int* __restrict__ koff,
                                              const int* __restrict__ rt_base, int rstride, int Kb,
                                              const float* __restrict__ Wp, const float* __restrict__ bias, int N, int bmod,
                                              int nb, int rt0, Epi&& epi, Pre&& pre) {
    const int lane = bd_tid() & 63;
    const int col = nb * 16 + (lane & 15);
    const float b = (bias != nullptr && col < N) ? bias[bmod > 0 ? col % bmod : col] : 0.f;
    floatx4 acc[RTC], acc2[RTC], mul[RTC];
    int abase[RTC];
#pragma unroll
    for (int r = 0; r < RTC; ++r) {
        acc[r] = floatx4{b, b, b, b};
        acc2[r] = floatx4{0.f, 0.f, 0.f, 0.f};
        abase[r] = rt_base[rt0 + r] + (lane & 15) * rstride + 4 * (lane >> 4);
        mul[r] = pre(rt0 + r, nb);
    }
    const floatx4* __restrict__ W4 = reinterpret_cast<const floatx4*>(Wp) + lane + (size_t)nb * Kb * 64;
    pipelined_k<2>(
        Kb,
        [&](int kb) {
            ConvFrag<RTC> f;
            f.b = W4[(size_t)kb * 64];
            const int ko = koff[kb];
#pragma unroll
            for (int r = 0; r < RTC; ++r) f.a[r] = *reinterpret_cast<const floatx4*>(Pt + abase[r] + ko);
            return f;
        },
        [&](const ConvFrag<RTC>& f) {
#pragma unroll
            for (int r = 0; r < RTC; ++r) {
                acc[r] = mfma16(f.a[r][0], f.b[0], acc[r]);
                acc2[r] = mfma16(f.a[r][1], f.b[1], acc2[r]);
            }
#pragma unroll
            for (int r = 0; r < RTC; ++r) {
                acc[r] = mfma16(f.a[r][2], f.b[2], acc[r]);
                acc2[r] = mfma16(f.a[r][3], f.b[3], acc2[r]);
            }
        });
#pragma unroll
    for (int r = 0; r < RTC; ++r) epi(rt0 + r, nb, acc[r] + acc2[r], mul[r]);
}

template <int RT>
__global__ __launch_bounds__(kThreads) void conv_patch_kernel(bd_conv_args a, int tiles_y, int tiles_x, int PH, int PW) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int Kb = a.K >> 4, Nb = cdiv(a.N, 16);
    const int Cp = a.C + 4;
    float* Pt = smem;                                              // [PH][PW][Cp]
    int* koff = reinterpret_cast<int*>(Pt + (size_t)PH * PW * Cp); // [Kb]
    int* rt_base = koff + Kb;                                      // [RT]
    int* rowoff = rt_base + RT;                                    // [RT*16]
    int* rowflag = rowoff + 16 * RT;                               // [RT*16]
    const int bx = blockIdx.x;
    const int img = bx / (tiles_y * tiles_x), trem = bx - img * tiles_y * tiles_x;
    const int y0 = (trem / tiles_x) * RT, x0 = (trem % tiles_x) * 16;       // tile origin in the row grid
    const int nseg_b = a.seglen / a.C;                                      // taps along x
    // input pixel of grid point (y, x), tap (s, b): iy = y*sy + y0_ + s*ss, ix = x*sx + x0_ + b
    const int ya = (y0 * a.sy + a.y0) + (a.ss < 0 ? -(a.nseg - 1) : 0);     // first patch row in the image
    const int xa = x0 * a.sx + a.x0;                                        // first patch column
    // ---- patch -> LDS (zero outside the image) ----
    const int c4n = a.C >> 2;
    for (int e = threadIdx.x; e < PH * PW * c4n; e += blockDim.x) {
        const int c4 = e % c4n, pix = e / c4n;
        const int px = pix % PW, py = pix / PW;
        const int iy = ya + py, ix = xa + px;
        floatx4 v = floatx4{0.f, 0.f, 0.f, 0.f};
        if (iy >= 0 && iy < a.IH && ix >= 0 && ix < a.IW)
            v = *reinterpret_cast<const floatx4*>(a.in + (((size_t)img * a.IH + iy) * a.IW + ix) * a.C + 4 * c4);
        *reinterpret_cast<floatx4*>(Pt + (size_t)pix * Cp + 4 * c4) = v;
    }
    const int cbn = a.C >> 4;
    for (int kb = threadIdx.x; kb < Kb; kb += blockDim.x) {
        const int cb = kb % cbn, t = kb / cbn;
        const int b = t % nseg_b, s = t / nseg_b;
        const int prow = (a.ss < 0 ? (a.nseg - 1 - s) : s);                 // patch row offset of tap s
        koff[kb] = (prow * PW + b) * Cp + cb * 16;
    }
    for (int r = threadIdx.x; r < RT * 16; r += blockDim.x) {
        const int y = y0 + (r >> 4), x = x0 + (r & 15);
        rowoff[r] = (y < a.gh && x < a.gw) ? ((img * a.OH + y * a.osy + a.oy0) * a.OW + x * a.osx + a.ox0) * a.ldo : -1;
        rowflag[r] = (2 * y + 1 < a.OH ? 1 : 0) | (2 * x + 1 < a.OW ? 2 : 0);
        if ((r & 15) == 0) rt_base[r >> 4] = ((r >> 4) * a.sy * PW) * Cp;
    }
    lds_barrier();
    const int nrt = min(RT, a.gh - y0);
    const int tidc = bd_tid();
    const int lane = tidc & 63, wave = bd_wave(tidc);
    const int U = nrt * Nb;
    const int ub = U / kWaves, urem = U - ub * kWaves;
    const int u0 = wave * ub + min(wave, urem), u1 = u0 + ub + (wave < urem ? 1 : 0);
    const int rstride = a.sx * Cp;
    auto outpos = [&](int nb, int& coff, int& need) {
        const int col = nb * 16 + (lane & 15);
        coff = col;
        need = 0;
        if (a.fuse_cq > 0) {            // column = class * Cq + channel: pixel (2y + py, 2x + px)
            const int cls = col / a.fuse_cq, py = cls >> 1, px = cls & 1;
            coff = (py * a.OW + px) * a.ldo + (col - cls * a.fuse_cq);
            need = py | (px << 1);
        }
        return col < a.N;
    };
    auto pre = [&](int rt, int nb) {          // BD_ACT_ELU_GRAD: ELU'(saved output) of this lane's four elements
        floatx4 m = floatx4{1.f, 1.f, 1.f, 1.f};
        int coff, need;
        if (a.act != BD_ACT_ELU_GRAD || !outpos(nb, coff, need)) return m;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = rt * 16 + 4 * (lane >> 4) + r;
            const int off = rowoff[row];
            if (off >= 0 && (rowflag[row] & need) == need) m[r] = elu_grad_from_out(a.aux[(size_t)off + coff]);
        }
        return m;
    };
    auto epi = [&](int rt, int nb, floatx4 acc, floatx4 mul) {
        int coff, need;
        if (!outpos(nb, coff, need)) return;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = rt * 16 + 4 * (lane >> 4) + r;
            const int off = rowoff[row];
            if (off >= 0 && (rowflag[row] & need) == need)
                a.out[(size_t)off + coff] = a.act == BD_ACT_ELU ? elu(acc[r]) : acc[r] * mul[r];
        }
    };
    int u = u0;
    while (u < u1) {
        const int nb = u / nrt, rt0 = u - nb * nrt;
        const int cnt = min(min(u1 - u, nrt - rt0), 2);
        if (cnt == 2) patch_segment<2>(Pt, koff, rt_base, rstride, Kb, a.w, a.bias, a.N, a.fuse_cq, nb, rt0, epi, pre);
        else patch_segment<1>(Pt, koff, rt_base, rstride, Kb, a.w, a.bias, a.N, a.fuse_cq, nb, rt0, epi, pre);
        u += cnt;
    }
}

template <int RT>
static int launch_patch(const bd_conv_args& a, hipStream_t s) {
    const int span_y = a.ss < 0 ? a.nseg - 1 : a.nseg - a.sy;              // extra patch rows beyond RT*sy
    const int nseg_b = a.seglen / a.C;
    const int span_x = nseg_b - a.sx;
    const int PH = RT * a.sy + span_y, PW = 16 * a.sx + span_x;
    const int Kb = a.K >> 4;
    const size_t lds = ((size_t)PH * PW * (a.C + 4) + Kb + RT + 32 * RT) * sizeof(float);
    BD_REQUIRE(lds <= (size_t)kMaxLds, "bd_conv_gemm(patch): needs %zu B of LDS", lds);
    if (lds > 64 * 1024 && allow_big_lds(conv_patch_kernel<RT>)) return -1;
    const int tiles_y = cdiv(a.gh, RT), tiles_x = cdiv(a.gw, 16);
    hipLaunchKernelGGL(conv_patch_kernel<RT>, dim3((unsigned)(a.imgs * tiles_y * tiles_x)), dim3(kThreads), lds, s, a, tiles_y,
                       tiles_x, PH, PW);
    BD_CHECK_LAUNCH("bd_conv_gemm(patch)");
    return 0;
}

template <int RT>
static int launch_conv(const bd_conv_args& a, hipStream_t s) {
    const int Kb = cdiv(a.K, 16);
    const size_t lds = ((size_t)RT * Kb * kFragFloats + 32 * RT) * sizeof(float);
    BD_REQUIRE(lds <= (size_t)kMaxLds, "bd_conv_gemm: K=%d needs %zu B of LDS at %d rows per workgroup", a.K, lds, 16 * RT);
    if (lds > 64 * 1024 && allow_big_lds(conv_gemm_kernel<RT>)) return -1;
    const long M = (long)a.imgs * a.gh * a.gw;
    hipLaunchKernelGGL(conv_gemm_kernel<RT>, dim3((unsigned)cdiv((int)M, 16 * RT)), dim3(kThreads), lds, s, a);
    BD_CHECK_LAUNCH("bd_conv_gemm");
    return 0;
}


// ---- thin-image forward: stride-2 VALID convolution of a C <= 4 channel image into 32 channels ------------------------
// Conv2d(3 -> 32, k4) forward (src/models.py:538) and the dgrad of ConvTranspose2d(32 -> 3, k6) (src/models.py:347: a k6
// convolution of the 3-channel image gradient): K = 48 / 108, N = 32, 2.2-2.4 million output pixels -- 0.05-0.1 ms of MFMA
// work that the row-tile gather (conv_gemm_kernel<8>: K = 48 floats per row, scalar gathers, a barrier per 128 rows) ran
// in 0.5-1.1 ms.  Same scheme as the thin-image weight gradients (wgrad.hip): every WAVE is its own pipeline over the grid
// rows it = wave, wave + 8, ... of the workgroup's images; the k image rows under a grid row are ONE contiguous piece
// (k x IW x C floats) that the wave DMAs into its private, double-buffered LDS band; the whole weight matrix lives in
// registers (B operand of step s: W[n][4 s + (lane >> 4)]); the A operand of pixel x, k index (ky, kx, c) is one
// ds_read_b32 at band[ky][(2 x + kx) C + c].  No workgroup barrier at all.
constexpr int kThinSteps = 27;            // K <= 108
__host__ __device__ inline int thin_band_al(int k, int roww) { return (k * roww + 64 + 255) & ~255; }

template <int KS>
__global__ __launch_bounds__(kThreads) void conv_thin_f_kernel(const float* __restrict__ in, int imgs, int IH, int IW, int C,
                                                               int kk, const float* __restrict__ W, int ldw, int K,
                                                               const float* __restrict__ bias, int act,
                                                               const float* __restrict__ aux, float* __restrict__ out, int gh, int gw,
                                                               int ipw) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    typedef const __attribute__((address_space(1))) void* glb_ptr_t;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int roww = IW * C, band = kk * roww, band_al = thin_band_al(kk, roww);
    float* mine = lds + (size_t)wave * 2 * band_al;
    for (int i = lane; i < 2 * band_al; i += 64) mine[i] = 0.f;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int kq = lane >> 4, n = lane & 15;
    float w0[KS], w1[KS];
    int aoff[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int k = 4 * s + kq;
        const bool ok = k < K;
        w0[s] = ok ? W[(size_t)n * ldw + k] : 0.f;
        w1[s] = ok ? W[(size_t)(16 + n) * ldw + k] : 0.f;
        const int ky = k / (kk * C);
        aoff[s] = ok ? ky * roww + (k - ky * kk * C) : 0;
    }
    const float b0 = bias ? bias[n] : 0.f, b1 = bias ? bias[16 + n] : 0.f;
    const int img0 = blockIdx.x * ipw, img1 = min(imgs, img0 + ipw);
    const int items = (img1 - img0) * gh;
    const int npc = cdiv(band, 256);
    auto issue = [&](int it, float* buf) {
        const int img = img0 + it / gh, y = it - (it / gh) * gh;
        const float* sb = in + ((size_t)img * IH + 2 * y) * roww;
        for (int ch = 0; ch < npc; ++ch)
            if (4 * lane + 256 * ch < band)
                __builtin_amdgcn_global_load_lds((glb_ptr_t)(sb + 4 * lane + 256 * ch), (lds_ptr_t)(buf + 256 * ch), 16, 0, 0);
    };
    auto wait_keep_newest = [&]() {
        switch (npc) {
            case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        }
    };
    const int nrt = cdiv(gw, 16);
    int it = wave, b = 0;
    if (it < items) issue(it, mine);
    for (; it < items; it += kWaves, b ^= 1) {
        // (the output stores of the previous item are still in flight: vmcnt counts them too, and retires in order, so the
        //  waits below also cover them -- a few hundred cycles per item that the other wave of the SIMD fills)
        const int img = img0 + it / gh, y = it - (it / gh) * gh;
        const size_t obase = ((size_t)img * gh + y) * gw * 32;
        // BD_ACT_ELU_GRAD: the multipliers of this grid row, requested BEFORE the next band's DMA (vmcnt retires in order: a
        // load issued after the DMA could only be waited for together with it)
        float mul0[2][4], mul1[2][4];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                mul0[rt][r] = mul1[rt][r] = 1.f;
                const int x = rt * 16 + 4 * kq + r;
                if (act == BD_ACT_ELU_GRAD && x < gw) {
                    mul0[rt][r] = elu_grad_from_out(aux[obase + x * 32 + n]);
                    mul1[rt][r] = elu_grad_from_out(aux[obase + x * 32 + 16 + n]);
                }
            }
        if (it + kWaves < items) {
            issue(it + kWaves, mine + (b ^ 1) * band_al);
            wait_keep_newest();
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const float* Bd = mine + b * band_al;
        float* orow = out + obase;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            if (rt >= nrt) break;
            const float* Bx = Bd + 2 * (rt * 16 + n) * C;              // A operand row m = lane & 15 -> pixel x = 16 rt + m
            floatx4 acc0 = floatx4{b0, b0, b0, b0}, acc1 = floatx4{b1, b1, b1, b1};
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const float a = Bx[aoff[s]];
                acc0 = mfma16(a, w0[s], acc0);
                acc1 = mfma16(a, w1[s], acc1);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int x = rt * 16 + 4 * kq + r;
                if (x < gw) {
                    orow[x * 32 + n] = act == BD_ACT_ELU ? elu(acc0[r]) : acc0[r] * mul0[rt][r];
                    orow[x * 32 + 16 + n] = act == BD_ACT_ELU ? elu(acc1[r]) : acc1[r] * mul1[rt][r];
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // this band is refilled by the next iteration's DMA
    }
}

}  // namespace bd

extern "C" {
using namespace bd;

int bd_conv_gemm(const bd_conv_args* a, void* stream) {
    BD_REQUIRE(a && a->in && a->out && a->w && a->imgs > 0 && a->gh > 0 && a->gw > 0 && a->N > 0 && a->K > 0 &&
                   a->nseg > 0 && a->seglen > 0 && a->nseg * a->seglen == a->K && a->C > 0 && a->IH > 0 && a->IW > 0 &&
                   a->OH > 0 && a->OW > 0 && a->ldo >= (a->fuse_cq > 0 ? a->fuse_cq : a->N),
               "bd_conv_gemm: bad arguments");
    BD_REQUIRE((long)a->imgs * a->gh * a->gw < (1L << 31) && (long)a->imgs * a->OH * a->OW * a->ldo < (1L << 31),
               "bd_conv_gemm: image batch too large for 32-bit element offsets");
    BD_REQUIRE(!a->mask || (a->vec4 && (1 << a->cshift) == a->C), "bd_conv_gemm: masked gathers need C a power of two >= 4");
    BD_REQUIRE(a->act != BD_ACT_ELU_GRAD || a->aux, "bd_conv_gemm: BD_ACT_ELU_GRAD needs the saved outputs (aux)");
    BD_REQUIRE(a->fuse_cq == 0 || (a->mask && a->N == 4 * a->fuse_cq && a->osy == 2 && a->osx == 2 && a->oy0 == 0 && a->ox0 == 0),
               "bd_conv_gemm: fused classes need pattern T with N = 4*fuse_cq");
    BD_REQUIRE(!a->vec4 || (a->C % 4 == 0 && a->seglen % 4 == 0), "bd_conv_gemm: vec4 gathers need C, seglen multiples of 4");
    // patch form (input patch staged in LDS once per 2-D tile): channels a multiple of 16, a grid at least 12 wide (a
    // row tile is 16 consecutive grid columns) and few enough output channels that the gather traffic matters
    static const char* patch_env = getenv("BD_CONV_PATCH");
    const bool patch_ok = !(patch_env && patch_env[0] == '0') && a->C % 16 == 0 && a->gw >= 12 && a->N <= 128 &&
                          (a->mask ? (a->sy == 1 && a->sx == 1 && a->ss == -1) : (a->sy == 2 && a->sx == 2 && a->ss == 1 && a->x0 == 0 && a->y0 == 0));
    if (patch_ok) {
        const int Nb = cdiv(a->N, 16);
        int rt = 8;                                   // >= 8 (row tile, column block) units, patch <= ~56 KB
        while (rt > 1 && (rt / 2) * Nb >= 8) rt >>= 1;
        auto patch_bytes = [&](int r) {
            const int span_y = a->ss < 0 ? a->nseg - 1 : a->nseg - a->sy, span_x = a->seglen / a->C - a->sx;
            return (size_t)(r * a->sy + span_y) * (16 * a->sx + span_x) * (a->C + 4) * sizeof(float);
        };
        while (rt > 1 && patch_bytes(rt) > 56 * 1024) rt >>= 1;
        if (rt > a->gh) while (rt > 1 && rt / 2 >= a->gh) rt >>= 1;
        switch (rt) {
            case 8: return launch_patch<8>(*a, (hipStream_t)stream);
            case 4: return launch_patch<4>(*a, (hipStream_t)stream);
            case 2: return launch_patch<2>(*a, (hipStream_t)stream);
            default: return launch_patch<1>(*a, (hipStream_t)stream);
        }
    }
    // rows per workgroup: as many 16-row tiles as LDS holds (<= 8): the packed weights (K x N) are streamed once per
    // workgroup, 8*RT FLOP per byte.  BD_CONV_RT caps it (tuning).
    const int Kb = cdiv(a->K, 16);
    static const char* cap_env = getenv("BD_CONV_RT");
    int rt = 8;
    if (cap_env && atoi(cap_env) >= 1 && atoi(cap_env) < 8) rt = atoi(cap_env) >= 4 ? 4 : (atoi(cap_env) >= 2 ? 2 : 1);
    while (rt > 1 && ((size_t)rt * Kb * kFragFloats + 32 * rt) * sizeof(float) > 150 * 1024) rt >>= 1;
    switch (rt) {
        case 8: return launch_conv<8>(*a, (hipStream_t)stream);
        case 4: return launch_conv<4>(*a, (hipStream_t)stream);
        case 2: return launch_conv<2>(*a, (hipStream_t)stream);
        default: return launch_conv<1>(*a, (hipStream_t)stream);
    }
}

int bd_conv_pack_class(const float* src, float* dst, int Couter, int Cinner, int ksz, int py, int px, int Ta, int Tb,
                       void* stream) {
    BD_REQUIRE(src && dst && Couter > 0 && Cinner > 0 && ksz > 0 && Ta > 0 && Tb > 0 && py + 2 * (Ta - 1) < ksz &&
                   px + 2 * (Tb - 1) < ksz, "bd_conv_pack_class: bad arguments");
    hipLaunchKernelGGL(conv_pack_class_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream, src, dst, Couter, Cinner, ksz, py,
                       px, Ta, Tb);
    BD_CHECK_LAUNCH("bd_conv_pack_class");
    return 0;
}

int bd_conv_pack_fused(const float* src, float* dst, int Couter, int Cinner, int ksz, void* stream) {
    BD_REQUIRE(src && dst && Couter > 0 && Cinner > 0 && ksz > 0, "bd_conv_pack_fused: bad arguments");
    hipLaunchKernelGGL(conv_pack_fused_kernel, dim3(128), dim3(256), 0, (hipStream_t)stream, src, dst, Couter, Cinner, ksz);
    BD_CHECK_LAUNCH("bd_conv_pack_fused");
    return 0;
}

int bd_elu_backward(float* g, const float* y, size_t n, void* stream) {
    BD_REQUIRE(g && y && n > 0 && (n & 3) == 0, "bd_elu_backward: bad arguments (n must be a multiple of 4)");
    const size_t n4 = n >> 2;
    const int blocks = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
    hipLaunchKernelGGL(elu_backward_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g, y, n4);
    BD_CHECK_LAUNCH("bd_elu_backward");
    return 0;
}

size_t bd_colsum_ws_floats(int N) { return (size_t)bd::kColsumBlocks * (size_t)N; }

int bd_colsum(const float* rows, size_t M, int N, float* out, float* ws, void* stream) {
    BD_REQUIRE(rows && out && ws && M > 0 && N > 0 && N <= 256, "bd_colsum: bad arguments (N <= 256)");
    int Np = 1;
    while (Np < N) Np <<= 1;
    const int nb = (int)(M < (size_t)kColsumBlocks * 64 ? (M + 63) / 64 : kColsumBlocks);
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, rows, M, N, Np, ws);
    BD_CHECK_LAUNCH("bd_colsum");
    hipLaunchKernelGGL(colsum_final_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, ws, nb, N, out);
    BD_CHECK_LAUNCH("bd_colsum(final)");
    return 0;
}

int bd_image_layout(const float* src, float* dst, int imgs, int C, int HW, int to_nhwc, void* stream) {
    BD_REQUIRE(src && dst && imgs > 0 && C > 0 && HW > 0, "bd_image_layout: bad arguments");
    const size_t total = (size_t)imgs * C * HW;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(layout_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, dst, imgs, C, HW, to_nhwc);
    BD_CHECK_LAUNCH("bd_image_layout");
    return 0;
}

int bd_conv_thin_forward(const float* in, int imgs, int IH, int IW, int C, int k, const float* W, int ldw, const float* bias,
                         int act, const float* aux, float* out, void* stream) {
    using namespace bd;
    BD_REQUIRE(in && W && out && imgs > 0 && IH > 0 && IW > 0 && C >= 1 && C <= 4 && k >= 2 && IH >= k && IW >= k,
               "bd_conv_thin_forward: bad arguments");
    BD_REQUIRE(act != BD_ACT_ELU_GRAD || aux, "bd_conv_thin_forward: BD_ACT_ELU_GRAD needs the saved outputs (aux)");
    const int K = k * k * C, gh = (IH - k) / 2 + 1, gw = (IW - k) / 2 + 1, roww = IW * C;
    BD_REQUIRE(K <= 4 * kThinSteps && ldw >= K && gw <= 32, "bd_conv_thin_forward: K = %d (<= %d), output width %d (<= 32)", K,
               4 * kThinSteps, gw);
    BD_REQUIRE((roww & 3) == 0 && ((uintptr_t)in & 15) == 0, "bd_conv_thin_forward: image rows must be 16-byte aligned");
    BD_REQUIRE((size_t)imgs * gh * gw * 32 < (1ull << 31), "bd_conv_thin_forward: image batch too large");
    const size_t lds = (size_t)kWaves * 2 * thin_band_al(k, roww) * sizeof(float);
    const int ipw = cdiv(imgs, 512);                      // two workgroups per CU (<= 80 KB of LDS each)
    const int grid = cdiv(imgs, ipw);
    hipStream_t s = (hipStream_t)stream;
    if (K <= 48) {
        if (lds > 64 * 1024 && allow_big_lds(conv_thin_f_kernel<12>)) return -1;
        hipLaunchKernelGGL(conv_thin_f_kernel<12>, dim3(grid), dim3(kThreads), lds, s, in, imgs, IH, IW, C, k, W, ldw, K, bias, act, aux,
                           out, gh, gw, ipw);
    } else {
        if (lds > 64 * 1024 && allow_big_lds(conv_thin_f_kernel<kThinSteps>)) return -1;
        hipLaunchKernelGGL(conv_thin_f_kernel<kThinSteps>, dim3(grid), dim3(kThreads), lds, s, in, imgs, IH, IW, C, k, W, ldw, K, bias,
                           act, aux, out, gh, gw, ipw);
    }
    BD_CHECK_LAUNCH("bd_conv_thin_forward");
    return 0;
}

}  // extern "C"
