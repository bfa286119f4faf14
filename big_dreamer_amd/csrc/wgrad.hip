// wgrad.hip -- weight gradients of a Linear layer over all rows of a pass:
//   dW[N x K] = dpre^T[N x M] * act[M x K],   db[N] = sum_m dpre[m][n]
// (what autograd's AddmmBackward computes for every nn.Linear on the path).  TN GEMM with a long
// reduction (M = 2450 .. 34300 rows) and a small output, so the rows are split over workgroups; partial
// tiles go to a slab workspace and are summed in fixed order (bitwise reproducible, no float atomics).
// 64x64 output tile per workgroup, v_mfma_f32_32x32x2_f32 (exact fp32), operands staged through LDS.
#include "bd_device.h"
#include "bd_host.h"
#include <stdint.h>
#include <stdlib.h>

namespace bd {

typedef float floatx16 __attribute__((ext_vector_type(16)));
constexpr int kWT = 64;      // output tile edge
constexpr int kWM = 32;      // rows per LDS stage

__global__ __launch_bounds__(kThreads) void wgrad_kernel(const float* __restrict__ dpre, int ldp,
                                                         const float* __restrict__ act, int lda, int M, int N, int K,
                                                         int has_bias, int rows_per_split, float* __restrict__ ws) {
    __shared__ float P[kWM][kWT];   // dpre chunk  [m][n]
    __shared__ float A[kWM][kWT];   // act chunk   [m][k]
    const int Kext = K + has_bias;
    const int n0 = blockIdx.x * kWT, k0 = blockIdx.y * kWT;
    const int m_begin = blockIdx.z * rows_per_split;
    const int m_end = min(M, m_begin + rows_per_split);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wn = (wave >> 1) * 32, wk = (wave & 1) * 32;
    floatx16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const int c = threadIdx.x & 63, r0 = threadIdx.x >> 6;   // each thread: column c, rows r0, r0+4, ...
    for (int m0 = m_begin; m0 < m_end; m0 += kWM) {
        float pv[kWM / 4], av[kWM / 4];
#pragma unroll
        for (int i = 0; i < kWM / 4; ++i) {
            const int m = m0 + r0 + 4 * i;
            const bool ok = m < m_end;
            pv[i] = (ok && n0 + c < N) ? dpre[(size_t)m * ldp + n0 + c] : 0.f;
            const int k = k0 + c;
            av[i] = ok ? (k < K ? act[(size_t)m * lda + k] : (k == K && has_bias ? 1.f : 0.f)) : 0.f;
        }
        __syncthreads();   // previous stage's reads done
#pragma unroll
        for (int i = 0; i < kWM / 4; ++i) {
            P[r0 + 4 * i][c] = pv[i];
            A[r0 + 4 * i][c] = av[i];
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < kWM / 2; ++s) {
            const float a = P[2 * s + (lane >> 5)][wn + (lane & 31)];
            const float b = A[2 * s + (lane >> 5)][wk + (lane & 31)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
    }
    // D[i][j]: j = lane&31, i = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    float* slab = ws + (size_t)blockIdx.z * N * Kext;
    const int k = k0 + wk + (lane & 31);
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int n = n0 + wn + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        if (n < N && k < Kext) slab[(size_t)n * Kext + k] = acc[reg];
    }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, int splits, int N, int K,
                                                           int has_bias, float* __restrict__ dW, int ldw,
                                                           float* __restrict__ db, int accumulate) {
    const int Kext = K + has_bias;
    const int total = N * Kext;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int z = 0; z < splits; ++z) s += ws[(size_t)z * total + e];
        const int n = e / Kext, k = e - n * Kext;
        float* dst = k < K ? dW + (size_t)n * ldw + k : db + n;
        *dst = accumulate ? *dst + s : s;
    }
}

// ---- grouped form: all weight-gradient GEMMs of one backward pass in ONE launch -----------------------------
// A train step needs 36 of these GEMMs (68 parameter tensors); as separate launches most are latency-bound
// (M = 2450 rows, 60..600 x 3..1024 outputs) and the launch/reduce pairs alone cost ~1.3 ms per step.  The grouped
// kernel walks a descriptor table in HBM: blockIdx.x -> (GEMM, output tile, row split); the grouped reduce sums the
// slabs of every GEMM in fixed order.  A descriptor may take its activations from two sources split at row M1
// (rows whose "previous belief" is the initial state / whose actor input is the start feature), which replaces the
// accumulate pass of the ungrouped form.
__global__ __launch_bounds__(kThreads) void wgrad_grouped_kernel(const bd_wgrad_desc* __restrict__ descs, int n,
                                                                 float* __restrict__ ws) {
    __shared__ float P[kWM][kWT];
    __shared__ float A[kWM][kWT];
    int g = 0;
    while (g + 1 < n && (int)blockIdx.x >= descs[g + 1].block_begin) ++g;     // uniform scan, n is small
    const bd_wgrad_desc d = descs[g];
    const int hb = d.db != nullptr;
    const int Kext = d.K + hb;
    int local = blockIdx.x - d.block_begin;
    const int z = local / (d.tiles_n * d.tiles_k);
    local -= z * d.tiles_n * d.tiles_k;
    const int n0 = (local / d.tiles_k) * kWT, k0 = (local % d.tiles_k) * kWT;
    const int m_begin = z * d.rows_per;
    const int m_end = min(d.M, m_begin + d.rows_per);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wn = (wave >> 1) * 32, wk = (wave & 1) * 32;
    floatx16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const int c = threadIdx.x & 63, r0 = threadIdx.x >> 6;
    float pv[kWM / 4], av[kWM / 4];
    auto fetch = [&](int m0) {
#pragma unroll
        for (int i = 0; i < kWM / 4; ++i) {
            const int m = m0 + r0 + 4 * i;
            const bool ok = m < m_end;
            pv[i] = (ok && n0 + c < d.N) ? d.dpre[(size_t)m * d.ldp + n0 + c] : 0.f;
            const int k = k0 + c;
            float v = 0.f;
            if (ok) {
                if (k < d.K) v = m < d.M1 ? d.act1[(size_t)m * d.lda1 + k] : d.act2[(size_t)(m - d.M1) * d.lda2 + k];
                else if (k == d.K && hb) v = 1.f;
            }
            av[i] = v;
        }
    };
    fetch(m_begin);
    for (int m0 = m_begin; m0 < m_end; m0 += kWM) {
        __syncthreads();   // previous stage's LDS reads are done
#pragma unroll
        for (int i = 0; i < kWM / 4; ++i) {
            P[r0 + 4 * i][c] = pv[i];
            A[r0 + 4 * i][c] = av[i];
        }
        __syncthreads();
        if (m0 + kWM < m_end) fetch(m0 + kWM);      // next chunk's global loads fly under this chunk's MFMAs
#pragma unroll
        for (int s = 0; s < kWM / 2; ++s) {
            const float a = P[2 * s + (lane >> 5)][wn + (lane & 31)];
            const float b = A[2 * s + (lane >> 5)][wk + (lane & 31)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
    }
    float* slab = ws + d.ws_off + (size_t)z * d.N * Kext;
    const int k = k0 + wk + (lane & 31);
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int nn = n0 + wn + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        if (nn < d.N && k < Kext) slab[(size_t)nn * Kext + k] = acc[reg];
    }
}

// ---- wide form of the grouped kernel (default) ----------------------------------------------------------------
// The 64x64 tiles above re-read every operand row once per tile column/row of the output (4x for a 200x200 layer) and
// pad 200 to 256 in both dimensions: 0.31 ms for the five GEMMs of an actor/critic pass whose MFMA floor is 0.09 ms.
// Here a workgroup owns up to 13x13 16-blocks of the output -- a whole 200 x (200+bias) layer -- so each operand row is
// read once per row split, and the 2x4 waves each keep up to 7x4 v_mfma_f32_16x16x4_f32 accumulators:
// per 4-row slice a wave reads 7 + 4 operand dwords from LDS for 28 MFMAs.  Rows are staged 16 at a time through a
// double-buffered LDS image with row stride 208 (== 16 mod 32: the 2 x 16-float rows a half-wave reads fall on
// disjoint banks); the next stage's global loads fly under this stage's MFMAs; one barrier per stage.
constexpr int kWB = 13;               // 16-blocks per workgroup tile edge
constexpr int kWBK = 12;              // ... along K when K needs several tiles: 3 blocks per wave column, no padded MFMAs
constexpr int kWBKDeep = 36;          // ... of a NARROW tile (<= 4 dpre blocks: conv layers with <= 64 channels): 9 blocks per wave column
// A narrow tile keeps few accumulators per wave (2 x 3 at 12 K blocks): 48 MFMAs per wave and 32-row stage against ~2.5k
// cycles of per-stage DMA issue / wait / barrier, and its 64-wide dpre operand was re-read once per 12-block K tile (6 x for
// the 1152-wide decoder layer: 2.9 x the algorithmic bytes, round-2 PMC).  Deep K tiles (up to 36 blocks: 2 x 9
// accumulators per wave) triple the MFMAs per staged row and cut the re-reads to 2 x.
__host__ __device__ __forceinline__ int wgrad_tiles_k(int NB, int KB, bool deep_ok) {
    if (KB <= kWB) return 1;
    return (NB <= 4 && deep_ok) ? cdiv(KB, kWBKDeep) : cdiv(KB, kWBK);
}
// the act operand of this GEMM takes the 16-byte LDS-DMA form (what the deep tiles are built for)
__host__ inline bool wgrad_act16(const bd_wgrad_desc& d) {
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    if ((d.K & 3) != 0 || !al16(d.act1)) return false;
    if (d.g_nseg > 0) return (d.g_C & 3) == 0 && (d.g_seglen & 3) == 0;
    return (d.lda1 & 3) == 0 && (d.M1 == d.M || ((d.lda2 & 3) == 0 && al16(d.act2)));
}
constexpr int kWLd = 240;             // LDS row stride in floats: 14 blocks + pad, == 16 mod 32
constexpr int kWRows = 16;            // rows per stage
constexpr int kWStage = 2 * kWRows * kWLd;   // floats per stage: dpre rows | act rows
constexpr int kWRing = 3;             // stage buffers (DMA runs two stages ahead)
constexpr int kWRowsTall = 32;         // rows per stage of a narrow tile (<= 64 dpre columns): half the stage overhead per row
constexpr int kWLdNarrow = 80;         // its dpre row stride: 64 + 16, == 16 mod 32
constexpr int kWLdMid = 144;           // dpre row stride of a mid tile (<= 128 columns): 8 blocks + 16, == 16 mod 32
constexpr int kWLdDeep = 592;          // act row stride of a deep narrow tile: 36 blocks + pad, == 16 mod 32
constexpr size_t kWideLdsBytes = 147456;   // >= 3 x 16 x (80 + 592) x 4 = 129 024 (deep), 3 x 32 x 320 x 4 = 122 880 (tall narrow),
                                           //    8 waves x 2 x (1280 + 1024) x 4 = 147 456 (thin-image bodies)
static_assert((size_t)kWRing * kWRows * (kWLdNarrow + kWLdDeep) * sizeof(float) <= kWideLdsBytes, "deep stage ring");
static_assert((size_t)kWRing * kWRowsTall * (kWLdMid + kWLd) * sizeof(float) <= kWideLdsBytes, "mid stage ring");
constexpr int kWThreads = 512;        // 8 waves: 2 per SIMD, so LDS latency and the stage barrier hide under the other wave

// WN x WK = 16-blocks per wave (the 2 x 4 waves cover up to 2WN x 4WK blocks).  The MFMA loop is branch-free: a wave
// whose share is smaller multiplies zero-filled LDS columns (the workgroup runs at the pace of its fullest wave
// anyway); only the stores are guarded.
template <int WN, int WK, int ROWS = kWRows, int PLD = kWLd, int ALD = kWLd>
__device__ __forceinline__ void wgrad_wide_body(const bd_wgrad_desc& d, float* __restrict__ ws, float* wlds, int hb, int Kext,
                                                int z, int n0, int k0, int nb_cnt, int kb_cnt) {
    const int m_begin = z * d.rows_per;
    const int m_end = min(d.M, m_begin + d.rows_per);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // 2 (n) x 4 (k) waves; wave = 2*wc + wr so that the two waves a SIMD hosts (w, w+4) carry 7x4+7x3, 6x4+6x3,
    // 7x3+7x3, 6x3+6x3 blocks of a 13x13 tile (the fullest SIMD sets the pace: 49 of 42.25 on average)
    const int wr = wave & 1, wc = wave >> 1;
    const int hn = (nb_cnt + 1) >> 1;
    const int my_nb0 = wr ? hn : 0, my_nb = wr ? nb_cnt - hn : hn;
    const int kq = kb_cnt >> 2, krem = kb_cnt & 3;
    const int my_kb0 = wc * kq + min(wc, krem), my_kb = kq + (wc < krem ? 1 : 0);

    constexpr int kStage = ROWS * (PLD + ALD);      // floats per stage: ROWS dpre rows (stride PLD) | ROWS act rows (stride ALD)
    constexpr int kRpw = ROWS / 8;                  // rows per wave and stage
    static_assert(kWRing * kStage * sizeof(float) <= kWideLdsBytes, "stage ring exceeds the LDS request");
    floatx4 acc[WN][WK];
#pragma unroll
    for (int i = 0; i < WN; ++i)
#pragma unroll
        for (int j = 0; j < WK; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    // Staging by LDS-DMA (global_load_lds_dword: wave-uniform LDS base + lane*4, per-lane source address), so no
    // staging registers compete with the accumulators.  Wave w carries rows 2w, 2w+1 of a stage; lanes cover the
    // columns lane + 64*cc.  Columns beyond the tile / beyond N, K are never written and stay zero from the
    // initial clear; the ones column (bias gradient) is written once; rows beyond the split's last row exist only in
    // its final stage and are cleared there with ordinary LDS stores.
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    typedef const __attribute__((address_space(1))) void* glb_ptr_t;
    const int ncol = min(nb_cnt * 16, d.N - n0), kcol = min(kb_cnt * 16, d.K - k0);   // real columns of this tile
    for (int i = threadIdx.x; i < kWRing * kStage; i += kWThreads) wlds[i] = 0.f;
    __syncthreads();
    if (hb && d.K >= k0 && d.K < k0 + kb_cnt * 16 && threadIdx.x < kWRing * ROWS)
        wlds[(threadIdx.x / ROWS) * kStage + ROWS * PLD + (threadIdx.x % ROWS) * ALD + (d.K - k0)] = 1.f;
    // 16-byte LDS-DMA (one 1 KiB wave-instruction per operand row) wherever rows and columns allow it: the dword form
    // moves 256 B per instruction and ran the whole kernel at ~1.2 TB/s of operand traffic, not at its MFMA rate.
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    const bool p4 = (ncol & 3) == 0 && (d.ldp & 3) == 0 && al16(d.dpre + n0);
    const bool gathered = d.g_nseg > 0;
    const bool a4 = (kcol & 3) == 0 && (k0 & 3) == 0 &&
                    (gathered ? ((d.g_C & 3) == 0 && (d.g_seglen & 3) == 0 && al16(d.act1))
                              : ((d.lda1 & 3) == 0 && al16(d.act1) && (d.M1 == d.M || ((d.lda2 & 3) == 0 && al16(d.act2)))));
    // gathered operand: the window offset of this lane's columns does not depend on the row -- once per workgroup
    // (16-byte form: chunk cc = the 256 floats 4 * lane + 256 * cc; dword form: the 64 floats lane + 64 * cc)
    constexpr int kCh4 = (ALD + 255) / 256;         // 16-byte chunks per act row (1 for the 13-block tiles, 3 for deep ones)
    constexpr int kCh1 = (ALD + 63) / 64;           // dword chunks per act row
    constexpr int kGo = kCh1;
    int goff[kGo];
#pragma unroll
    for (int cc = 0; cc < kGo; ++cc) goff[cc] = 0;
    if (gathered) {
#pragma unroll
        for (int cc = 0; cc < kGo; ++cc) {
            const int k = k0 + (a4 ? 4 * lane + 256 * cc : lane + 64 * cc), sgm = k / d.g_seglen;
            goff[cc] = sgm * d.g_IW * d.g_C + (k - sgm * d.g_seglen);
        }
    }
    auto issue = [&](float* buf, int m0) {
#pragma unroll
        for (int rr = 0; rr < kRpw; ++rr) {
            const int m = m0 + wave * kRpw + rr;                   // wave-uniform
            float* P = buf + (wave * kRpw + rr) * PLD;
            float* A = buf + ROWS * PLD + (wave * kRpw + rr) * ALD;
            if (m < m_end) {
                const float* prow = d.dpre + (size_t)m * d.ldp + n0;
                if (p4) {
                    if (4 * lane < ncol) __builtin_amdgcn_global_load_lds((glb_ptr_t)(prow + 4 * lane), (lds_ptr_t)P, 16, 0, 0);
                } else {
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) {
                        const int c = lane + 64 * cc;
                        if (c < ncol) __builtin_amdgcn_global_load_lds((glb_ptr_t)(prow + c), (lds_ptr_t)(P + 64 * cc), 4, 0, 0);
                    }
                }
                const float* abase;
                if (gathered) {       // the k x k x C window of output pixel m (conv.hip, pattern F)
                    const int img = m / (d.g_gh * d.g_gw), rem = m - img * d.g_gh * d.g_gw;
                    const int y = rem / d.g_gw, x = rem - y * d.g_gw;
                    abase = d.act1 + (((size_t)img * d.g_IH + 2 * y) * d.g_IW + 2 * x) * d.g_C;
                } else {
                    abase = (m < d.M1 ? d.act1 + (size_t)m * d.lda1 : d.act2 + (size_t)(m - d.M1) * d.lda2) + k0;
                }
                if (a4) {
#pragma unroll
                    for (int cc = 0; cc < kCh4; ++cc)
                        if (4 * lane + 256 * cc < kcol)
                            __builtin_amdgcn_global_load_lds((glb_ptr_t)(abase + (gathered ? goff[cc] : 4 * lane + 256 * cc)),
                                                             (lds_ptr_t)(A + 256 * cc), 16, 0, 0);
                } else {
#pragma unroll
                    for (int cc = 0; cc < kCh1; ++cc) {
                        const int c = lane + 64 * cc;
                        if (c < kcol)
                            __builtin_amdgcn_global_load_lds((glb_ptr_t)(abase + (gathered ? goff[cc] : c)), (lds_ptr_t)(A + 64 * cc), 4,
                                                             0, 0);
                    }
                }
            } else {
#pragma unroll
                for (int cc = 0; cc < kCh1; ++cc) {
                    const int c = lane + 64 * cc;
                    if (c < PLD) P[c] = 0.f;
                    if (c < ALD) A[c] = 0.f;
                }
            }
        }
    };
    const int nst = cdiv(m_end - m_begin, ROWS);
    // Ring of three stage buffers: the DMA of stage st+2 is issued at the top of stage st, and the end of stage st only
    // waits for stage st+1 (`vmcnt(n)` with n = this wave's DMA instructions per stage leaves the newest stage in
    // flight; vmcnt retires in order).  A narrow tile (N = 32 / 64: conv layers) has ~1.5k cycles of MFMAs per
    // 16-row stage against >= 2.5k cycles of loaded DMA latency: with two buffers every stage waited for its fetch.
    const int n_dma = kRpw * ((p4 ? 1 : cdiv(ncol, 64)) + (a4 ? cdiv(kcol, 256) : cdiv(kcol, 64)));    // 4, 6, ..., 16 per wave and full stage
    auto wait_keep_newest = [&]() {
        switch (n_dma) {
            case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
            case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
            case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
            case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
            case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
        }
    };
    __syncthreads();
    issue(wlds, m_begin);
    if (nst > 1) issue(wlds + kStage, m_begin + ROWS);
    __builtin_amdgcn_s_waitcnt(0);     // vmcnt(0) lgkmcnt(0): both have landed
    __syncthreads();
    const int lrow = lane >> 4, lcol = lane & 15;
    for (int st = 0; st < nst; ++st) {
        const float* buf = wlds + (st % kWRing) * kStage;
        // buffer (st+2) % 3 was last read in stage st-1, which every wave has left (barrier below)
        const bool more = st + 2 < nst;
        [[maybe_unused]] const int sb = (st == 10 || st == 11) ? (st - 10) * 8 : -1;   // diagnostic stamps (-DBD_STAMPS)
        BD_DSTAMP(sb, 0);
        if (more) issue(wlds + ((st + 2) % kWRing) * kStage, m_begin + (st + 2) * ROWS);
        BD_DSTAMP(sb, 1);
        const float* Pb = buf + lrow * PLD + my_nb0 * 16 + lcol;
        const float* Ab = buf + ROWS * PLD + lrow * ALD + my_kb0 * 16 + lcol;
#pragma unroll
        for (int sl = 0; sl < ROWS / 4; ++sl) {
            float a[WN], b[WK];
#pragma unroll
            for (int i = 0; i < WN; ++i) a[i] = Pb[sl * 4 * PLD + i * 16];
#pragma unroll
            for (int j = 0; j < WK; ++j) b[j] = Ab[sl * 4 * ALD + j * 16];
#pragma unroll
            for (int i = 0; i < WN; ++i)
#pragma unroll
                for (int j = 0; j < WK; ++j) acc[i][j] = mfma16(a[i], b[j], acc[i][j]);
        }
        BD_DSTAMP(sb, 2);
        // stage st+1 must have landed; a full stage st+2 (no tail rows: st + 3 < nst) may stay in flight
        if (more && st + 3 < nst) wait_keep_newest();
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        BD_DSTAMP(sb, 3);
        lds_barrier();                 // lgkmcnt(0) (tail-row clears) + s_barrier, without draining the DMA
        BD_DSTAMP(sb, 4);
    }
    // lane holds D[n = 4*(lane>>4) + r][k = lane&15] of each 16x16 block
    float* slab = ws + d.ws_off + (size_t)z * d.N * Kext;
#pragma unroll
    for (int i = 0; i < WN; ++i) {
        if (i < my_nb) {
#pragma unroll
            for (int j = 0; j < WK; ++j) {
                if (j < my_kb) {
                    const int k = k0 + (my_kb0 + j) * 16 + lcol;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int nn = n0 + (my_nb0 + i) * 16 + 4 * lrow + r;
                        if (nn < d.N && k < Kext) slab[(size_t)nn * Kext + k] = acc[i][j][r];
                    }
                }
            }
        }
    }
}

// Dense form of the wide body for the layers of the 200-wide chains: plain row-major operands whose rows and columns
// are 16-byte aligned (no gathered window), a tile 13 blocks tall.  Two things differ from the general body: (i) every
// WAVE runs the instantiation that fits its own share exactly (7|6 x 4|3|2|1 blocks) instead of the workgroup's
// largest -- on a 13 x 13 tile the padded 7 x 4 body issues 224 block-MFMAs per slice for 169 useful ones -- and the
// wave -> share table pairs the shares so that the fullest SIMD carries 46 blocks (28 + 18), not 49; (ii) the DMA of a
// stage is four instructions per wave off per-lane base pointers set up once (the general body re-derives alignment,
// gather geometry and source selection per stage: 1.1-1.5k cycles per stage, s_memtime stamps).
template <int WN, int WK>
__device__ __forceinline__ void wgrad_dense_body(const bd_wgrad_desc& d, float* __restrict__ ws, float* wlds, int hb, int Kext,
                                                 int z, int n0, int k0, int nb_cnt, int kb_cnt, int my_nb0, int my_kb0) {
    const int m_begin = z * d.rows_per;
    const int m_end = min(d.M, m_begin + d.rows_per);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    floatx4 acc[WN][WK];
#pragma unroll
    for (int i = 0; i < WN; ++i)
#pragma unroll
        for (int j = 0; j < WK; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    typedef const __attribute__((address_space(1))) void* glb_ptr_t;
    const int ncol = min(nb_cnt * 16, d.N - n0), kcol = min(kb_cnt * 16, d.K - k0);
    for (int i = threadIdx.x; i < kWRing * kWStage; i += kWThreads) wlds[i] = 0.f;
    __syncthreads();
    if (hb && d.K >= k0 && d.K < k0 + kb_cnt * 16 && threadIdx.x < kWRing * kWRows)
        wlds[(threadIdx.x >> 4) * kWStage + kWRows * kWLd + (threadIdx.x & 15) * kWLd + (d.K - k0)] = 1.f;
    const bool lp = 4 * lane < ncol, la = 4 * lane < kcol;
    const float* __restrict__ pl = d.dpre + n0 + 4 * lane;
    const float* __restrict__ a1l = d.act1 + k0 + 4 * lane;
    const float* __restrict__ a2l = (d.M1 < d.M ? d.act2 : d.act1) + k0 + 4 * lane;
    auto issue = [&](float* buf, int m0) {
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int m = m0 + wave * 2 + rr;                      // wave-uniform
            float* P = buf + (wave * 2 + rr) * kWLd;
            float* A = P + kWRows * kWLd;
            if (m < m_end) {
                if (lp) __builtin_amdgcn_global_load_lds((glb_ptr_t)(pl + (size_t)m * d.ldp), (lds_ptr_t)P, 16, 0, 0);
                const float* ar = m < d.M1 ? a1l + (size_t)m * d.lda1 : a2l + (size_t)(m - d.M1) * d.lda2;
                if (la) __builtin_amdgcn_global_load_lds((glb_ptr_t)ar, (lds_ptr_t)A, 16, 0, 0);
            } else {
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                    const int c = lane + 64 * cc;
                    if (c < kWLd) {
                        P[c] = 0.f;
                        A[c] = 0.f;
                    }
                }
            }
        }
    };
    const int nst = cdiv(m_end - m_begin, kWRows);
    __syncthreads();
    issue(wlds, m_begin);
    if (nst > 1) issue(wlds + kWStage, m_begin + kWRows);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    const int lrow = lane >> 4, lcol = lane & 15;
    for (int st = 0; st < nst; ++st) {
        const float* buf = wlds + (st % kWRing) * kWStage;
        const bool more = st + 2 < nst;
        [[maybe_unused]] const int sb = (st == 10 || st == 11) ? (st - 10) * 8 : -1;   // diagnostic stamps (-DBD_STAMPS)
        BD_DSTAMP(sb, 0);
        // The two waves of a SIMD (w, w + 4) issue the DMA of stage st+2 at different points of the stage: one before its
        // MFMAs, the other after its second slice -- each wave's DMA instructions then go out while its SIMD-mate streams
        // MFMAs, instead of both issuing first with the matrix pipe idle.
        if (more && wave < 4) issue(wlds + ((st + 2) % kWRing) * kWStage, m_begin + (st + 2) * kWRows);
        BD_DSTAMP(sb, 1);
        const float* Pb = buf + lrow * kWLd + my_nb0 * 16 + lcol;
        const float* Ab = buf + kWRows * kWLd + lrow * kWLd + my_kb0 * 16 + lcol;
#pragma unroll
        for (int sl = 0; sl < kWRows / 4; ++sl) {
            float a[WN], b[WK];
#pragma unroll
            for (int i = 0; i < WN; ++i) a[i] = Pb[sl * 4 * kWLd + i * 16];
#pragma unroll
            for (int j = 0; j < WK; ++j) b[j] = Ab[sl * 4 * kWLd + j * 16];
#pragma unroll
            for (int i = 0; i < WN; ++i)
#pragma unroll
                for (int j = 0; j < WK; ++j) acc[i][j] = mfma16(a[i], b[j], acc[i][j]);
            if (sl == 1 && more && wave >= 4) issue(wlds + ((st + 2) % kWRing) * kWStage, m_begin + (st + 2) * kWRows);
        }
        BD_DSTAMP(sb, 2);
        // stage st+1 must have landed; a full stage st+2 (4 DMA instructions of this wave) may stay in flight
        if (more && st + 3 < nst) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        BD_DSTAMP(sb, 3);
        lds_barrier();
        BD_DSTAMP(sb, 4);
    }
    float* slab = ws + d.ws_off + (size_t)z * d.N * Kext;
#pragma unroll
    for (int i = 0; i < WN; ++i) {
#pragma unroll
        for (int j = 0; j < WK; ++j) {
            const int k = k0 + (my_kb0 + j) * 16 + lcol;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int nn = n0 + (my_nb0 + i) * 16 + 4 * lrow + r;
                if (nn < d.N && k < Kext) slab[(size_t)nn * Kext + k] = acc[i][j][r];
            }
        }
    }
}

// ---- thin-image conv weight gradients: the two layers that touch the 3-channel 64 x 64 image -----------------------------
// Conv2d(3 -> 32, k4) (dpre = gradient of its output, 31 x 31 grid) and ConvTranspose2d(32 -> 3, k6) (dpre = its input, 30 x 30
// grid): N = 32, K = 48 / 108, M = 2.2-2.4 MILLION rows -- 0.05 ms of MFMA work and 0.4 GB of operands each, which the
// staged-row bodies above turned into 1.1-1.2 ms (tools/conv_probe.py: a 32-row stage moves two 64-192-byte DMA pieces per
// row and pays a workgroup barrier for 48 MFMAs per wave).  Here every WAVE is its own pipeline: it owns the grid rows
// it = wave, wave + 8, ... of the workgroup's images, and for a grid row it DMAs the two CONTIGUOUS pieces it needs -- the
// band of k image rows under that grid row (k x 768 B) and the row's dpre values (gw x 128 B) -- into its private,
// double-buffered LDS region (1 KiB wave-instructions, fragment-agnostic), then forms both MFMA operands with ds_read_b32:
//   A (dpre):  lane (n = l & 15, q = l >> 4)  ->  dpre[x = 4 s + q][16 nb + n]
//   B (window): lane (j = l & 15, q)          ->  band[ky][(2 x + kx) C + c]   with (ky, kx, c) of k index 16 kb + j
// No workgroup barrier until the final cross-wave sum; the bias gradient is a VALU column sum of the A values.
// Pixels x >= gw of the last slice multiply zero dpre rows (the region's tail is zero-filled once and never written).
constexpr int kThinMaxKB = 7;                 // K <= 112
constexpr int kThinDpreFloats = 1024;         // gw <= 32 pixels x 32 channels
__host__ __device__ inline int thin_band_floats(int k, int roww) { return (k * roww + 64 + 255) & ~255; }
__host__ inline bool wgrad_thin_ok(const bd_wgrad_desc& d) {
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    if (d.g_nseg <= 0 || d.g_C > 4 || d.N != 32 || d.ldp != 32 || d.g_gw > 32) return false;
    if (cdiv(d.K, 16) > kThinMaxKB || d.g_seglen != d.g_nseg * d.g_C) return false;
    const int roww = d.g_IW * d.g_C;
    if ((roww & 3) != 0 || !al16(d.act1) || !al16(d.dpre)) return false;
    // 8 waves x 2 buffers x (band + dpre row) must fit the kernel's LDS request, and the final partials too
    const size_t per_wave = 2 * (size_t)(thin_band_floats(d.g_nseg, roww) + kThinDpreFloats) * sizeof(float);
    return 8 * per_wave <= kWideLdsBytes && (size_t)8 * 2 * kThinMaxKB * 1024 <= kWideLdsBytes;
}

#ifndef BD_WGRAD_TALL_MID
#define BD_WGRAD_TALL_MID 1
#endif
__device__ __forceinline__ bool tall_mid() { return BD_WGRAD_TALL_MID != 0; }

template <int KB>
__device__ __forceinline__ void wgrad_thin_body(const bd_wgrad_desc& d, float* __restrict__ ws, float* wlds, int z) {
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    typedef const __attribute__((address_space(1))) void* glb_ptr_t;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hb = d.db != nullptr;
    const int Kext = d.K + hb;
    const int gh = d.g_gh, gw = d.g_gw, C = d.g_C, kk = d.g_nseg;
    const int roww = d.g_IW * C, band = kk * roww, drow = gw * 32;
    const int band_al = thin_band_floats(kk, roww);
    const int per_buf = band_al + kThinDpreFloats;
    float* mine = wlds + (size_t)wave * 2 * per_buf;
    for (int i = lane; i < 2 * per_buf; i += 64) mine[i] = 0.f;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int imgs = d.M / (gh * gw), ipw = d.rows_per / (gh * gw);
    const int img0 = z * ipw, img1 = min(imgs, img0 + ipw);
    const int items = (img1 - img0) * gh;                       // (image, grid row) pairs of this workgroup
    const int nband = cdiv(band, 256), ndrow = cdiv(drow, 256); // 1 KiB DMA pieces
    auto issue = [&](int it, float* buf) {
        const int img = img0 + it / gh, y = it - (it / gh) * gh;
        const float* sb = d.act1 + ((size_t)img * d.g_IH + 2 * y) * roww;
        const float* sd = d.dpre + ((size_t)img * gh + y) * gw * 32;
        for (int ch = 0; ch < nband; ++ch)
            if (4 * lane + 256 * ch < band)
                __builtin_amdgcn_global_load_lds((glb_ptr_t)(sb + 4 * lane + 256 * ch), (lds_ptr_t)(buf + 256 * ch), 16, 0, 0);
        for (int ch = 0; ch < ndrow; ++ch)
            if (4 * lane + 256 * ch < drow)
                __builtin_amdgcn_global_load_lds((glb_ptr_t)(sd + 4 * lane + 256 * ch), (lds_ptr_t)(buf + band_al + 256 * ch), 16, 0, 0);
    };
    const int n_dma = nband + ndrow;                            // per item and wave (7 or 9)
    auto wait_keep_newest = [&]() {
        switch (n_dma) {
            case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
            case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
            case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
            case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
            case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
            case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        }
    };
    // lane constants of the window operand: k index 16 kb + (lane & 15) -> (ky, kx, c) -> ky * roww + kx * C + c
    int boff[KB];
    bool bok[KB];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        const int kidx = kb * 16 + (lane & 15);
        bok[kb] = kidx < d.K;
        const int ky = kidx / d.g_seglen;
        boff[kb] = bok[kb] ? ky * roww + (kidx - ky * d.g_seglen) : 0;
    }
    floatx4 acc[2][KB];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < KB; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
    float bsum[2] = {0.f, 0.f};
    const int q = lane >> 4, n = lane & 15;
    const int nsl = cdiv(gw, 4);
    int it = wave, b = 0;
    if (it < items) issue(it, mine);
    for (; it < items; it += 8, b ^= 1) {
        const bool more = it + 8 < items;
        if (more) {
            issue(it + 8, mine + (b ^ 1) * per_buf);
            wait_keep_newest();
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const float* Bd = mine + b * per_buf;
        const float* Dp = Bd + band_al;
        for (int sl = 0; sl < nsl; ++sl) {
            const int x = 4 * sl + q;
            const float a0 = Dp[x * 32 + n], a1 = Dp[x * 32 + 16 + n];
            float bv[KB];
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) bv[kb] = bok[kb] ? Bd[boff[kb] + 2 * x * C] : 0.f;
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                acc[0][kb] = mfma16(a0, bv[kb], acc[0][kb]);
                acc[1][kb] = mfma16(a1, bv[kb], acc[1][kb]);
            }
            bsum[0] += a0;
            bsum[1] += a1;
        }
        // this buffer is refilled by the DMA issued at the top of the NEXT iteration: its LDS reads must have returned
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    // ---- cross-wave sum (fixed order) -> this split's slab ----
    __syncthreads();
    floatx4* R4 = reinterpret_cast<floatx4*>(wlds);
    float* BS = wlds + (size_t)8 * 2 * KB * 256;                 // [8 waves][2][64]
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < KB; ++j) R4[((wave * 2 + i) * KB + j) * 64 + lane] = acc[i][j];
        BS[(wave * 2 + i) * 64 + lane] = bsum[i];
    }
    __syncthreads();
    float* slab = ws + d.ws_off + (size_t)z * d.N * Kext;
    for (int e = threadIdx.x; e < 2 * KB * 64; e += blockDim.x) {
        const int blk = e >> 6, l = e & 63;
        floatx4 s4 = floatx4{0.f, 0.f, 0.f, 0.f};
        for (int w = 0; w < 8; ++w) s4 += R4[(w * 2 * KB + blk) * 64 + l];
        const int i = blk / KB, j = blk - i * KB;
        const int k = j * 16 + (l & 15);
        if (k < d.K) {
#pragma unroll
            for (int r = 0; r < 4; ++r) slab[(size_t)(i * 16 + 4 * (l >> 4) + r) * Kext + k] = s4[r];
        }
    }
    if (hb && threadIdx.x < 32) {
        const int i = threadIdx.x >> 4, nn = threadIdx.x & 15;
        float t = 0.f;
        for (int w = 0; w < 8; ++w)
            for (int qq = 0; qq < 4; ++qq) t += BS[(w * 2 + i) * 64 + qq * 16 + nn];
        slab[(size_t)(i * 16 + nn) * Kext + d.K] = t;
    }
}

__global__ __launch_bounds__(kWThreads) void wgrad_wide_kernel(const bd_wgrad_desc* __restrict__ descs, int n,
                                                         float* __restrict__ ws) {
    extern __shared__ float wlds[];   // ring of kWRing stages: [ P: rows x stride | A: rows x kWLd ] (kWideLdsBytes)
    int g = 0;
    while (g + 1 < n && (int)blockIdx.x >= descs[g + 1].block_begin) ++g;     // uniform scan, n is small
    const bd_wgrad_desc d = descs[g];
    const int hb = d.db != nullptr;
    const int Kext = d.K + hb;
    const int NB = cdiv(d.N, 16), KB = cdiv(Kext, 16);
    const int nbw = cdiv(NB, d.tiles_n), kbw = cdiv(KB, d.tiles_k);           // <= kWB (bd_wgrad_plan)
    int local = blockIdx.x - d.block_begin;
    const int z = local / (d.tiles_n * d.tiles_k);
    local -= z * d.tiles_n * d.tiles_k;
    const int tn = local / d.tiles_k, tk = local - tn * d.tiles_k;
    const int n0 = tn * nbw * 16, k0 = tk * kbw * 16;
    const int nb_cnt = min(nbw, NB - tn * nbw), kb_cnt = min(kbw, KB - tk * kbw);
    if (d.g_pad == 1) {          // thin-image conv layer (bd_wgrad_plan: wgrad_thin_ok): wave-private row pipelines
        const int kb_t = cdiv(d.K, 16);
        if (kb_t <= 3) wgrad_thin_body<3>(d, ws, wlds, z);
        else wgrad_thin_body<kThinMaxKB>(d, ws, wlds, z);
        return;
    }
    // The MFMA loop is branch-free over WN x WK blocks per wave, so the instantiation must fit the tile: a 4 x 12-block
    // tile (N = 64: conv layers) on the 7 x 4 body would issue 28 MFMAs per slice for 6 useful ones.
    const int hn = (nb_cnt + 1) >> 1, hk = (kb_cnt + 3) >> 2;      // blocks per wave row / wave column
    {   // dense 13-block-tall tiles: exact per-wave instantiations (wgrad_dense_body)
        auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
        const int ncol = min(nb_cnt * 16, d.N - n0), kcol = min(kb_cnt * 16, d.K - k0);
        const bool dense = nb_cnt == 13 && kb_cnt >= 4 && d.g_nseg == 0 && (ncol & 3) == 0 && (d.ldp & 3) == 0 &&
                           al16(d.dpre + n0) && (kcol & 3) == 0 && (k0 & 3) == 0 && (d.lda1 & 3) == 0 && al16(d.act1) &&
                           (d.M1 == d.M || ((d.lda2 & 3) == 0 && al16(d.act2)));
        if (dense) {
            const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
            // hardware waves w and w+4 share a SIMD: (n half, k quarter) pairs 28+18, 24+18, 21+21, 21+18 on a 13 x 13 tile
            // (n half, k quarter) of wave w:  w0 (0,0)  w1 (1,0)  w2 (0,1)  w3 (0,2)  w4 (1,3)  w5 (1,2)  w6 (0,3)  w7 (1,1)
            const int wr = (0xB2 >> wave) & 1;
            const int c = (0x13232100 >> (4 * wave)) & 3;
            const int kq = kb_cnt >> 2, krem = kb_cnt & 3;
            const int my_nb0 = wr ? 7 : 0, my_nb = wr ? 6 : 7;
            const int my_kb0 = c * kq + min(c, krem), my_kb = kq + (c < krem ? 1 : 0);
#define BD_WD(WN, WK) wgrad_dense_body<WN, WK>(d, ws, wlds, hb, Kext, z, n0, k0, nb_cnt, kb_cnt, my_nb0, my_kb0)
#define BD_WD_ROW(WN)                 \
    do {                              \
        if (my_kb == 4) BD_WD(WN, 4); \
        else if (my_kb == 3) BD_WD(WN, 3); \
        else if (my_kb == 2) BD_WD(WN, 2); \
        else BD_WD(WN, 1);            \
    } while (0)
            if (my_nb == 7) BD_WD_ROW(7);
            else BD_WD_ROW(6);
#undef BD_WD_ROW
#undef BD_WD
            return;
        }
    }
    if (nb_cnt <= 4) {
        // narrow tile (<= 64 dpre columns: conv layers with 32 / 64 output channels, the 3-channel image): a wave holds
        // only 1-2 x 1-4 accumulators, i.e. 4-32 MFMAs per 16-row stage against ~2.5k cycles of DMA issue / wait / barrier
        // per stage -- 32-row stages (compact dpre rows keep the ring inside the LDS) halve that overhead per row
#define BD_WN_BODY(WN, WK) wgrad_wide_body<WN, WK, kWRowsTall, kWLdNarrow>(d, ws, wlds, hb, Kext, z, n0, k0, nb_cnt, kb_cnt)
#define BD_WN_ROW(WN)                      \
    do {                                   \
        if (hk <= 1) BD_WN_BODY(WN, 1);    \
        else if (hk <= 2) BD_WN_BODY(WN, 2); \
        else if (hk <= 3) BD_WN_BODY(WN, 3); \
        else BD_WN_BODY(WN, 4);            \
    } while (0)
        if (hk > 4) {
            // deep K tile (wgrad_tiles_k: up to 36 blocks, 9 per wave column): 16-row stages with a 592-float act row
            // (37 blocks, == 16 mod 32); operands must take the 16-byte DMA form (conv layers with C % 4 == 0 do)
#define BD_WDEEP(WN, WK) wgrad_wide_body<WN, WK, kWRows, kWLdNarrow, kWLdDeep>(d, ws, wlds, hb, Kext, z, n0, k0, nb_cnt, kb_cnt)
            if (hn <= 1) { if (hk <= 8) BD_WDEEP(1, 8); else BD_WDEEP(1, 9); }
            else { if (hk <= 8) BD_WDEEP(2, 8); else BD_WDEEP(2, 9); }
#undef BD_WDEEP
            return;
        }
        if (hn <= 1) BD_WN_ROW(1);
        else BD_WN_ROW(2);
#undef BD_WN_ROW
#undef BD_WN_BODY
        return;
    }
    if (nb_cnt <= 8 && tall_mid()) {
        // mid tile (<= 128 dpre columns: the 128-channel conv layers): 32-row stages too -- a wave holds <= 4 x 3
        // accumulators, 48 MFMAs per 16-row stage against the same ~2.5k cycles of per-stage overhead
#define BD_WM_BODY(WN, WK) wgrad_wide_body<WN, WK, kWRowsTall, kWLdMid>(d, ws, wlds, hb, Kext, z, n0, k0, nb_cnt, kb_cnt)
#define BD_WM_ROW(WN)                      \
    do {                                   \
        if (hk <= 1) BD_WM_BODY(WN, 1);    \
        else if (hk <= 2) BD_WM_BODY(WN, 2); \
        else if (hk <= 3) BD_WM_BODY(WN, 3); \
        else BD_WM_BODY(WN, 4);            \
    } while (0)
        if (hn <= 2) BD_WM_ROW(2);
        else BD_WM_ROW(4);
#undef BD_WM_ROW
#undef BD_WM_BODY
        return;
    }
#define BD_WG_BODY(WN, WK) wgrad_wide_body<WN, WK>(d, ws, wlds, hb, Kext, z, n0, k0, nb_cnt, kb_cnt)
#define BD_WG_ROW(WN)                      \
    do {                                   \
        if (hk <= 1) BD_WG_BODY(WN, 1);    \
        else if (hk <= 2) BD_WG_BODY(WN, 2); \
        else if (hk <= 3) BD_WG_BODY(WN, 3); \
        else BD_WG_BODY(WN, 4);            \
    } while (0)
    if (hn <= 1) BD_WG_ROW(1);
    else if (hn <= 2) BD_WG_ROW(2);
    else if (hn <= 4) BD_WG_ROW(4);
    else BD_WG_ROW(7);
#undef BD_WG_ROW
#undef BD_WG_BODY
}

__global__ __launch_bounds__(256) void wgrad_grouped_reduce_kernel(const bd_wgrad_desc* __restrict__ descs, int n,
                                                                   const float* __restrict__ ws) {
    int g = 0;
    while (g + 1 < n && (int)blockIdx.x >= descs[g + 1].red_begin) ++g;
    const bd_wgrad_desc d = descs[g];
    const int hb = d.db != nullptr;
    const int Kext = d.K + hb;
    const int total = d.N * Kext;
    const int e = (blockIdx.x - d.red_begin) * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const float* p = ws + d.ws_off + e;
    float s = 0.f;
    for (int z = 0; z < d.splits; ++z) s += p[(size_t)z * total];
    const int nn = e / Kext, k = e - nn * Kext;
    if (k < d.K) d.dW[(size_t)nn * d.ldw + k] = s;
    else d.db[nn] = s;
}

#ifdef BD_STAMPS
extern "C" int bd_debug_wstamps(unsigned long long* out64) {
    return hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_dstamps), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : -1;
}
#endif

// BD_WGRAD_WIDE=0 selects the 64x64-tile grouped kernel; BD_WGRAD_ROWS = rows per workgroup of the wide form
static bool wgrad_wide() {
    static const char* e = getenv("BD_WGRAD_WIDE");
    return !(e && e[0] == '0');
}
static bool thin_on() {        // BD_WGRAD_THIN=0: the staged-row bodies for the 3-channel layers too (A/B, tests)
    static const char* e = getenv("BD_WGRAD_THIN");
    return !(e && e[0] == '0');
}
static int wgrad_rows() {
    static const char* e = getenv("BD_WGRAD_ROWS");
    const int r = e ? atoi(e) : 0;
    return r >= kWRows ? cdiv(r, kWRows) * kWRows : 0;      // 0: fit one round (bd_wgrad_plan)
}

static void wgrad_plan(int M, int N, int K, int has_bias, int* splits, int* rows_per) {
    const int tiles = cdiv(N, kWT) * cdiv(K + has_bias, kWT);
    int s = 1024 / tiles;
    const int max_s = cdiv(M, 2 * kWM);
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
    int rp = cdiv(cdiv(M, s), kWM) * kWM;
    *rows_per = rp;
    *splits = cdiv(M, rp);
}

}  // namespace bd

extern "C" {

int bd_wgrad_plan(bd_wgrad_desc* descs, int n, int* total_blocks, int* total_red_blocks, size_t* ws_floats) {
    using namespace bd;
    BD_REQUIRE(descs && n > 0 && total_blocks && total_red_blocks && ws_floats, "bd_wgrad_plan: bad arguments");
    int blocks = 0, red = 0;
    size_t off = 0;
    // wide form: one workgroup per CU (402 registers per lane), so the launch should be ONE round of the chip: the
    // smallest row count per workgroup (multiple of the 16-row stage) for which all (tile, row split) pairs fit 256
    int rows_wide = wgrad_rows();
    if (wgrad_wide() && rows_wide == 0) {
        auto wgs = [&](int R) {
            long t = 0;
            for (int i = 0; i < n; ++i)
                t += (long)cdiv(cdiv(descs[i].N, 16), kWB) * wgrad_tiles_k(cdiv(descs[i].N, 16), cdiv(descs[i].K + (descs[i].db != nullptr), 16), wgrad_act16(descs[i])) *
                     cdiv(descs[i].M > 0 ? descs[i].M : 1, R);
            return t;
        };
        rows_wide = 128;
        while (wgs(rows_wide) > 256 && rows_wide < (1 << 20)) rows_wide += kWRows;
    }
    // Launches that contain conv weight gradients (gathered operand) mix row counts from 2 450 to 2.4 M: one row count
    // for all would leave a single workgroup streaming millions of rows.  There every GEMM gets rows_per = T / cost,
    // cost = blocks on the fullest SIMD of its tile + 12 (per-stage DMA / barrier overhead in the same unit), with T
    // chosen for about R rounds of 256 workgroups, R = total work / (256 x 1024 rows of a full 13 x 13 tile).
    bool any_gather = false;
    for (int i = 0; i < n; ++i) any_gather = any_gather || descs[i].g_nseg > 0;
    auto tile_cost = [](int nb, int kb) {
        const int hn = (nb + 1) >> 1, kq = kb >> 2, kr = kb & 3;
        const int c = hn * ((kq + (0 < kr ? 1 : 0)) + (kq + (2 < kr ? 1 : 0)));
        return (c > 0 ? c : 1) + 12;
    };
    auto desc_geo = [&](const bd_wgrad_desc& d, int* tiles, int* cost) {
        const int NB = cdiv(d.N, 16), KB = cdiv(d.K + (d.db != nullptr), 16);
        const int tn = cdiv(NB, kWB), tk = wgrad_tiles_k(NB, KB, wgrad_act16(d));
        *tiles = tn * tk;
        *cost = tile_cost(cdiv(NB, tn), cdiv(KB, tk));
    };
    // rows per workgroup of descriptor i under budget T (thin-image layers have their own fixed split: whole images)
    auto thin = [&](const bd_wgrad_desc& d) { return wgrad_wide() && thin_on() && wgrad_thin_ok(d); };
    auto rows_for = [&](const bd_wgrad_desc& d, double T) {
        int t, c;
        desc_geo(d, &t, &c);
        long r = (long)(T / c);
        r = (r / kWRows) * kWRows;
        return (int)(r < 4 * kWRows ? 4 * kWRows : (r > (1 << 22) ? (1 << 22) : r));
    };
    auto blocks_for = [&](double T) {
        long b = 0;
        for (int i = 0; i < n; ++i) {
            const bd_wgrad_desc& d = descs[i];
            if (thin(d)) {
                const int imgs = d.M / (d.g_gh * d.g_gw);
                b += cdiv(imgs, cdiv(imgs, 256));
                continue;
            }
            int t, c;
            desc_geo(d, &t, &c);
            b += (long)t * cdiv(d.M > 0 ? d.M : 1, rows_for(d, T));
        }
        return b;
    };
    double budget = 0.0;
    if (wgrad_wide() && any_gather) {
        double W = 0.0;
        for (int i = 0; i < n; ++i) {
            int t, c;
            desc_geo(descs[i], &t, &c);
            W += (double)t * (descs[i].M > 0 ? descs[i].M : 1) * c;
        }
        double R = W / (256.0 * 1024.0 * 61.0);
        if (R < 1.0) R = 1.0;
        if (R > 24.0) R = 24.0;
        const int rounds = (int)(R + 0.999);
        budget = W / (256.0 * (double)rounds);
        // The grid must be WHOLE rounds of the chip: rows_per is rounded per GEMM, and a launch of 258 workgroups for
        // "one round" ran two -- the second for two workgroups (conv 64 -> 128 alone: 0.51 ms for 0.25 ms of work).  Grow the
        // budget until the launch fits; `rounds` may grow by one when thin-image layers bring a round of their own.
        long target = 256L * rounds;
        if (blocks_for(budget * 8.0) > target) target += 256;
        for (int it = 0; it < 200 && blocks_for(budget) > target; ++it) budget *= 1.02;
    }
    for (int i = 0; i < n; ++i) {
        bd_wgrad_desc& d = descs[i];
        BD_REQUIRE(d.dpre && d.act1 && d.dW && d.M > 0 && d.N > 0 && d.K > 0 && d.M1 >= 0 && d.M1 <= d.M,
                   "bd_wgrad_plan: descriptor %d is malformed", i);
        BD_REQUIRE(d.M1 == d.M || d.act2, "bd_wgrad_plan: descriptor %d needs a second activation source", i);
        BD_REQUIRE(d.ldp >= d.N && (d.g_nseg > 0 || d.lda1 >= d.K) && d.ldw >= d.K && (d.M1 == d.M || d.lda2 >= d.K),
                   "bd_wgrad_plan: descriptor %d has a leading dimension that is too small", i);
        if (d.g_nseg > 0)
            BD_REQUIRE(wgrad_wide() && d.M1 == d.M && d.g_seglen > 0 && d.g_nseg * d.g_seglen == d.K && d.g_gh > 0 &&
                           d.g_gw > 0 && d.M % (d.g_gh * d.g_gw) == 0 && d.g_C > 0 && 2 * (d.g_gh - 1) + d.g_nseg <= d.g_IH &&
                           (2 * (d.g_gw - 1)) * d.g_C + d.g_seglen <= d.g_IW * d.g_C,
                       "bd_wgrad_plan: descriptor %d has an inconsistent gather geometry", i);
        const int hb = d.db != nullptr;
        d.g_pad = 0;
        if (wgrad_wide() && wgrad_thin_ok(d) && thin_on()) {
            // thin-image conv layer: one (whole-output) tile, whole images per workgroup, ~one round of 256 workgroups
            const int px = d.g_gh * d.g_gw, imgs = d.M / px;
            const int ipw = cdiv(imgs, 256);
            d.g_pad = 1;
            d.tiles_n = d.tiles_k = 1;
            d.rows_per = ipw * px;
            d.splits = cdiv(imgs, ipw);
        } else if (wgrad_wide()) {
            // tiles of <= 13 x 13 16-blocks, balanced; the row split is chosen below for the whole launch
            d.tiles_n = cdiv(cdiv(d.N, 16), kWB);
            d.tiles_k = wgrad_tiles_k(cdiv(d.N, 16), cdiv(d.K + hb, 16), wgrad_act16(d));
            d.rows_per = rows_wide;
            if (budget > 0.0) d.rows_per = rows_for(d, budget);
            d.splits = cdiv(d.M, d.rows_per);
        } else {
            d.tiles_n = cdiv(d.N, kWT);
            d.tiles_k = cdiv(d.K + hb, kWT);
            int s = d.M / 512;                      // ~512 rows (16 LDS stages) per workgroup
            if (s < 1) s = 1;
            if (s > 64) s = 64;
            d.rows_per = cdiv(cdiv(d.M, s), kWM) * kWM;
            d.splits = cdiv(d.M, d.rows_per);
        }
        d.block_begin = blocks;
        d.red_begin = red;
        d.ws_off = off;
        blocks += d.tiles_n * d.tiles_k * d.splits;
        red += cdiv(d.N * (d.K + hb), 256);
        off += (size_t)d.splits * d.N * (d.K + hb);
    }
    *total_blocks = blocks;
    *total_red_blocks = red;
    *ws_floats = off;
    return 0;
}

int bd_wgrad_grouped(const bd_wgrad_desc* descs_dev, int n, int total_blocks, int total_red_blocks, float* ws,
                     void* stream) {
    return bd_wgrad_grouped_phase(descs_dev, n, total_blocks, total_red_blocks, ws, 0, stream);
}

int bd_wgrad_grouped_phase(const bd_wgrad_desc* descs_dev, int n, int total_blocks, int total_red_blocks, float* ws,
                           int phase, void* stream) {
    using namespace bd;
    BD_REQUIRE(descs_dev && ws && n > 0 && n <= 4096 && total_blocks > 0 && total_red_blocks > 0 && phase >= 0 && phase <= 2,
               "bd_wgrad_grouped: bad arguments");
    if (phase == 2) {
    } else if (wgrad_wide()) {
        static_assert(kThreads == 256, "the 64x64-tile wgrad kernels are written for four waves");
        static bool lds_ok = false;
        if (!lds_ok) {
            if (allow_big_lds(wgrad_wide_kernel)) return -1;
            lds_ok = true;
        }
        hipLaunchKernelGGL(wgrad_wide_kernel, dim3(total_blocks), dim3(kWThreads), kWideLdsBytes, (hipStream_t)stream,
                           descs_dev, n, ws);
    } else {
        hipLaunchKernelGGL(wgrad_grouped_kernel, dim3(total_blocks), dim3(kThreads), 0, (hipStream_t)stream, descs_dev, n, ws);
    }
    BD_CHECK_LAUNCH("bd_wgrad_grouped");
    if (phase == 1) return 0;
    hipLaunchKernelGGL(wgrad_grouped_reduce_kernel, dim3(total_red_blocks), dim3(256), 0, (hipStream_t)stream, descs_dev, n,
                       ws);
    BD_CHECK_LAUNCH("bd_wgrad_grouped(reduce)");
    return 0;
}

size_t bd_wgrad_ws_floats(int M, int N, int K) {
    // the split count depends on whether the bias column is appended: cover both forms
    int s0, s1, rp;
    bd::wgrad_plan(M, N, K, 0, &s0, &rp);
    bd::wgrad_plan(M, N, K, 1, &s1, &rp);
    const size_t a = (size_t)s0 * N * K, b = (size_t)s1 * N * (K + 1);
    return a > b ? a : b;
}

int bd_wgrad(const float* dpre, int ldp, const float* act, int lda, int M, int N, int K, float* dW, int ldw,
             float* db, int accumulate, float* ws, size_t ws_floats, void* stream) {
    using namespace bd;
    BD_REQUIRE(dpre && act && dW && ws && M > 0 && N > 0 && K > 0, "bd_wgrad: bad arguments");
    BD_REQUIRE(ldp >= N && lda >= K && ldw >= K, "bd_wgrad: leading dimension too small");
    const int hb = db != nullptr;
    int splits, rows_per;
    wgrad_plan(M, N, K, hb, &splits, &rows_per);
    BD_REQUIRE((size_t)splits * N * (K + hb) <= ws_floats, "bd_wgrad: workspace too small (%zu floats, need %zu)",
               ws_floats, (size_t)splits * N * (K + hb));
    hipLaunchKernelGGL(wgrad_kernel, dim3(cdiv(N, kWT), cdiv(K + hb, kWT), splits), dim3(kThreads), 0,
                       (hipStream_t)stream, dpre, ldp, act, lda, M, N, K, hb, rows_per, ws);
    BD_CHECK_LAUNCH("bd_wgrad");
    const int total = N * (K + hb);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv(total, 256) < 1024 ? cdiv(total, 256) : 1024), dim3(256), 0,
                       (hipStream_t)stream, ws, splits, N, K, hb, dW, ldw, db, accumulate);
    BD_CHECK_LAUNCH("bd_wgrad(reduce)");
    return 0;
}

}  // extern "C"
