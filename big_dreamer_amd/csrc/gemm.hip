// gemm.hip -- plain fp32 GEMM  C[M x N] (+)= A[M x K] B[N x K]^T  on v_mfma_f32_16x16x4_f32 (exact fp32 products and sums).
//
// The one plain GEMM on the pixel path whose K does not fit the LDS-resident row tiles of the gather-GEMM kernels: the
// dgrad of ConvTranspose2d(E -> 128, k5, s2) applied to a 1 x 1 map (src/models.py:338-341), i.e.
// d l0[M x E] = g[M x 3200] W[E x 3200]^T with M = (L-1) * B = 2450.  (Round 2 sent it to rocBLAS through torch.mm.)
//
// Tiling for a chip of 256 CUs and a SMALL problem (2450 x 1024 outputs): a workgroup (4 waves) owns 16*RTM rows x 64
// columns, RTM picked by the host so that the grid is as close to a whole number of rounds of 256 workgroups as it gets
// (2450 x 1024 with RTM = 10: 16 x 16 = 256 workgroups, one round).  Eight waves as 2 row halves x 4 column blocks: a wave
// owns RTM / 2 row tiles x one column block -- per 16-deep K block RTM / 2 + 1 fragment reads (ds_read_b128) for 2 * RTM
// MFMAs -- and the two waves of a SIMD cover each other's LDS waits.  K advances 32 per step through a
// double-buffered LDS image in MFMA fragment order (bd_device.h); the next step's global loads are in flight while the
// MFMAs of the current one issue; one workgroup barrier per step.
// blockIdx -> tile: blocks b and b + 8 share an XCD (round-robin dispatch), so each XCD is given a CONTIGUOUS range of
// (row tile, column tile) pairs with the column tile fastest: the A rows of a tile are re-read from that XCD's L2 by the
// workgroups of the same row tile instead of from every XCD.
#include "bd_device.h"
#include "bd_host.h"
#include <stdlib.h>

namespace bd {

constexpr int kGemmThreads = 512;             // 8 waves: 2 (row halves) x 4 (column blocks); two waves per SIMD
constexpr int kGemmKS = 32;                   // K per step (two fragment blocks)
constexpr int kGemmTN = 64;                   // columns per workgroup (one 16-column block per wave column)

// four consecutive floats of a row: one 16-byte load when the operand allows it (`vec`), else up to `left` scalar loads
// (a weight matrix sits at an arbitrary float offset of the flat parameter buffer)
__device__ __forceinline__ floatx4 ld4(const float* __restrict__ p, int left, int vec) {
    if (vec) return *reinterpret_cast<const floatx4*>(p);
    floatx4 v = floatx4{0.f, 0.f, 0.f, 0.f};
    v[0] = p[0];
    if (left > 1) v[1] = p[1];
    if (left > 2) v[2] = p[2];
    if (left > 3) v[3] = p[3];
    return v;
}

template <int RTM>
__global__ __launch_bounds__(kGemmThreads) void gemm_nt_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B,
                                                               int ldb, float* __restrict__ C, int ldc, int M, int N, int K,
                                                               int accumulate, int ntm, int ntn, int vec) {
    static_assert(RTM % 2 == 0, "row tiles split over two wave rows");
    constexpr int TM = 16 * RTM, RTW = RTM / 2;
    constexpr int kAFloats = RTM * 2 * kFragFloats, kBFloats = 4 * 2 * kFragFloats;
    __shared__ __attribute__((aligned(16))) float lds[2 * (kAFloats + kBFloats)];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave & 3, wm = wave >> 2;      // waves w and w + 4 share a SIMD: same column block, different row halves
    // XCD-contiguous tile order (see the file comment); grids that are not a multiple of 8 fall back to the plain order
    const int total = ntm * ntn;
    int logical = blockIdx.x;
    if ((total & 7) == 0) logical = (blockIdx.x & 7) * (total >> 3) + (blockIdx.x >> 3);
    const int tm = logical / ntn, tn = logical - tm * ntn;
    const int row0 = tm * TM, col0 = tn * kGemmTN;

    // staging: float4 q of a [rows x 32] slab -> row = q / 8, k = 4 * (q % 8)
    constexpr int kAq = TM * 8, kBq = kGemmTN * 8;
    constexpr int kAper = (kAq + kGemmThreads - 1) / kGemmThreads;
    static_assert(kBq == kGemmThreads, "one B float4 per thread and step");
    floatx4 ra[kAper], rb;

    auto load_step = [&](int k0) {
#pragma unroll
        for (int i = 0; i < kAper; ++i) {
            const int q = tid + i * kGemmThreads;
            const int r = q >> 3, k = k0 + 4 * (q & 7);
            ra[i] = floatx4{0.f, 0.f, 0.f, 0.f};
            if (q < kAq && row0 + r < M && k < K) ra[i] = ld4(A + (size_t)(row0 + r) * lda + k, K - k, vec & 1);
        }
        {
            const int r = tid >> 3, k = k0 + 4 * (tid & 7);
            rb = floatx4{0.f, 0.f, 0.f, 0.f};
            if (col0 + r < N && k < K) rb = ld4(B + (size_t)(col0 + r) * ldb + k, K - k, vec & 2);
        }
    };
    auto store_step = [&](int buf) {
        float* __restrict__ Af = lds + buf * (kAFloats + kBFloats);
        float* __restrict__ Bf = Af + kAFloats;
#pragma unroll
        for (int i = 0; i < kAper; ++i) {
            const int q = tid + i * kGemmThreads;
            const int r = q >> 3, k = 4 * (q & 7);
            if (q < kAq) *reinterpret_cast<floatx4*>(Af + ((r >> 4) * 2 + (k >> 4)) * kFragFloats + frag_idx(r & 15, k & 15)) = ra[i];
        }
        {
            const int r = tid >> 3, k = 4 * (tid & 7);
            *reinterpret_cast<floatx4*>(Bf + ((r >> 4) * 2 + (k >> 4)) * kFragFloats + frag_idx(r & 15, k & 15)) = rb;
        }
    };

    floatx4 acc[RTW];
#pragma unroll
    for (int rt = 0; rt < RTW; ++rt) acc[rt] = floatx4{0.f, 0.f, 0.f, 0.f};

    const int nsteps = cdiv(K, kGemmKS);
    load_step(0);
    store_step(0);
    lds_barrier();
    for (int s = 0; s < nsteps; ++s) {
        if (s + 1 < nsteps) load_step((s + 1) * kGemmKS);
        const float* __restrict__ Af = lds + (s & 1) * (kAFloats + kBFloats);
        const floatx4* __restrict__ A4 = reinterpret_cast<const floatx4*>(Af) + (wm * RTW * 2) * 64 + lane;
        const floatx4* __restrict__ B4 = reinterpret_cast<const floatx4*>(Af + kAFloats) + (wn * 2) * 64 + lane;
        // both K blocks' fragments are requested up front: the second block's reads land under the first block's MFMAs
        floatx4 b4[2], a4[2][RTW];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            b4[kb] = B4[kb * 64];
#pragma unroll
            for (int rt = 0; rt < RTW; ++rt) a4[kb][rt] = A4[(rt * 2 + kb) * 64];
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int rt = 0; rt < RTW; ++rt) acc[rt] = mfma16(a4[kb][rt][j], b4[kb][j], acc[rt]);
        if (s + 1 < nsteps) store_step((s + 1) & 1);
        lds_barrier();
    }
    const int col = col0 + wn * 16 + (lane & 15);
    if (col < N) {
#pragma unroll
        for (int rt = 0; rt < RTW; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + (wm * RTW + rt) * 16 + 4 * (lane >> 4) + r;
                if (row < M) {
                    float* p = C + (size_t)row * ldc + col;
                    *p = accumulate ? *p + acc[rt][r] : acc[rt][r];
                }
            }
    }
}

// ---- the aligned form: operands staged by LDS-DMA ------------------------------------------------------------------------
// K % 16 == 0, lda / ldb multiples of 4 floats, A / B 16-byte aligned (every weight of the flat parameter buffers is:
// engine.ParamGroup).  One `global_load_lds_dwordx4` wave-instruction moves one whole 16 x 16 fragment block: lane l
// fetches the 16 bytes (row = l & 15, k = 4 * (l >> 4) .. + 3) and the DMA lands them at lane * 16 of a wave-uniform LDS
// base -- exactly the fragment order the MFMA reads expect, with no staging registers, no ds_write (the register-staged
// form above pays 8-way bank conflicts on them) and a ring of three stages: the DMAs of step s + 2 are issued at the top of
// step s and the end of step s waits for the wave's OWN DMAs of step s + 1 only (`vmcnt(n)`, in order), then the barrier.
// Rows beyond M / columns beyond N re-read the last valid row: their products are never stored.
constexpr int kGemmRing = 3;

template <int RTM>
__global__ __launch_bounds__(kGemmThreads) void gemm_nt_dma_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B,
                                                                   int ldb, float* __restrict__ C, int ldc, int M, int N, int K,
                                                                   int accumulate, int ntm, int ntn) {
    static_assert(RTM % 2 == 0, "row tiles split over two wave rows");
    constexpr int TM = 16 * RTM, RTW = RTM / 2;
    constexpr int kBlocks = 2 * RTM + 8;                       // fragment blocks per stage: A [RTM][2] then B [4][2]
    constexpr int kStage = kBlocks * kFragFloats;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    typedef const __attribute__((address_space(1))) void* glb_ptr_t;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 3, wm = wave >> 2;
    const int total = ntm * ntn;
    int logical = blockIdx.x;
    if ((total & 7) == 0) logical = (blockIdx.x & 7) * (total >> 3) + (blockIdx.x >> 3);
    const int tm = logical / ntn, tn = logical - tm * ntn;
    const int row0 = tm * TM, col0 = tn * kGemmTN;

    // this wave's blocks of a stage: b = wave, wave + 8, ... < kBlocks; per block a per-lane source pointer at k = 0
    constexpr int kPer = (kBlocks + 7) / 8;
    const float* src[kPer];
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
        const int b = wave + 8 * i;
        src[i] = A;
        if (b < kBlocks) {
            const int kb = b & 1, t = b >> 1;                  // t < RTM: A row tile t; else B column block t - RTM
            if (t < RTM) {
                int r = row0 + t * 16 + (lane & 15);
                r = r < M ? r : M - 1;
                src[i] = A + (size_t)r * lda + kb * 16 + 4 * (lane >> 4);
            } else {
                int r = col0 + (t - RTM) * 16 + (lane & 15);
                r = r < N ? r : N - 1;
                src[i] = B + (size_t)r * ldb + kb * 16 + 4 * (lane >> 4);
            }
        }
    }
    const int nsteps = cdiv(K, kGemmKS);
    auto issue = [&](int s) -> int {                           // returns the number of DMAs this wave issued
        float* buf = lds + (s % kGemmRing) * kStage;
        const int k0 = s * kGemmKS;
        const bool half = k0 + 16 >= K;                        // last step of a K that is 16 mod 32: only the kb = 0 blocks exist
        int n = 0;
#pragma unroll
        for (int i = 0; i < kPer; ++i) {
            const int b = wave + 8 * i;
            if (b < kBlocks && !(half && (b & 1))) {
                __builtin_amdgcn_global_load_lds((glb_ptr_t)(src[i] + k0), (lds_ptr_t)(buf + b * kFragFloats), 16, 0, 0);
                ++n;
            }
        }
        return n;
    };
    auto wait_older = [&](int keep) {                          // all but the `keep` newest DMAs of this wave have landed
        switch (keep) {
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        }
    };
    static_assert(kPer <= 4, "wait_older covers up to four DMAs per wave and stage");

    floatx4 acc[RTW];
#pragma unroll
    for (int rt = 0; rt < RTW; ++rt) acc[rt] = floatx4{0.f, 0.f, 0.f, 0.f};

    issue(0);
    if (nsteps > 1) issue(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // prologue: both stages have landed
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
        const int n_new = s + 2 < nsteps ? issue(s + 2) : 0;
        const float* __restrict__ buf = lds + (s % kGemmRing) * kStage;
        const floatx4* __restrict__ A4 = reinterpret_cast<const floatx4*>(buf) + (wm * RTW * 2) * 64 + lane;
        const floatx4* __restrict__ B4 = reinterpret_cast<const floatx4*>(buf) + ((RTM + wn) * 2) * 64 + lane;
        const int nkb = (s * kGemmKS + 16 >= K) ? 1 : 2;
        floatx4 b4[2], a4[2][RTW];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            b4[kb] = B4[kb * 64];
#pragma unroll
            for (int rt = 0; rt < RTW; ++rt) a4[kb][rt] = A4[(rt * 2 + kb) * 64];
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
            if (kb < nkb) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int rt = 0; rt < RTW; ++rt) acc[rt] = mfma16(a4[kb][rt][j], b4[kb][j], acc[rt]);
            }
        // stage s + 1 must have landed for EVERY wave before anyone reads it: the wave's own DMAs first (all but the n_new
        // just issued for stage s + 2, which stay in flight; vmcnt retires in order), then the barrier
        wait_older(n_new);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    const int col = col0 + wn * 16 + (lane & 15);
    if (col < N) {
#pragma unroll
        for (int rt = 0; rt < RTW; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + (wm * RTW + rt) * 16 + 4 * (lane >> 4) + r;
                if (row < M) {
                    float* p = C + (size_t)row * ldc + col;
                    *p = accumulate ? *p + acc[rt][r] : acc[rt][r];
                }
            }
    }
}

template <int RTM>
static int launch_gemm_dma(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K, int accumulate,
                           hipStream_t s) {
    const int ntm = cdiv(M, 16 * RTM), ntn = cdiv(N, kGemmTN);
    const size_t lds = (size_t)kGemmRing * (2 * RTM + 8) * kFragFloats * sizeof(float);
    if (lds > 64 * 1024 && allow_big_lds(gemm_nt_dma_kernel<RTM>)) return -1;
    hipLaunchKernelGGL(gemm_nt_dma_kernel<RTM>, dim3(ntm * ntn), dim3(kGemmThreads), lds, s, A, lda, B, ldb, C, ldc, M, N, K,
                       accumulate, ntm, ntn);
    BD_CHECK_LAUNCH("bd_gemm_nt");
    return 0;
}

template <int RTM>
static int launch_gemm(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K, int accumulate,
                       int vec, hipStream_t s) {
    const int ntm = cdiv(M, 16 * RTM), ntn = cdiv(N, kGemmTN);
    hipLaunchKernelGGL(gemm_nt_kernel<RTM>, dim3(ntm * ntn), dim3(kGemmThreads), 0, s, A, lda, B, ldb, C, ldc, M, N, K, accumulate,
                       ntm, ntn, vec);
    BD_CHECK_LAUNCH("bd_gemm_nt");
    return 0;
}

}  // namespace bd

extern "C" {
using namespace bd;

int bd_gemm_nt(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K, int accumulate,
               void* stream) {
    BD_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0, "bd_gemm_nt: bad arguments");
    BD_REQUIRE(lda >= K && ldb >= K && ldc >= N, "bd_gemm_nt: leading dimensions");
    // 16-byte loads per operand where its base, leading dimension and K allow them (bit 0: A, bit 1: B)
    const int vec = (((K & 3) == 0 && (lda & 3) == 0 && ((uintptr_t)A & 15) == 0) ? 1 : 0) |
                    (((K & 3) == 0 && (ldb & 3) == 0 && ((uintptr_t)B & 15) == 0) ? 2 : 0);
    // rows per workgroup: the candidate whose grid wastes the least of its last round of 256 workgroups
    const int ntn = cdiv(N, kGemmTN);
    int best = 10;
    double best_cost = 1e30;
    for (int rtm : {10, 8, 6, 4}) {
        const int wgs = cdiv(M, 16 * rtm) * ntn;
        const double cost = (double)cdiv(wgs, 256) * rtm * (1.0 + 0.25 / rtm);     // rounds x work per workgroup (+ per-step overhead)
        if (cost < best_cost) { best_cost = cost; best = rtm; }
    }
    hipStream_t s = (hipStream_t)stream;
    static const char* force = getenv("BD_GEMM_DMA");              // "0": the register-staged form for every shape (tests)
    const bool dma = vec == 3 && (K & 15) == 0 && !(force && force[0] == '0');
    switch (best) {
        case 10: return dma ? launch_gemm_dma<10>(A, lda, B, ldb, C, ldc, M, N, K, accumulate, s)
                            : launch_gemm<10>(A, lda, B, ldb, C, ldc, M, N, K, accumulate, vec, s);
        case 8: return dma ? launch_gemm_dma<8>(A, lda, B, ldb, C, ldc, M, N, K, accumulate, s)
                           : launch_gemm<8>(A, lda, B, ldb, C, ldc, M, N, K, accumulate, vec, s);
        case 6: return dma ? launch_gemm_dma<6>(A, lda, B, ldb, C, ldc, M, N, K, accumulate, s)
                           : launch_gemm<6>(A, lda, B, ldb, C, ldc, M, N, K, accumulate, vec, s);
        default: return dma ? launch_gemm_dma<4>(A, lda, B, ldb, C, ldc, M, N, K, accumulate, s)
                            : launch_gemm<4>(A, lda, B, ldb, C, ldc, M, N, K, accumulate, vec, s);
    }
}

}  // extern "C"
