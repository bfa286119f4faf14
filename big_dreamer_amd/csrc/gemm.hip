// gemm.hip -- plain fp32 GEMM  C[M x N] (+)= A[M x K] B[N x K]^T  on v_mfma_f32_16x16x4_f32 (exact fp32 products and sums).
//
// The one plain GEMM on the pixel path whose K does not fit the LDS-resident row tiles of the gather-GEMM kernels: the
// dgrad of ConvTranspose2d(E -> 128, k5, s2) applied to a 1 x 1 map (src/models.py:338-341), i.e.
// d l0[M x E] = g[M x 3200] W[E x 3200]^T with M = (L-1) * B = 2450.  (Round 2 sent it to rocBLAS through torch.mm.)
//
// Tiling for a chip of 256 CUs and a SMALL problem (2450 x 1024 outputs): a workgroup (4 waves) owns 16*RTM rows x 64
// columns, RTM picked by the host so that the grid is as close to a whole number of rounds of 256 workgroups as it gets
// (2450 x 1024 with RTM = 10: 16 x 16 = 256 workgroups, one round).  Wave w owns column block w of the tile and all RTM row
// tiles: per 16-deep K block it reads RTM + 1 fragments (ds_read_b128) for 4 * RTM MFMAs.  K advances 32 per step through a
// double-buffered LDS image in MFMA fragment order (bd_device.h); the next step's global loads are in flight while the
// MFMAs of the current one issue; one workgroup barrier per step.
// blockIdx -> tile: blocks b and b + 8 share an XCD (round-robin dispatch), so each XCD is given a CONTIGUOUS range of
// (row tile, column tile) pairs with the column tile fastest: the A rows of a tile are re-read from that XCD's L2 by the
// workgroups of the same row tile instead of from every XCD.
#include "bd_device.h"
#include "bd_host.h"

namespace bd {

constexpr int kGemmThreads = 256;
constexpr int kGemmKS = 32;                   // K per step (two fragment blocks)
constexpr int kGemmTN = 64;                   // columns per workgroup (one 16-column block per wave)

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
    constexpr int TM = 16 * RTM;
    constexpr int kAFloats = RTM * 2 * kFragFloats, kBFloats = 4 * 2 * kFragFloats;
    __shared__ __attribute__((aligned(16))) float lds[2 * (kAFloats + kBFloats)];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-contiguous tile order (see the file comment); grids that are not a multiple of 8 fall back to the plain order
    const int total = ntm * ntn;
    int logical = blockIdx.x;
    if ((total & 7) == 0) logical = (blockIdx.x & 7) * (total >> 3) + (blockIdx.x >> 3);
    const int tm = logical / ntn, tn = logical - tm * ntn;
    const int row0 = tm * TM, col0 = tn * kGemmTN;

    // staging: float4 q of a [rows x 32] slab -> row = q / 8, k = 4 * (q % 8)
    constexpr int kAq = TM * 8, kBq = kGemmTN * 8;
    constexpr int kAper = (kAq + kGemmThreads - 1) / kGemmThreads, kBper = kBq / kGemmThreads;
    floatx4 ra[kAper], rb[kBper];

    auto load_step = [&](int k0) {
#pragma unroll
        for (int i = 0; i < kAper; ++i) {
            const int q = tid + i * kGemmThreads;
            const int r = q >> 3, k = k0 + 4 * (q & 7);
            ra[i] = floatx4{0.f, 0.f, 0.f, 0.f};
            if (q < kAq && row0 + r < M && k < K) ra[i] = ld4(A + (size_t)(row0 + r) * lda + k, K - k, vec & 1);
        }
#pragma unroll
        for (int i = 0; i < kBper; ++i) {
            const int q = tid + i * kGemmThreads;
            const int r = q >> 3, k = k0 + 4 * (q & 7);
            rb[i] = floatx4{0.f, 0.f, 0.f, 0.f};
            if (col0 + r < N && k < K) rb[i] = ld4(B + (size_t)(col0 + r) * ldb + k, K - k, vec & 2);
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
#pragma unroll
        for (int i = 0; i < kBper; ++i) {
            const int q = tid + i * kGemmThreads;
            const int r = q >> 3, k = 4 * (q & 7);
            *reinterpret_cast<floatx4*>(Bf + ((r >> 4) * 2 + (k >> 4)) * kFragFloats + frag_idx(r & 15, k & 15)) = rb[i];
        }
    };

    floatx4 acc[RTM];
#pragma unroll
    for (int rt = 0; rt < RTM; ++rt) acc[rt] = floatx4{0.f, 0.f, 0.f, 0.f};

    const int nsteps = cdiv(K, kGemmKS);
    load_step(0);
    store_step(0);
    lds_barrier();
    for (int s = 0; s < nsteps; ++s) {
        if (s + 1 < nsteps) load_step((s + 1) * kGemmKS);
        const float* __restrict__ Af = lds + (s & 1) * (kAFloats + kBFloats);
        const floatx4* __restrict__ A4 = reinterpret_cast<const floatx4*>(Af) + lane;
        const floatx4* __restrict__ B4 = reinterpret_cast<const floatx4*>(Af + kAFloats) + lane;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const floatx4 b4 = B4[(wave * 2 + kb) * 64];
            floatx4 a4[RTM];
#pragma unroll
            for (int rt = 0; rt < RTM; ++rt) a4[rt] = A4[(rt * 2 + kb) * 64];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int rt = 0; rt < RTM; ++rt) acc[rt] = mfma16(a4[rt][j], b4[j], acc[rt]);
        }
        if (s + 1 < nsteps) store_step((s + 1) & 1);
        lds_barrier();
    }
    const int col = col0 + wave * 16 + (lane & 15);
    if (col < N) {
#pragma unroll
        for (int rt = 0; rt < RTM; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + rt * 16 + 4 * (lane >> 4) + r;
                if (row < M) {
                    float* p = C + (size_t)row * ldc + col;
                    *p = accumulate ? *p + acc[rt][r] : acc[rt][r];
                }
            }
    }
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
    switch (best) {
        case 10: return launch_gemm<10>(A, lda, B, ldb, C, ldc, M, N, K, accumulate, vec, s);
        case 8: return launch_gemm<8>(A, lda, B, ldb, C, ldc, M, N, K, accumulate, vec, s);
        case 6: return launch_gemm<6>(A, lda, B, ldb, C, ldc, M, N, K, accumulate, vec, s);
        default: return launch_gemm<4>(A, lda, B, ldb, C, ldc, M, N, K, accumulate, vec, s);
    }
}

}  // extern "C"
