// wgrad_onehot.hip -- weight gradient of the ONE-HOT columns of a first layer (Categorical latents, configs[4]).
//
// The heads on imagined features take [h; one-hot s] (src/dreamer.py:320-322,370-391 through DenseModel, src/models.py:365-408),
// s = D factors x C classes as a dense S = D*C wide 0/1 vector with exactly one 1 per factor.  Autograd forms
//     dW[n][Be + f*C + c] = sum_m dpre[m][n] * s[m][f*C + c]
// as a dense K = S contraction (2 * M * N * S flops: 28 GFLOP for the critic at M = 68 600, N = 200, S = 1024 -- five times
// the rest of that pass, and the wide weight-gradient kernel streams the 4 KB one-hot row of every transition through LDS to
// multiply by zeros).  With the class indices the sampler already wrote (sidx [M x D] bytes) it is a segmented sum:
//     dW[n][Be + f*C + c] = sum over the rows m with sidx[m][f] == c of dpre[m][n]         (M * D * N adds: 32x fewer operations).
//
// Kernel 1: workgroup = (row slab, group of 8 factors, block of 64 columns n); WAVE w owns factor 8*fg + w and a private
// C x 64 table in LDS; lane = column.  Per row: the class index is wave-uniform (read once per 64 rows, one byte per lane, and
// broadcast by v_readlane), the row's 64 values are one coalesced 256-byte load (all eight waves read the same bytes: L1), and
// the update is ONE fire-and-forget `ds_add_f32` at tab[c][lane] -- no two lanes share an address, a wave's LDS operations
// execute in issue order, so every cell sums its rows in row order: bitwise reproducible.  Kernel 2 sums the slabs' tables in
// slab order and writes the transposed result into the [N][ldw] weight-gradient matrix.
#include "bd_device.h"
#include "bd_host.h"
#include <stdlib.h>

namespace bd {

constexpr int kOhThreads = 512;          // 8 waves = 8 factors per workgroup
constexpr int kOhWaves = 8;
constexpr int kOhNB = 64;                // columns per workgroup (one per lane)
constexpr int kOhMaxC = 64;              // table = 8 waves x C x 64 floats <= 128 KB
constexpr int kOhMaxSlabs = 16;

// rows [0, M1) take their indices from sidx1, rows [M1, M) from sidx2[row - M1] (the actor's first layer: start states, then
// imagined states -- the two-source rule of bd_wgrad_desc)
__global__ __launch_bounds__(kOhThreads) void wgrad_onehot_kernel(const float* __restrict__ dpre, int ldp,
                                                                  const unsigned char* __restrict__ sidx1, int M1,
                                                                  const unsigned char* __restrict__ sidx2, int M, int N, int D, int C,
                                                                  int rows_per_slab, int nfg, float* __restrict__ ws, int Npad) {
    extern __shared__ __attribute__((aligned(16))) float tab_all[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int b = blockIdx.x;
    const int nblk = b % (Npad / kOhNB);
    b /= (Npad / kOhNB);
    const int fg = b % nfg, slab = b / nfg;
    const int f = fg * kOhWaves + wave;
    const int n = nblk * kOhNB + lane;
    float* tab = tab_all + (size_t)wave * C * kOhNB;
    for (int i = lane; i < C * kOhNB; i += 64) tab[i] = 0.f;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (a wave's own LDS operations retire in order anyway)
    const int m0 = slab * rows_per_slab, m1 = min(M, m0 + rows_per_slab);
    if (f < D) {
        const bool nok = n < N;
        for (int mb = m0; mb < m1; mb += 64) {
            // class index of row mb + lane (one byte per lane), broadcast row by row below
            const int mr = mb + lane;
            int cidx = 0;
            if (mr < m1) cidx = mr < M1 ? (int)sidx1[(size_t)mr * D + f] : (int)sidx2[(size_t)(mr - M1) * D + f];
            const int cnt = min(64, m1 - mb);
            const float* __restrict__ src = dpre + (size_t)mb * ldp + n;
#pragma unroll 1
            for (int r0 = 0; r0 < cnt; r0 += 16) {
                float v[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) v[j] = (nok && r0 + j < cnt) ? src[(size_t)(r0 + j) * ldp] : 0.f;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    if (r0 + j < cnt) {
                        const int c = __builtin_amdgcn_readlane(cidx, r0 + j);      // uniform: r0 + j is wave-uniform
                        __hip_atomic_fetch_add(tab + c * kOhNB + lane, v[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // this slab's table of factor f: ws[slab][k = f*C + c][Npad]
        float* __restrict__ out = ws + ((size_t)slab * D * C + (size_t)f * C) * Npad + nblk * kOhNB + lane;
        for (int c = 0; c < C; ++c) out[(size_t)c * Npad] = tab[c * kOhNB + lane];
    }
}

// dW[n][k] = sum over the slabs of ws[slab][k][n], slab order fixed; a 32 x 32 tile through LDS so that both sides coalesce
__global__ __launch_bounds__(256) void wgrad_onehot_reduce_kernel(const float* __restrict__ ws, int slabs, int S, int N, int Npad,
                                                                  float* __restrict__ dW, int ldw) {
    __shared__ float tile[32][33];
    const int kt = blockIdx.x * 32, nt = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
    for (int j = ty; j < 32; j += 8) {
        const int k = kt + j, n = nt + tx;
        float acc = 0.f;
        if (k < S && n < N)
            for (int s = 0; s < slabs; ++s) acc += ws[((size_t)s * S + k) * Npad + n];
        tile[j][tx] = acc;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int n = nt + j, k = kt + tx;
        if (n < N && k < S) dW[(size_t)n * ldw + k] = tile[tx][j];
    }
}

static int onehot_slabs(int M) {
    int s = M / 1024;                    // at least ~1k rows per slab: the table write-out and the reduce are per slab
    if (s < 1) s = 1;
    return s > kOhMaxSlabs ? kOhMaxSlabs : s;
}

}  // namespace bd

extern "C" {
using namespace bd;

int bd_wgrad_onehot_ok(int D, int C) { return D > 0 && C > 0 && C <= kOhMaxC; }

size_t bd_wgrad_onehot_ws_floats(int M, int N, int D, int C) {
    const int Npad = cdiv(N, kOhNB) * kOhNB;
    return (size_t)onehot_slabs(M) * D * C * Npad;
}

int bd_wgrad_onehot(const float* dpre, int ldp, const unsigned char* sidx1, int M1, const unsigned char* sidx2, int M, int N, int D,
                    int C, float* dW, int ldw, float* ws, size_t ws_floats, void* stream) {
    BD_REQUIRE(dpre && sidx1 && dW && ws && M > 0 && N > 0 && ldp >= N && M1 >= 0 && M1 <= M && (M1 == M || sidx2),
               "bd_wgrad_onehot: bad arguments");
    BD_REQUIRE(bd_wgrad_onehot_ok(D, C), "bd_wgrad_onehot: %d x %d latents unsupported (C <= %d)", D, C, kOhMaxC);
    BD_REQUIRE(ldw >= D * C, "bd_wgrad_onehot: ldw = %d below the %d one-hot columns", ldw, D * C);
    BD_REQUIRE(ws_floats >= bd_wgrad_onehot_ws_floats(M, N, D, C), "bd_wgrad_onehot: workspace too small");
    const int Npad = cdiv(N, kOhNB) * kOhNB, nfg = cdiv(D, kOhWaves), slabs = onehot_slabs(M);
    const int rows_per_slab = cdiv(cdiv(M, slabs), 64) * 64;
    const size_t lds = (size_t)kOhWaves * C * kOhNB * sizeof(float);
    if (lds > 64 * 1024 && allow_big_lds(wgrad_onehot_kernel)) return -1;
    hipStream_t s = (hipStream_t)stream;
    // (slabs beyond the rows write zero tables: rows_per_slab is rounded up, the last slabs may be empty)
    hipLaunchKernelGGL(wgrad_onehot_kernel, dim3(slabs * nfg * (Npad / kOhNB)), dim3(kOhThreads), lds, s, dpre, ldp, sidx1, M1, sidx2, M,
                       N, D, C, rows_per_slab, nfg, ws, Npad);
    BD_CHECK_LAUNCH("bd_wgrad_onehot");
    hipLaunchKernelGGL(wgrad_onehot_reduce_kernel, dim3(cdiv(D * C, 32), cdiv(N, 32)), dim3(256), 0, s, ws, slabs, D * C, N, Npad, dW,
                       ldw);
    BD_CHECK_LAUNCH("bd_wgrad_onehot (reduce)");
    return 0;
}

}  // extern "C"
