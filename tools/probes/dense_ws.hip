// dense_ws.hip -- weight-stationary dense layer for tall inputs (M >> N): forward out = act(in W^T + b); backward (dgrad)
// out = (in W^T) * ELU'(saved), i.e. the gradient w.r.t. the previous layer's pre-activation, with W the packed transpose.
//
// The chain kernels (mlp.hip) keep a 16-row tile's activations in LDS across all layers and re-stream every layer's
// weights from L2 per tile: 160 KB per 200x200 layer per 16 rows, which at ~18 B/clk/CU holds the big dense chains
// (34 300 imagined rows) at ~38 % of the fp32 MFMA peak.  Here the roles are swapped for ONE layer: a persistent
// workgroup per CU keeps the layer's weights in registers for its whole life -- wave w owns output column block w, and
// the packed weight layout ([nb][kb][lane] float4) is already the MFMA B operand, so a 200-wide contraction is 13 float4
// = 52 VGPRs per lane -- and walks its share of the row tiles, double-buffered through LDS in fragment order (a tile row
// of 200 floats is 50 float4 loads that land as 50 float4 LDS stores).  Steady state has no exposed global latency and
// no weight traffic; activations make one HBM round trip per layer.
#include "bd_device.h"
#include "bd_host.h"
#include "bd_probes.h"
#include <stdlib.h>

namespace bd {

char* err_buf() {
    static thread_local char buf[512];
    return buf;
}

constexpr int kWsWaves = 16;                 // up to 16 output column blocks (N <= 256)
constexpr int kWsThreads = kWsWaves * 64;

struct WsArgs {
    const float* in;      // [M x K] row-major, leading dimension ldi
    const float* saved;   // backward only: saved post-activation output [M x N] (ld = ldo) of the layer whose
                          // pre-activation gradient is produced: out = acc * ELU'(saved); null in the forward
    const float* w;       // packed (N, K)
    const float* bias;    // [N] or null
    float* out;           // [M x N], leading dimension ldo
    int M, N, K, ldi, ldo, act;
};

template <int KB>
__global__ __launch_bounds__(kWsThreads) void dense_ws_kernel(WsArgs a) {
    __shared__ __attribute__((aligned(16))) float buf[2][KB * kFragFloats];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Nb = (a.N + 15) >> 4;
    const int ntiles = (a.M + 15) >> 4;
    const int K4 = a.K >> 2;                                  // float4 per row (K % 4 == 0, checked by the host)
    // this thread's float4 of a tile: row r, k = 4 * c4.  Rows vary fastest across lanes: the fragment layout keeps the 16
    // rows of one (kb, k-quad) in 256 contiguous bytes, so a wave's ds_write_b128 is conflict-free (with k fastest every
    // lane of a wave lands in the same two bank groups, ~32-way conflicts); the global side still reads 64 contiguous
    // bytes per row per wave.
    const int r = tid & 15, c4 = tid >> 4;
    const bool loader = c4 < K4;
    const int lds_off = (((c4 >> 2) * 64 + (c4 & 3) * 16 + r) << 2);   // frag_idx(r, 4 * c4)

    for (int i = tid; i < 2 * KB * kFragFloats; i += kWsThreads) (&buf[0][0])[i] = 0.f;   // k >= K stays zero

    floatx4 w[KB];
    float b = 0.f;
    const bool worker = wave < Nb;
    if (worker) {
        const floatx4* __restrict__ W4 = reinterpret_cast<const floatx4*>(a.w) + (size_t)wave * KB * 64 + lane;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) w[kb] = W4[kb * 64];
        const int col = wave * 16 + (lane & 15);
        b = (a.bias != nullptr && col < a.N) ? a.bias[col] : 0.f;
    }

    auto fetch = [&](int tile) -> floatx4 {
        floatx4 v = floatx4{0.f, 0.f, 0.f, 0.f};
        const int row = tile * 16 + r;
        if (loader && row < a.M) {
            const size_t off = (size_t)row * a.ldi + 4 * c4;
            v = *reinterpret_cast<const floatx4*>(a.in + off);
        }
        return v;
    };

    const int col = wave * 16 + (lane & 15);
    // epilogue of one tile: activation (forward) or ELU' of the saved output (backward), then the store
    auto finish = [&](int t, const floatx4& acc, const float (&sv)[4]) {
        if (!worker || col >= a.N) return;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = t * 16 + 4 * (lane >> 4) + q;
            if (row < a.M)
                a.out[(size_t)row * a.ldo + col] =
                    a.saved != nullptr ? acc[q] * elu_grad_from_out(sv[q]) : act_apply(a.act, acc[q]);
        }
    };

    int tile = blockIdx.x;
    floatx4 stage = fetch(tile);
    __syncthreads();                                           // zero fill done
    if (loader) *reinterpret_cast<floatx4*>(&buf[0][lds_off]) = stage;
    __syncthreads();
    // The epilogue of tile i is deferred to the start of iteration i+1 (see the note at its call).
    floatx4 prev = floatx4{0.f, 0.f, 0.f, 0.f};
    float prev_sv[4] = {1.f, 1.f, 1.f, 1.f};
    int prev_tile = -1;
    for (int it = 0; tile < ntiles; tile += gridDim.x, ++it) {
        const int nxt = tile + gridDim.x;
        if (nxt < ntiles) stage = fetch(nxt);                  // in flight under the MFMAs below
        float sv[4] = {1.f, 1.f, 1.f, 1.f};                    // epilogue operand: fetched a whole tile ahead of its use
        if (worker && a.saved != nullptr && col < a.N) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = tile * 16 + 4 * (lane >> 4) + q;
                if (row < a.M) sv[q] = a.saved[(size_t)row * a.ldo + col];
            }
        }
        // stores of the previous tile go out BEFORE this tile's contraction: the wait in front of the LDS write below is
        // s_waitcnt vmcnt(0) (the branchy epilogue defeats the compiler's counting), and by then they have had a whole
        // MFMA phase to retire -- issued after the contraction they cost their full latency once per tile (9 of 44 us)
        if (prev_tile >= 0) finish(prev_tile, prev, prev_sv);
        floatx4 acc0 = floatx4{b, b, b, b}, acc1 = floatx4{0.f, 0.f, 0.f, 0.f};
        if (worker) {
            const floatx4* __restrict__ A4 = reinterpret_cast<const floatx4*>(buf[it & 1]) + lane;
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                const floatx4 x = A4[kb * 64];
                acc0 = mfma16(x[0], w[kb][0], acc0);
                acc1 = mfma16(x[1], w[kb][1], acc1);
                acc0 = mfma16(x[2], w[kb][2], acc0);
                acc1 = mfma16(x[3], w[kb][3], acc1);
            }
        }
        prev = acc0 + acc1;
#pragma unroll
        for (int q = 0; q < 4; ++q) prev_sv[q] = sv[q];
        prev_tile = tile;
        if (nxt < ntiles && loader) *reinterpret_cast<floatx4*>(&buf[(it + 1) & 1][lds_off]) = stage;
        lds_barrier();
    }
    if (prev_tile >= 0) finish(prev_tile, prev, prev_sv);
}

// Diagnostic: nothing but v_mfma_f32_16x16x4_f32 on register operands -- 4 waves per SIMD, four independent accumulator
// chains per wave -- to measure what the matrix pipes sustain on this part (clock under load included).
__global__ __launch_bounds__(kWsThreads) void mfma_probe_kernel(int iters, float* __restrict__ out) {
    const float a0 = (float)(threadIdx.x & 7) * 0.125f, b0 = (float)(threadIdx.x & 3) * 0.25f;
    floatx4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = floatx4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = mfma16(a0, b0, acc[i]);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) out[0] = s;        // keeps the chains alive; never true in practice
}

// Diagnostic variants of the MFMA loop: (1) operands change with every instruction (16 A registers x 16 B registers, as a
// real contraction has them) instead of one constant pair; (2) additionally the A operands come from LDS (ds_read_b128 per
// four MFMAs, prefetched one group ahead).
template <int VARIANT>
__global__ __launch_bounds__(kWsThreads) void mfma_probe2_kernel(int iters, float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float lds[16 * 256];
    for (int i = threadIdx.x; i < 16 * 256; i += blockDim.x) lds[i] = (float)(i & 15) * 0.0625f;
    __syncthreads();
    float a[16], b[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        a[i] = (float)((threadIdx.x + i) & 7) * 0.125f;
        b[i] = (float)((threadIdx.x + 3 * i) & 3) * 0.25f;
    }
    floatx4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = floatx4{0.f, 0.f, 0.f, 0.f};
    const floatx4* __restrict__ X = reinterpret_cast<const floatx4*>(lds) + (threadIdx.x & 63);
    for (int it = 0; it < iters; ++it) {
        if constexpr (VARIANT == 1) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = mfma16(a[(u * 4 + i) & 15], b[(u * 4 + i + 5) & 15], acc[i]);
            }
        } else {
            floatx4 x = X[0];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const floatx4 nx = X[((u + 1) & 15) * 64];
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = mfma16(x[i], b[(u * 4 + i) & 15], acc[i]);
                x = nx;
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) out[0] = s;
}

template <int KB>
static int launch_ws(const WsArgs& a, hipStream_t s) {
    const int ntiles = (a.M + 15) / 16;
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return fail("dense_ws: no device");
        cus = p.multiProcessorCount;
    }
    const int grid = ntiles < cus ? ntiles : cus;
    hipLaunchKernelGGL(dense_ws_kernel<KB>, dim3(grid), dim3(kWsThreads), 0, s, a);
    BD_CHECK_LAUNCH("bd_dense_ws");
    return 0;
}

}  // namespace bd

extern "C" {
using namespace bd;

int bd_dense_ws_supported(int M, int N, int K) {
    const int Kb = (K + 15) / 16;
    return M >= 16 && N >= 1 && N <= 16 * kWsWaves && K % 4 == 0 && 16 * (K / 4) <= kWsThreads && (Kb == 13);
}

int bd_dense_ws(const float* in, int ldi, const float* w_packed, const float* bias, const float* saved, int M, int N, int K,
                int act, float* out, int ldo, void* stream) {
    BD_REQUIRE(in && w_packed && out && M > 0 && N > 0 && K > 0, "bd_dense_ws: bad arguments");
    BD_REQUIRE(bd_dense_ws_supported(M, N, K), "bd_dense_ws: shape M=%d N=%d K=%d not supported", M, N, K);
    BD_REQUIRE(ldi >= K && ldi % 4 == 0 && ldo >= N, "bd_dense_ws: leading dimensions (ldi %d must be a multiple of 4)", ldi);
    BD_REQUIRE(((uintptr_t)in & 15) == 0, "bd_dense_ws: the input must be 16-byte aligned");
    const WsArgs a{in, saved, w_packed, bias, out, M, N, K, ldi, ldo, act};
    return launch_ws<13>(a, (hipStream_t)stream);
}

/* diagnostic: `iters` x 32 MFMAs per wave on 16 waves x `blocks` workgroups; flops = blocks * 16 * iters * 32 * 2048 */
int bd_mfma_probe(int blocks, int iters, float* out, void* stream) {
    BD_REQUIRE(blocks > 0 && iters > 0 && out, "bd_mfma_probe: bad arguments");
    static const char* th = getenv("BD_PROBE_THREADS");       // diagnostic: waves per workgroup = threads / 64 (default 16)
    const int threads = th ? atoi(th) : kWsThreads;
    static const char* vr = getenv("BD_PROBE_VARIANT");
    const int variant = vr ? atoi(vr) : 0;
    if (variant == 1) hipLaunchKernelGGL(mfma_probe2_kernel<1>, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, iters, out);
    else if (variant == 2) hipLaunchKernelGGL(mfma_probe2_kernel<2>, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, iters, out);
    else hipLaunchKernelGGL(mfma_probe_kernel, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, iters, out);
    BD_CHECK_LAUNCH("bd_mfma_probe");
    return 0;
}

}  // extern "C"

extern "C" const char* bd_probe_last_error(void) { return bd::err_buf(); }
