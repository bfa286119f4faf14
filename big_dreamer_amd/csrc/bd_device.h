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

#ifndef BD_WAVES
#define BD_WAVES 4
#endif
constexpr int kWaves = BD_WAVES;   // waves per workgroup (BD_WAVES/4 per SIMD)
constexpr int kThreads = kWaves * 64;
constexpr int kFragFloats = 256;   // floats per [16 rows x 16 k] fragment block

__device__ __forceinline__ floatx4 mfma16(float a, float b, floatx4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__host__ __device__ __forceinline__ int cdiv(int a, int b) { return (a + b - 1) / b; }

// threadIdx.x behind an opaque move.  hipcc hoists everything that depends only on the thread index -- per-lane 64-bit
// weight pointers of ~30 matrices, row / column indices and output addresses of every epilogue -- out of the
// persistent kernels' time loops and then SPILLS it (60-68 VGPRs at 8 waves per workgroup): each reload inside a
// step is a scratch load behind s_waitcnt vmcnt(0), i.e. a full memory round trip on the critical path (the
// Gaussian-head phases spent 3-4k cycles on eight of them, s_memtime stamps).  A fresh opaque value per primitive
// call / per time step keeps that arithmetic (a handful of VALU instructions) inside the phase that uses it.
__device__ __forceinline__ int bd_tid() {
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}
// Logical wave index: the hardware wave rotated by the workgroup index.  The column blocks of a 200-wide layer do not
// divide evenly over the waves (13 blocks over 4 waves: one wave gets 4, three get 3), and hardware wave i of every
// workgroup sits on SIMD i; rotating the roles makes the 2-3 co-resident workgroups of the dense-chain kernels
// put their heavy wave on different SIMDs.  Every use of a wave index inside a kernel goes through this.
static_assert((kWaves & (kWaves - 1)) == 0, "kWaves must be a power of two");
__device__ __forceinline__ int bd_wave(int tid) { return ((tid >> 6) + (int)blockIdx.x) & (kWaves - 1); }

// Kernel arguments re-read from the kernarg segment.  The scan kernels take ~60 pointers: kept live across the time loop
// they exceed the 102 SGPRs and hipcc spills them into VGPR lanes -- v_writelane / v_readlane were 1 600 of the 4 800
// VALU instructions of the imagination forward kernel, executed on the SIMD that should be issuing MFMAs (fp32 MFMA and
// VALU do not overlap on gfx950).  BD_KARGS(T) names the by-value argument block through a constant-address-space pointer
// and BD_KARGS_FRESH() makes that pointer opaque again, so that each phase loads the few fields it uses with s_load
// (scalar cache) instead of holding all of them.
#define BD_KARGS(T, name) const __attribute__((address_space(4))) T* name = \
    (const __attribute__((address_space(4))) T*)__builtin_amdgcn_kernarg_segment_ptr()
#define BD_KARGS_FRESH(name) asm volatile("" : "+s"(name))

// Workgroup barrier that orders LDS traffic only.  __syncthreads() makes hipcc emit s_waitcnt vmcnt(0), and
// on CDNA4 vmcnt counts global STORES too: every phase of a persistent kernel would wait for its
// saved-activation stores to be acknowledged by L2.  The tile kernels hand data between phases through LDS
// only (global buffers are write-only or read-only within a launch), so: drain this wave's LDS operations,
// then a bare s_barrier; stores stay in flight.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// float index of element (row, k) of a [16 x 16*Kb] fragment tile
__device__ __forceinline__ int frag_idx(int row, int k) {
    return ((k >> 4) * 64 + ((k >> 2) & 3) * 16 + row) * 4 + (k & 3);
}

// ---- math ----------------------------------------------------------------------------------------
// The activation epilogues sit on the critical path of every phase of the persistent kernels (one wave finishes
// its column blocks, then runs 8 transcendental chains before the workgroup barrier): libm-grade expm1f / log1pf /
// tanhf cost 2-3k cycles per phase (s_memtime stamps), a quarter of a 200x200 layer.  These forms use the hardware
// v_exp_f32 / v_log_f32 / v_rcp_f32 (1 ulp) instead.  Their ABSOLUTE error is <= ~2e-7 for outputs of magnitude
// <= 1 (ELU's negative branch, sigmoid, tanh) and for softplus, which is what the stated fp32 tolerance (2e-5 abs/rel
// on forward tensors) is about; they lose RELATIVE accuracy only where the result itself is ~1e-7.
// -DBD_EXACT_MATH restores the libm forms (matching torch CPU fp32 to the last bits).
#ifdef BD_EXACT_MATH
__device__ __forceinline__ float elu(float x) { return x > 0.f ? x : expm1f(x); }
__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float tanh_act(float x) { return tanhf(x); }
// F.softplus(beta=1, threshold=20)
__device__ __forceinline__ float softplusf(float x) { return x > 20.f ? x : log1pf(expf(x)); }
// 1 - exp(-x), x >= 0  (= sigmoid(raw) when x = softplus(raw))
__device__ __forceinline__ float one_minus_exp_neg(float x) { return -expm1f(-x); }
#else
__device__ __forceinline__ float elu(float x) { return x > 0.f ? x : __expf(x) - 1.f; }
__device__ __forceinline__ float sigmoidf(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float tanh_act(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(__expf(2.f * x) + 1.f); }
__device__ __forceinline__ float softplusf(float x) { return x > 20.f ? x : __logf(1.f + __expf(x)); }
__device__ __forceinline__ float one_minus_exp_neg(float x) { return 1.f - __expf(-x); }
#endif
// derivative of ELU expressed through its output y (y<=0 <=> x<=0): 1 or exp(x) = y+1
__device__ __forceinline__ float elu_grad_from_out(float y) { return y > 0.f ? 1.f : y + 1.f; }
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
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;   // hardware wave: any block size
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double s = 0.0;
    if (bd_tid() == 0) {
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
    for (int idx = bd_tid(); idx < RT * 16 * Kp; idx += blockDim.x) {
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

// The same for the tall chain kernels, two floats per lane: wave w takes rows w, w+kWaves, ..., a lane the column
// pairs 2*lane, 2*lane+128, ...; all of a batch of rows is requested before the first LDS write.  Needs even w0 / ld0 /
// ld1 and 8-byte aligned bases (checked by the caller: tile_pairs_ok); an odd last column is read alone.
__host__ __device__ __forceinline__ bool tile_pairs_ok(const float* in0, int ld0, int w0, const float* in1, int ld1) {
    return ((w0 | ld0) & 1) == 0 && ((uintptr_t)in0 & 7) == 0 && (in1 == nullptr || ((ld1 & 1) == 0 && ((uintptr_t)in1 & 7) == 0));
}
template <int RT>
__device__ __forceinline__ void load_tile_concat_pairs(float* __restrict__ X, int Kb, int row0, int M,
                                                       const float* __restrict__ in0, int ld0, int w0,
                                                       const float* __restrict__ in1, int ld1, int w1) {
    typedef float floatx2 __attribute__((ext_vector_type(2)));
    const int tid = bd_tid(), wave = tid >> 6, lane = tid & 63;
    const int Kp = Kb * 16, W = w0 + w1;
    constexpr int kRows = 16 * RT / kWaves;     // rows per wave
    constexpr int kBatch = 6;
    static_assert(16 * RT % kWaves == 0, "rows must divide over the waves");
    for (int k = 2 * lane; k < Kp; k += 128) {
#pragma unroll
        for (int b0 = 0; b0 < kRows; b0 += kBatch) {
            floatx2 v[kBatch];
#pragma unroll
            for (int b = 0; b < kBatch; ++b) {
                const int r = wave + (b0 + b) * kWaves, grow = row0 + r;
                v[b] = floatx2{0.f, 0.f};
                if (b0 + b < kRows && grow < M) {
                    if (k < w0) v[b] = *reinterpret_cast<const floatx2*>(in0 + (size_t)grow * ld0 + k);
                    else if (k + 1 < W) v[b] = *reinterpret_cast<const floatx2*>(in1 + (size_t)grow * ld1 + (k - w0));
                    else if (k < W) v[b][0] = in1[(size_t)grow * ld1 + (k - w0)];
                }
            }
#pragma unroll
            for (int b = 0; b < kBatch; ++b) {
                const int r = wave + (b0 + b) * kWaves;
                if (b0 + b < kRows)
                    *reinterpret_cast<floatx2*>(X + (r >> 4) * Kb * kFragFloats + frag_idx(r & 15, k)) = v[b];
            }
        }
    }
}

// ---- software-pipelined block loop ------------------------------------------------------------------
// load(kb) returns the fragments of block kb (global weight float4s + LDS activation float4s); mma(frag)
// consumes them.  Two register sets of D blocks each: while the MFMAs consume set A, the loads of the next D
// blocks land in set B (and vice versa), so L2 latency overlaps MFMA issue.  The sets are distinct variables
// on purpose: with a single ring hipcc coalesces "cur = ring[i]; ring[i] = load" into one register range and
// must drain (vmcnt(0)) before it can refill.  The (< D) tail blocks are fetched first and consumed last.
// All indices are compile-time constants (fully unrolled) -> everything is in VGPRs and hipcc emits counted
// s_waitcnt vmcnt(N).  (Measured on MI355X: a deeper flat ring pinned with sched_barrier was slower.)
template <int D, class LoadF, class MmaF>
__device__ __forceinline__ void pipelined_k(int Kb, LoadF&& load, MmaF&& mma) {
    using Frag = decltype(load(0));
    const int G = Kb / D, nt = Kb - G * D;   // full groups, tail blocks (< D)
    Frag A[D], B[D];
    Frag T[D > 1 ? D - 1 : 1];
#pragma unroll
    for (int i = 0; i < D - 1; ++i)
        if (i < nt) T[i] = load(G * D + i);
    if (G > 0) {
#pragma unroll
        for (int i = 0; i < D; ++i) A[i] = load(i);
        int g = 1;
        for (; g + 1 < G; g += 2) {
#pragma unroll
            for (int i = 0; i < D; ++i) B[i] = load(g * D + i);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < D; ++i) mma(A[i]);
#pragma unroll
            for (int i = 0; i < D; ++i) A[i] = load((g + 1) * D + i);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < D; ++i) mma(B[i]);
        }
        if (g < G) {
#pragma unroll
            for (int i = 0; i < D; ++i) B[i] = load(g * D + i);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < D; ++i) mma(A[i]);
#pragma unroll
            for (int i = 0; i < D; ++i) mma(B[i]);
        } else {
#pragma unroll
            for (int i = 0; i < D; ++i) mma(A[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < D - 1; ++i)
        if (i < nt) mma(T[i]);
}

// Three register sets, one K block each: block kb is consumed while kb+1 and kb+2 are in flight (prefetch distance two
// stages).  For the GRU stages (eight to ten float4 per stage, 24 MFMAs = 768 cycles of matrix-pipe time) the two-set form
// covers one stage of latency, less than a loaded L2 round trip: with two waves per SIMD the pipe sits at ~65 % in that
// phase (s_memtime stamps: 62k cycles for a 40k-cycle contraction); D = 2 sets of two blocks spill.  Experiment only:
// the extra live registers cost more than the deeper prefetch buys (see BD_GRU_PIPE3 below).
template <class LoadF, class MmaF>
__device__ __forceinline__ void pipelined_k3(int Kb, LoadF&& load, MmaF&& mma) {
    using Frag = decltype(load(0));
    Frag A = load(0), B = A, C = A;
    if (Kb > 1) B = load(1);
    int kb = 0;
    for (; kb + 3 <= Kb; kb += 3) {
        C = load(kb + 2);
        __builtin_amdgcn_sched_barrier(0);
        mma(A);
        if (kb + 3 < Kb) A = load(kb + 3);
        __builtin_amdgcn_sched_barrier(0);
        mma(B);
        if (kb + 4 < Kb) B = load(kb + 4);
        __builtin_amdgcn_sched_barrier(0);
        mma(C);
    }
    if (kb < Kb) mma(A);
    if (kb + 1 < Kb) mma(B);
}

// ---- the core contraction: out[16*RT x N] = sum_s X_s[16*RT x K_s] * W_s^T + bias -----------------------
// The reference's concatenated inputs (torch.cat([belief, state]), cat(state, action)) are kept as separate
// LDS fragment tiles ("segments") with separately packed weight column blocks.
struct Seg {
    const float* X;   // LDS fragment tile(s) [RT][Kb][64][4]
    const float* W;   // packed weights for this column block (out = N, in = 16*Kb)
    int Kb;
};

template <int RT, int NI>
struct LinFrag {
    floatx4 a[RT];
    floatx4 b[NI];
};

// Epilogue operands (saved activations, noise, hoisted projections: cold, streamed-once HBM data) must be in
// flight BEFORE the contraction, or every phase pays an extra serial HBM round trip (~5k cycles, measured with
// s_memtime stamps).  `pre(rt, nb)` is called per accumulator before the K loop and its result is handed to
// `epi(rt, nb, acc, pre_result)`.  NoPre is the default for epilogues without such operands.
struct NoPreVal {};
struct Pre4 {            // four scalars, one per accumulator row r = 0..3
    float v[4];
};
struct PreGate {         // GRU backward operands per accumulator row: r, z, n, (W_hn h + b_hn), h_prev, d feat
    float r[4], z[4], n[4], hn[4], hprev[4], dfeat[4];
};
struct NoPre {
    __device__ __forceinline__ NoPreVal operator()(int, int) const { return NoPreVal{}; }
};

// Stores of activations that are saved for the backward pass (written once, read once much later): with -DBD_NT_SAVES=1
// they carry the non-temporal hint, so that the stream of saves does not push the step's weights out of the XCD's L2.
#ifndef BD_NT_SAVES
#define BD_NT_SAVES 0
#endif
__device__ __forceinline__ void st_save(float* p, float v) {
#if BD_NT_SAVES
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}

// Diagnostic build only (-DBD_STAMPS): s_memtime at points inside a tile primitive, recorded by thread 0 of workgroup 0
// when the caller passes a slot base >= 0 (each translation unit has its own table).  Never in the shipped .so.
// The slot bound matters: primitives forward `sb + 16` to the primitives nested in them, and an unchecked base of 48 once
// stamped slots 64-69 -- past the table, into the unmapped page behind the module's data segment (gpurun_out/stamps39.log).
#ifdef BD_STAMPS
#ifndef BD_STAMP_BLOCK
#define BD_STAMP_BLOCK 0          // workgroup that records (-DBD_STAMP_BLOCK=n)
#endif
#ifndef BD_STAMP_THREAD
#define BD_STAMP_THREAD 0
#endif
static __device__ unsigned long long g_dstamps[64];
#define BD_DSTAMP(base, k)                                                                                  \
    do {                                                                                                    \
        if ((base) >= 0 && (base) + (k) < 64 && blockIdx.x == BD_STAMP_BLOCK && threadIdx.x == BD_STAMP_THREAD) \
            g_dstamps[(base) + (k)] = __builtin_amdgcn_s_memtime();                                          \
    } while (0)
#else
#define BD_DSTAMP(base, k)
#endif

// Accumulators in the transposed form (linear_sweep<..., TR = true>): lane holds row lane&15, columns
// nb*16 + 4*(lane>>4) + r.  N % 4 == 0 (host-checked by the tall chain launchers); the bias vector sits at an arbitrary
// float offset of the flat parameter buffer, so it is read with scalar loads.
__device__ __forceinline__ floatx4 tall_bias(const float* __restrict__ bias, int N, int nb, int lane) {
    const int col0 = nb * 16 + 4 * (lane >> 4);
    floatx4 b = floatx4{0.f, 0.f, 0.f, 0.f};
    if (bias != nullptr && col0 < N) {
        if (col0 + 4 <= N) {
            __builtin_memcpy(&b, bias + col0, 16);      // one global_load_dwordx4 (4-byte aligned address)
        } else {                                        // (the scans take any width)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (col0 + r < N) b[r] = bias[col0 + r];
        }
    }
    return b;
}

// TR: the two MFMA operands change places, D^T = W X^T.  The registers are the same ones (a weight fragment is a valid A
// operand, an activation fragment a valid B operand); what changes is the accumulator: lane holds
// out[row = lane&15][col = nb*16 + 4*(lane>>4) + r], r = 0..3 -- four CONSECUTIVE columns of one row, which is exactly
// one lane's 16 bytes of the next layer's fragment tile (frag[nb][lane][r]) and 16 contiguous bytes of a row-major
// output.  The epilogue becomes one ds_write_b128 + one global_store_dwordx4 per block instead of four conflicting
// ds_write_b32 + four dword stores.
template <int NSEG, int RT, int NI, int D, bool TR = false>
__device__ __forceinline__ void linear_sweep(const Seg (&seg)[NSEG], int nb0, floatx4 (*acc)[RT]) {
    const int lane = bd_tid() & 63;
    constexpr bool kSplit = (NI * RT == 1);
    floatx4 acc2 = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NSEG; ++s) {
        const int Kb = seg[s].Kb;
        const floatx4* __restrict__ X4 = reinterpret_cast<const floatx4*>(seg[s].X) + lane;
        const floatx4* __restrict__ W4 = reinterpret_cast<const floatx4*>(seg[s].W) + lane + (size_t)nb0 * Kb * 64;
        const size_t wstride = (size_t)kWaves * Kb * 64;
        pipelined_k<D>(
            Kb,
            [&](int kb) {
                LinFrag<RT, NI> f;
#pragma unroll
                for (int i = 0; i < NI; ++i) f.b[i] = W4[i * wstride + (size_t)kb * 64];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) f.a[rt] = X4[(rt * Kb + kb) * 64];
                return f;
            },
            [&](const LinFrag<RT, NI>& f) {
                if constexpr (kSplit) {
                    if constexpr (TR) {
                        acc[0][0] = mfma16(f.b[0][0], f.a[0][0], acc[0][0]);
                        acc2 = mfma16(f.b[0][1], f.a[0][1], acc2);
                        acc[0][0] = mfma16(f.b[0][2], f.a[0][2], acc[0][0]);
                        acc2 = mfma16(f.b[0][3], f.a[0][3], acc2);
                    } else {
                        acc[0][0] = mfma16(f.a[0][0], f.b[0][0], acc[0][0]);
                        acc2 = mfma16(f.a[0][1], f.b[0][1], acc2);
                        acc[0][0] = mfma16(f.a[0][2], f.b[0][2], acc[0][0]);
                        acc2 = mfma16(f.a[0][3], f.b[0][3], acc2);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int i = 0; i < NI; ++i)
#pragma unroll
                            for (int rt = 0; rt < RT; ++rt)
                                acc[i][rt] = TR ? mfma16(f.b[i][j], f.a[rt][j], acc[i][rt]) : mfma16(f.a[rt][j], f.b[i][j], acc[i][rt]);
                }
            });
    }
    if constexpr (kSplit) acc[0][0] += acc2;
}

// NI column blocks nb0, nb0+kWaves, ... of this wave (all valid), RT row tiles, all segments.
template <int NSEG, int RT, int NI, int D, bool TR = false, class Pre, class Epi>
__device__ __forceinline__ void linear_blocks(const Seg (&seg)[NSEG], const float* __restrict__ bias, int N, int nb0,
                                              Pre&& pre, Epi&& epi, int sb = -1) {
    const int lane = bd_tid() & 63;
    BD_DSTAMP(sb, 0);
    decltype(pre(0, 0)) pf[NI][RT];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) pf[i][rt] = pre(rt, nb0 + i * kWaves);
    floatx4 acc[NI][RT];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        floatx4 b4;
        if constexpr (TR) {
            b4 = tall_bias(bias, N, nb0 + i * kWaves, lane);
        } else {
            const int col = (nb0 + i * kWaves) * 16 + (lane & 15);
            const float b = (bias != nullptr && col < N) ? bias[col] : 0.f;
            b4 = floatx4{b, b, b, b};
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[i][rt] = b4;
    }
    BD_DSTAMP(sb, 1);
    linear_sweep<NSEG, RT, NI, D, TR>(seg, nb0, acc);
    BD_DSTAMP(sb, 2);
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) epi(rt, nb0 + i * kWaves, acc[i][rt], pf[i][rt]);
    BD_DSTAMP(sb, 3);
}

// ---- split-K over waves for narrow outputs ----------------------------------------------------------------
// When a layer has at most two (column block, row tile) pairs (N <= 32: the mean/std heads, the N=1 heads of the
// reward/value chains, the d/d(state|action) of the embed layer) the column-block decomposition leaves all but
// one or two waves idle while those walk every K block in sequence: ~one L2 round trip per block pair.  Here
// every wave takes the K blocks kb = wave, wave+kWaves, ... of ALL pairs, partial accumulators meet in an LDS
// scratch ([kWaves][pairs][64 lanes] float4) and the first `pairs` waves sum them in fixed order (deterministic)
// and run the epilogue.  Costs one extra barrier, removes ~Kb/kWaves round trips from the step's critical path.
constexpr int kSplitPairs = 2;
constexpr int kHeadMaxN = 64;                                   // widest Gaussian head tile_dual_head_elem spreads
constexpr int kHeadPlainFloats = 2 * 16 * kHeadMaxN;            // [2][16 rows][<= 64 columns] behind the partials
constexpr int kSplitPartialFloats = kWaves * kSplitPairs * 2 * kFragFloats;   // dual form needs 2 outputs per pair
constexpr int kSplitScratchFloats = kSplitPartialFloats + kHeadPlainFloats;

template <int RT, int NSEG, bool TR = false, class Pre, class Epi>
__device__ __forceinline__ void tile_linear_splitk(const Seg (&seg)[NSEG], const float* __restrict__ bias, int N,
                                                   float* __restrict__ scratch, Pre&& pre, Epi&& epi) {
    const int lane = bd_tid() & 63, wave = bd_wave(bd_tid());
    const int Nb = (N + 15) >> 4;
    const int P = Nb * RT;                      // <= kSplitPairs (checked by the caller)
    // the reducing waves fetch their bias and epilogue operands up front
    const int my_nb = wave / RT, my_rt = wave - my_nb * RT;
    decltype(pre(0, 0)) pf{};
    floatx4 my_bias = floatx4{0.f, 0.f, 0.f, 0.f};
    if (wave < P) {
        pf = pre(my_rt, my_nb);
        if constexpr (TR) {
            my_bias = tall_bias(bias, N, my_nb, lane);
        } else {
            const int col = my_nb * 16 + (lane & 15);
            const float b = (bias != nullptr && col < N) ? bias[col] : 0.f;
            my_bias = floatx4{b, b, b, b};
        }
    }
    floatx4 acc[kSplitPairs];
#pragma unroll
    for (int p = 0; p < kSplitPairs; ++p) acc[p] = floatx4{0.f, 0.f, 0.f, 0.f};
    // A wave's share is tiny (<= 2 K blocks x <= 2 pairs per segment for K <= 16*kWaves): issue ALL its loads
    // before the first MFMA, so the share costs one L2 round trip instead of one per (k block, pair).
    constexpr int KW = 2;
#pragma unroll
    for (int s = 0; s < NSEG; ++s) {
        const int Kb = seg[s].Kb;
        const floatx4* __restrict__ X4 = reinterpret_cast<const floatx4*>(seg[s].X) + lane;
        const floatx4* __restrict__ W4 = reinterpret_cast<const floatx4*>(seg[s].W) + lane;
        for (int kb0 = wave; kb0 < Kb; kb0 += kWaves * KW) {
            floatx4 xa[KW][kSplitPairs], wb[KW][kSplitPairs];
#pragma unroll
            for (int i = 0; i < KW; ++i) {
                const int kb = kb0 + i * kWaves;
#pragma unroll
                for (int p = 0; p < kSplitPairs; ++p) {
                    if (kb < Kb && p < P) {
                        const int nb = p / RT, rt = p % RT;
                        xa[i][p] = X4[(rt * Kb + kb) * 64];
                        wb[i][p] = W4[((size_t)nb * Kb + kb) * 64];
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < KW; ++i) {
                const int kb = kb0 + i * kWaves;
#pragma unroll
                for (int p = 0; p < kSplitPairs; ++p) {
                    if (kb < Kb && p < P) {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[p] = TR ? mfma16(wb[i][p][j], xa[i][p][j], acc[p]) : mfma16(xa[i][p][j], wb[i][p][j], acc[p]);
                    }
                }
            }
        }
    }
    floatx4* __restrict__ S4 = reinterpret_cast<floatx4*>(scratch);
#pragma unroll
    for (int p = 0; p < kSplitPairs; ++p)
        if (p < P) S4[(wave * kSplitPairs + p) * 64 + lane] = acc[p];
    lds_barrier();
    if (wave < P) {
        floatx4 r = my_bias;
        for (int w = 0; w < kWaves; ++w) r += S4[(w * kSplitPairs + wave) * 64 + lane];
        epi(my_rt, my_nb, r, pf);
    }
}

// Each wave owns column blocks nb = wave, wave+kWaves, ...; two are kept in flight where two exist (independent
// MFMA chains: 16x16x4 f32 issues every 32 cycles but a dependent one needs 40), the odd last one runs alone.
// epi(rt, nb, acc): lane holds out[row = 16*rt + 4*(lane>>4) + r][col = nb*16 + (lane&15)], r = 0..3.
template <int RT, int NSEG, bool TR = false, class Pre, class Epi>
__device__ __forceinline__ void tile_linear_pre(const Seg (&seg)[NSEG], const float* __restrict__ bias, int N, Pre&& pre,
                                                Epi&& epi, float* __restrict__ scratch = nullptr, int sb = -1) {
    const int wave = bd_wave(bd_tid());
    const int Nb = (N + 15) >> 4;
    if (scratch != nullptr && Nb * RT <= kSplitPairs) {   // workgroup-uniform
        tile_linear_splitk<RT, NSEG, TR>(seg, bias, N, scratch, pre, epi);
        return;
    }
#ifndef BD_PIPE_D
#define BD_PIPE_D 2
#endif
    constexpr int D = BD_PIPE_D;      // K blocks per register set of the software pipeline (two sets); measured on
                                      // MI355X after the kernels became spill-free: D = 3 / 4 are 4-6 % slower than 2
    for (int nb0 = wave; nb0 < Nb; nb0 += 2 * kWaves) {
        if (nb0 + kWaves < Nb) linear_blocks<NSEG, RT, 2, D, TR>(seg, bias, N, nb0, pre, epi, sb);
        else linear_blocks<NSEG, RT, 1, D, TR>(seg, bias, N, nb0, pre, epi, sb);
    }
}

template <int RT, int NSEG, class Epi>
__device__ __forceinline__ void tile_linear_g(const Seg (&seg)[NSEG], const float* __restrict__ bias, int N, Epi&& epi,
                                              float* __restrict__ scratch = nullptr, int sb = -1) {
    tile_linear_pre<RT, NSEG>(seg, bias, N, NoPre{}, [&](int rt, int nb, floatx4 acc, NoPreVal) { epi(rt, nb, acc); },
                              scratch, sb);
}

// single-segment convenience forms
template <int RT, class Epi>
__device__ __forceinline__ void tile_linear(const float* __restrict__ X, int Kb, const float* __restrict__ Wp,
                                            const float* __restrict__ bias, int N, Epi&& epi,
                                            float* __restrict__ scratch = nullptr) {
    const Seg seg[1] = {{X, Wp, Kb}};
    tile_linear_g<RT, 1>(seg, bias, N, epi, scratch);
}

template <int NSEG, class Epi>
__device__ __forceinline__ void tile_linear_seg(const Seg (&seg)[NSEG], const float* __restrict__ bias, int N,
                                                Epi&& epi, float* __restrict__ scratch = nullptr, int sb = -1) {
    tile_linear_g<1, NSEG>(seg, bias, N, [&](int, int nb, floatx4 acc) { epi(nb, acc); }, scratch, sb);
}
// the same with TRANSPOSED accumulators (linear_sweep<TR>): epi(nb, acc) gets out[row = lane & 15][nb*16 + 4*(lane>>4) + r]
template <int NSEG, class Epi>
__device__ __forceinline__ void tile_linear_seg_tr(const Seg (&seg)[NSEG], const float* __restrict__ bias, int N, Epi&& epi,
                                                   int sb = -1) {
    tile_linear_pre<1, NSEG, true>(seg, bias, N, NoPre{}, [&](int, int nb, floatx4 acc, NoPreVal) { epi(nb, acc); }, nullptr, sb);
}

// ---- tall workgroups: RT row tiles, balanced (row tile, column block) pairs, epilogue deferred -----------------
// A 200-wide layer has 13 column blocks; handing whole blocks to 4 waves gives 4+3+3+3, i.e. the matrix pipe of three
// SIMDs idles for a quarter of every sweep.  With RT row tiles per workgroup the unit of work is the (row tile, block)
// pair: every wave takes the blocks wave, wave+4, ... of ALL row tiles (`per` = Nb / kWaves of them: one weight
// fragment feeds RT MFMAs, one activation fragment feeds `per`), and the RT pairs of each leftover block go round-robin
// over the waves, one at most per wave (13 blocks, RT = 3: 9 + 1 pairs on three waves, 9 on the fourth).
// The accumulators stay in registers across a workgroup barrier (TallAcc) so the layer's output can overwrite its
// input in LDS: one image per workgroup instead of two, which is what lets three 48-row workgroups share a CU.
constexpr int kTallMaxPer = 3;    // column blocks per wave in the main part (N <= 15 blocks + leftover rule, host-checked)

template <int RT>
struct TallAcc {
    floatx4 main[kTallMaxPer][RT];
    floatx4 left;
};

__host__ __device__ __forceinline__ bool tall_shape_ok(int N, int RT) {
    const int Nb = (N + 15) >> 4, per = Nb / kWaves;
    return per <= kTallMaxPer && (Nb - per * kWaves) * RT <= kWaves;
}

// pair owned by this wave among the leftover blocks: returns false when it has none
__device__ __forceinline__ bool tall_left_pair(int Nb, int RT, int wave, int& rt, int& nb) {
    const int per = Nb / kWaves, nleft = (Nb - per * kWaves) * RT;
    if (wave >= nleft) return false;
    const int lb = wave / RT;
    rt = wave - lb * RT;
    nb = per * kWaves + lb;
    return true;
}

template <int RT, int NSEG>
__device__ __forceinline__ void tall_sweep(const Seg (&seg)[NSEG], const float* __restrict__ bias, int N, TallAcc<RT>& t) {
    const int lane = bd_tid() & 63, wave = bd_wave(bd_tid());
    const int Nb = (N + 15) >> 4, per = Nb / kWaves;
#pragma unroll
    for (int i = 0; i < kTallMaxPer; ++i) {
        const floatx4 b = i < per ? tall_bias(bias, N, wave + i * kWaves, lane) : floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) t.main[i][rt] = b;
    }
    if (per == 3) linear_sweep<NSEG, RT, 3, 1, true>(seg, wave, t.main);
    else if (per == 2) linear_sweep<NSEG, RT, 2, 1, true>(seg, wave, t.main);
    else if (per == 1) linear_sweep<NSEG, RT, 1, 2, true>(seg, wave, t.main);
    int lrt = 0, lnb = 0;
    if (tall_left_pair(Nb, RT, wave, lrt, lnb)) {
        floatx4 one[1][1] = {{tall_bias(bias, N, lnb, lane)}};
        Seg s1[NSEG];
#pragma unroll
        for (int s = 0; s < NSEG; ++s) s1[s] = Seg{seg[s].X + (size_t)lrt * seg[s].Kb * kFragFloats, seg[s].W, seg[s].Kb};
        linear_sweep<NSEG, 1, 1, 2, true>(s1, lnb, one);
        t.left = one[0][0];
    }
}

// f(rt, nb, acc) for every pair this wave holds
template <int RT, class F>
__device__ __forceinline__ void tall_foreach(int N, const TallAcc<RT>& t, F&& f) {
    const int wave = bd_wave(bd_tid());
    const int Nb = (N + 15) >> 4, per = Nb / kWaves;
#pragma unroll
    for (int i = 0; i < kTallMaxPer; ++i)
        if (i < per) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) f(rt, wave + i * kWaves, t.main[i][rt]);
        }
    int lrt = 0, lnb = 0;
    if (tall_left_pair(Nb, RT, wave, lrt, lnb)) f(lrt, lnb, t.left);
}

// t[pair] = g(rt, nb) for every pair this wave holds (epilogue operands fetched ahead of a barrier)
template <int RT, class G>
__device__ __forceinline__ void tall_fill(int N, TallAcc<RT>& t, G&& g) {
    const int wave = bd_wave(bd_tid());
    const int Nb = (N + 15) >> 4, per = Nb / kWaves;
#pragma unroll
    for (int i = 0; i < kTallMaxPer; ++i)
        if (i < per) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) t.main[i][rt] = g(rt, wave + i * kWaves);
        }
    int lrt = 0, lnb = 0;
    if (tall_left_pair(Nb, RT, wave, lrt, lnb)) t.left = g(lrt, lnb);
}

template <int RT, class F>
__device__ __forceinline__ void tall_foreach2(int N, const TallAcc<RT>& t, const TallAcc<RT>& u, F&& f) {
    const int wave = bd_wave(bd_tid());
    const int Nb = (N + 15) >> 4, per = Nb / kWaves;
#pragma unroll
    for (int i = 0; i < kTallMaxPer; ++i)
        if (i < per) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) f(rt, wave + i * kWaves, t.main[i][rt], u.main[i][rt]);
        }
    int lrt = 0, lnb = 0;
    if (tall_left_pair(Nb, RT, wave, lrt, lnb)) f(lrt, lnb, t.left, u.left);
}

// Two outputs sharing one column index (mean / raw-std rows of a Gaussian head): out0 = sum_s X_s W0_s^T + bias0,
// out1 = sum_s X_s W1_s^T + bias1 (a null W0/W1 means the segment does not feed that output).  The two chains
// are independent, which also hides the MFMA dependent-issue latency.
struct Seg2 {
    const float* X;
    const float* W0;
    const float* W1;
    int Kb;
};

struct DualFrag {
    floatx4 a, p, q;
};

template <int NSEG, class Pre, class Epi>
__device__ __forceinline__ void tile_linear_dual_pre(const Seg2 (&seg)[NSEG], const float* __restrict__ bias0,
                                                     const float* __restrict__ bias1, int N, Pre&& pre, Epi&& epi,
                                                     float* __restrict__ scratch = nullptr, int sb = -1) {
    const int lane = bd_tid() & 63, wave = bd_wave(bd_tid());
    const int Nb = (N + 15) >> 4;
    if (scratch != nullptr && Nb <= kSplitPairs) {   // split-K over waves (see tile_linear_splitk); both W0, W1 given
        BD_DSTAMP(sb, 0);
        decltype(pre(0)) pf{};
        float mb0 = 0.f, mb1 = 0.f;
        if (wave < Nb) {                              // reducing waves: bias + epilogue operands up front
            pf = pre(wave);
            const int col = wave * 16 + (lane & 15);
            mb0 = (bias0 != nullptr && col < N) ? bias0[col] : 0.f;
            mb1 = (bias1 != nullptr && col < N) ? bias1[col] : 0.f;
        }
        floatx4 a0[kSplitPairs], a1[kSplitPairs];
#pragma unroll
        for (int p = 0; p < kSplitPairs; ++p) a0[p] = a1[p] = floatx4{0.f, 0.f, 0.f, 0.f};
        constexpr int KW = 2;      // all loads of a wave's share first, then the MFMAs (see tile_linear_splitk)
#pragma unroll
        for (int s = 0; s < NSEG; ++s) {
            const int Kb = seg[s].Kb;
            const floatx4* __restrict__ X4 = reinterpret_cast<const floatx4*>(seg[s].X) + lane;
            const floatx4* __restrict__ W0 = reinterpret_cast<const floatx4*>(seg[s].W0) + lane;
            const floatx4* __restrict__ W1 = reinterpret_cast<const floatx4*>(seg[s].W1) + lane;
            for (int kb0 = wave; kb0 < Kb; kb0 += kWaves * KW) {
                floatx4 xa[KW], pw[KW][kSplitPairs], qw[KW][kSplitPairs];
#pragma unroll
                for (int i = 0; i < KW; ++i) {
                    const int kb = kb0 + i * kWaves;
                    if (kb < Kb) {
                        xa[i] = X4[kb * 64];
#pragma unroll
                        for (int p = 0; p < kSplitPairs; ++p)
                            if (p < Nb) {
                                pw[i][p] = W0[((size_t)p * Kb + kb) * 64];
                                qw[i][p] = W1[((size_t)p * Kb + kb) * 64];
                            }
                    }
                }
#pragma unroll
                for (int i = 0; i < KW; ++i) {
                    const int kb = kb0 + i * kWaves;
                    if (kb < Kb) {
#pragma unroll
                        for (int p = 0; p < kSplitPairs; ++p)
                            if (p < Nb) {
#pragma unroll
                                for (int j = 0; j < 4; ++j) {
                                    a0[p] = mfma16(xa[i][j], pw[i][p][j], a0[p]);
                                    a1[p] = mfma16(xa[i][j], qw[i][p][j], a1[p]);
                                }
                            }
                    }
                }
            }
        }
        BD_DSTAMP(sb, 1);
        floatx4* __restrict__ S4 = reinterpret_cast<floatx4*>(scratch);
#pragma unroll
        for (int p = 0; p < kSplitPairs; ++p)
            if (p < Nb) {
                S4[((wave * kSplitPairs + p) * 2 + 0) * 64 + lane] = a0[p];
                S4[((wave * kSplitPairs + p) * 2 + 1) * 64 + lane] = a1[p];
            }
        BD_DSTAMP(sb, 2);
        lds_barrier();
        BD_DSTAMP(sb, 3);
        if (wave < Nb) {
            floatx4 r0 = floatx4{mb0, mb0, mb0, mb0}, r1 = floatx4{mb1, mb1, mb1, mb1};
            for (int w = 0; w < kWaves; ++w) {
                r0 += S4[((w * kSplitPairs + wave) * 2 + 0) * 64 + lane];
                r1 += S4[((w * kSplitPairs + wave) * 2 + 1) * 64 + lane];
            }
            BD_DSTAMP(sb, 4);
            epi(wave, r0, r1, pf);
            BD_DSTAMP(sb, 5);
        }
        return;
    }
    for (int nb = wave; nb < Nb; nb += kWaves) {
        const auto pf = pre(nb);
        const int col = nb * 16 + (lane & 15);
        const float b0 = (bias0 != nullptr && col < N) ? bias0[col] : 0.f;
        const float b1 = (bias1 != nullptr && col < N) ? bias1[col] : 0.f;
        floatx4 acc0 = floatx4{b0, b0, b0, b0}, acc1 = floatx4{b1, b1, b1, b1};
#pragma unroll
        for (int s = 0; s < NSEG; ++s) {
            const floatx4* __restrict__ X4 = reinterpret_cast<const floatx4*>(seg[s].X) + lane;
            const int Kb = seg[s].Kb;
            const size_t off = (size_t)nb * Kb * 64 + lane;
            const floatx4* __restrict__ W0 = reinterpret_cast<const floatx4*>(seg[s].W0) + off;
            const floatx4* __restrict__ W1 = reinterpret_cast<const floatx4*>(seg[s].W1) + off;
            if (seg[s].W0 != nullptr && seg[s].W1 != nullptr) {
                pipelined_k<2>(
                    Kb, [&](int kb) { return DualFrag{X4[kb * 64], W0[kb * 64], W1[kb * 64]}; },
                    [&](const DualFrag& f) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            acc0 = mfma16(f.a[j], f.p[j], acc0);
                            acc1 = mfma16(f.a[j], f.q[j], acc1);
                        }
                    });
            } else if (seg[s].W0 != nullptr) {
                for (int kb = 0; kb < Kb; ++kb) {
                    const floatx4 a4 = X4[kb * 64], p4 = W0[kb * 64];
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc0 = mfma16(a4[j], p4[j], acc0);
                }
            } else if (seg[s].W1 != nullptr) {
                for (int kb = 0; kb < Kb; ++kb) {
                    const floatx4 a4 = X4[kb * 64], q4 = W1[kb * 64];
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc1 = mfma16(a4[j], q4[j], acc1);
                }
            }
        }
        epi(nb, acc0, acc1, pf);
    }
}

struct NoPre1 {
    __device__ __forceinline__ NoPreVal operator()(int) const { return NoPreVal{}; }
};
template <int NSEG, class Epi>
__device__ __forceinline__ void tile_linear_dual(const Seg2 (&seg)[NSEG], const float* __restrict__ bias0,
                                                 const float* __restrict__ bias1, int N, Epi&& epi,
                                                 float* __restrict__ scratch = nullptr) {
    tile_linear_dual_pre<NSEG>(seg, bias0, bias1, N, NoPre1{},
                               [&](int nb, floatx4 a0, floatx4 a1, NoPreVal) { epi(nb, a0, a1); }, scratch);
}

// Gaussian head with an ELEMENT-wise epilogue.  tile_linear_dual_pre hands each (mean, raw) accumulator to the wave
// that reduced it: for a narrow head that is one or two waves running 4 rows x (softplus, tanh, ...) chains per lane
// back to back while the other waves wait (12k cycles of the 18k-cycle actor head, s_memtime stamps).  Here the
// reduced pre-activations go to LDS ([2][16][16*Nb] behind the split-K partials), and after one more barrier every
// thread finishes ONE (row, column) element: pre_elem(row, col) fetches its streamed operand (noise) before the
// contraction, epi_elem(row, col, mean_pre, raw_pre, operand) does the math and the stores.  N <= kHeadMaxN.
template <int NSEG, class PreE, class EpiE>
__device__ __forceinline__ void tile_dual_head_elem(const Seg2 (&seg)[NSEG], const float* __restrict__ bias0,
                                                    const float* __restrict__ bias1, int N, float* __restrict__ scratch,
                                                    PreE&& pre_elem, EpiE&& epi_elem, int sb = -1) {
    const int lane = bd_tid() & 63;
    const int Nb = (N + 15) >> 4, Np = Nb * 16;
    float* __restrict__ plain = scratch + kSplitPartialFloats;
    const int e0 = bd_tid();
    const int row0e = e0 / N, col0e = e0 - row0e * N;
    decltype(pre_elem(0, 0)) pv{};
    BD_DSTAMP(sb, 0);
#ifdef BD_STAMPS
    if (sb >= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // diagnostic: drain what the earlier phases left in flight
    BD_DSTAMP(sb, 6);
#endif
    if (e0 < 16 * N) pv = pre_elem(row0e, col0e);
#ifdef BD_STAMPS
    if (sb >= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // diagnostic: latency of the operand load alone
    BD_DSTAMP(sb, 7);
#endif
    BD_DSTAMP(sb, 1);
    tile_linear_dual_pre<NSEG>(
        seg, bias0, bias1, N, NoPre1{},
        [&](int nb, floatx4 a0, floatx4 a1, NoPreVal) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = (4 * (lane >> 4) + r) * Np + nb * 16 + (lane & 15);
                plain[o] = a0[r];
                plain[16 * Np + o] = a1[r];
            }
        },
        Nb <= kSplitPairs ? scratch : nullptr, sb >= 0 ? sb + 16 : -1);
    BD_DSTAMP(sb, 2);
    lds_barrier();
    BD_DSTAMP(sb, 3);
#ifdef BD_STAMPS
    if (sb >= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // diagnostic: separate the operand wait from the math
    BD_DSTAMP(sb, 4);
#endif
    if (e0 < 16 * N) epi_elem(row0e, col0e, plain[row0e * Np + col0e], plain[16 * Np + row0e * Np + col0e], pv);
    BD_DSTAMP(sb, 5);
    for (int e = e0 + blockDim.x; e < 16 * N; e += blockDim.x) {
        const int row = e / N, col = e - row * N;
        epi_elem(row, col, plain[row * Np + col], plain[16 * Np + row * Np + col], pre_elem(row, col));
    }
}

// (-DBD_GRU_PIPE3=1 selects the three-set pipeline for the GRU stages: measured SLOWER on MI355X -- imagination forward
// 0.93 vs 0.88 ms alone, step 3.54 vs 3.46 ms -- so the two-set form stays)
#ifndef BD_GRU_PIPE3
#define BD_GRU_PIPE3 0
#endif
#if BD_GRU_PIPE3
#define BD_GRU_PIPE pipelined_k3
#else
#define BD_GRU_PIPE pipelined_k<1>
#endif

// ---- GRU cell (nn.GRUCell, src/models.py:149,252): four accumulators per output column block ----------
//   R  = W_ir x + W_hr h + b_ir + b_hr      Z  = W_iz x + W_hz h + b_iz + b_hz
//   NI = W_in x + b_in                      NH = W_hn h + b_hn
// epi(nb, R, Z, NI, NH): lane holds rows 4*(lane>>4)+r, column nb*16+(lane&15).
struct GruW {
    const float *w_ir, *w_iz, *w_in, *w_hr, *w_hz, *w_hn;   // packed (Be, Be) each
    const float *b_ih, *b_hh;                               // [3*Be]
};

struct GruFrag {
    floatx4 ax, ah, bir, biz, bin, bhr, bhz, bhn;
};

struct GruHalfFrag {
    floatx4 ax, ah, b0, b1;
};

// One column block of the cell: the four gate accumulators of block nb from the biases up, handed to epi.
template <class Epi>
__device__ __forceinline__ void gru_block(const floatx4* __restrict__ X4, const floatx4* __restrict__ H4, int Kb, int Be,
                                          const GruW& w, int nb, int lane, Epi&& epi) {
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
    BD_GRU_PIPE(
        Kb,
        [&](int kb) {
            return GruFrag{X4[kb * 64], H4[kb * 64], Wir[kb * 64], Wiz[kb * 64], Win[kb * 64],
                           Whr[kb * 64], Whz[kb * 64], Whn[kb * 64]};
        },
        [&](const GruFrag& f) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                R = mfma16(f.ax[j], f.bir[j], R);
                Z = mfma16(f.ax[j], f.biz[j], Z);
                NI = mfma16(f.ax[j], f.bin[j], NI);
                NH = mfma16(f.ah[j], f.bhn[j], NH);
                R = mfma16(f.ah[j], f.bhr[j], R);
                Z = mfma16(f.ah[j], f.bhz[j], Z);
            }
        });
    epi(nb, R, Z, NI, NH);
}

// `scratch` (>= 4 x 64 float4 of LDS, optional): with 8 waves and 13 column blocks (Be = 200) the second round of blocks
// 8..12 leaves waves 5, 6, 7 idle while the SIMD of waves 0 and 4 carries four blocks against three on the others.  The
// three idle waves then take block 12 BY GATE -- wave 5 the reset gate (W_ir x + W_hr h), wave 6 the update gate, wave 7 the
// two candidate products -- the four accumulators meet in `scratch`, and after one barrier wave 4 (which gave the block up)
// runs its epilogue: 3 + 1/3 blocks on the fullest SIMD instead of 4 (312 MFMAs = 10k cycles of every step).
template <class Epi>
__device__ __forceinline__ void gru_tile(const float* __restrict__ X, const float* __restrict__ Hf, int Kb, int Be,
                                         const GruW& w, Epi&& epi, float* __restrict__ scratch = nullptr) {
    const int lane = bd_tid() & 63, wave = bd_wave(bd_tid());
    const int Nb = (Be + 15) >> 4;
    const floatx4* __restrict__ X4 = reinterpret_cast<const floatx4*>(X) + lane;
    const floatx4* __restrict__ H4 = reinterpret_cast<const floatx4*>(Hf) + lane;
    if (kWaves == 8 && Nb == 13 && scratch != nullptr) {        // workgroup-uniform
        gru_block(X4, H4, Kb, Be, w, wave, lane, epi);             // blocks 0..7
        floatx4* __restrict__ S4 = reinterpret_cast<floatx4*>(scratch);
        if (wave < 4) {
            gru_block(X4, H4, Kb, Be, w, wave + 8, lane, epi);     // blocks 8..11
        } else if (wave > 4) {
            const int nb = 12, col = nb * 16 + (lane & 15);
            const bool ok = col < Be;
            const int g = wave - 5;                                // 0: reset, 1: update, 2: candidate
            const float b0 = !ok ? 0.f : (g == 2 ? w.b_ih[2 * Be + col] : w.b_ih[g * Be + col] + w.b_hh[g * Be + col]);
            const float b1 = (ok && g == 2) ? w.b_hh[2 * Be + col] : 0.f;
            floatx4 A0 = floatx4{b0, b0, b0, b0}, A1 = floatx4{b1, b1, b1, b1};    // x product | h product
            const size_t off = (size_t)nb * Kb * 64 + lane;
            const float* wx = g == 0 ? w.w_ir : (g == 1 ? w.w_iz : w.w_in);
            const float* wh = g == 0 ? w.w_hr : (g == 1 ? w.w_hz : w.w_hn);
            const floatx4* __restrict__ Wx = reinterpret_cast<const floatx4*>(wx) + off;
            const floatx4* __restrict__ Wh = reinterpret_cast<const floatx4*>(wh) + off;
            pipelined_k<2>(
                Kb, [&](int kb) { return GruHalfFrag{X4[kb * 64], H4[kb * 64], Wx[kb * 64], Wh[kb * 64]}; },
                [&](const GruHalfFrag& f) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        A0 = mfma16(f.ax[j], f.b0[j], A0);
                        A1 = mfma16(f.ah[j], f.b1[j], A1);
                    }
                });
            if (g == 2) {
                S4[2 * 64 + lane] = A0;       // NI
                S4[3 * 64 + lane] = A1;       // NH
            } else {
                S4[g * 64 + lane] = A0 + A1;  // R or Z (bias + x product + h product)
            }
        }
        lds_barrier();
        if (wave == 4) epi(12, S4[lane], S4[64 + lane], S4[2 * 64 + lane], S4[3 * 64 + lane]);
        return;
    }
    for (int nb = wave; nb < Nb; nb += kWaves) gru_block(X4, H4, Kb, Be, w, nb, lane, epi);
}

// Backward of the two GRU matmuls: DX = W_ir^T dR + W_iz^T dZ + W_in^T dNI, DH = W_hr^T dR + W_hz^T dZ + W_hn^T dNH
struct GruWT {
    const float *wt_ir, *wt_iz, *wt_in, *wt_hr, *wt_hz, *wt_hn;   // packed transposes (Be, Be)
};

struct GruBwdFrag {
    floatx4 ar, az, ai, ah, bir, biz, bin, bhr, bhz, bhn;
};

// One output column block of the two transposed GRU products.
template <class PF, class Epi>
__device__ __forceinline__ void gru_bwd_block(const floatx4* __restrict__ R4, const floatx4* __restrict__ Z4,
                                              const floatx4* __restrict__ I4, const floatx4* __restrict__ H4, int Kb,
                                              const GruWT& w, int nb, int lane, const PF& pf, Epi&& epi) {
    floatx4 DX = floatx4{0.f, 0.f, 0.f, 0.f}, DH = floatx4{0.f, 0.f, 0.f, 0.f};
    floatx4 DX2 = DX, DH2 = DH;   // second chain per output: consecutive MFMAs stay independent
    const size_t off = (size_t)nb * Kb * 64 + lane;
    const floatx4* __restrict__ Wir = reinterpret_cast<const floatx4*>(w.wt_ir) + off;
    const floatx4* __restrict__ Wiz = reinterpret_cast<const floatx4*>(w.wt_iz) + off;
    const floatx4* __restrict__ Win = reinterpret_cast<const floatx4*>(w.wt_in) + off;
    const floatx4* __restrict__ Whr = reinterpret_cast<const floatx4*>(w.wt_hr) + off;
    const floatx4* __restrict__ Whz = reinterpret_cast<const floatx4*>(w.wt_hz) + off;
    const floatx4* __restrict__ Whn = reinterpret_cast<const floatx4*>(w.wt_hn) + off;
    BD_GRU_PIPE(
        Kb,
        [&](int kb) {
            return GruBwdFrag{R4[kb * 64], Z4[kb * 64], I4[kb * 64], H4[kb * 64], Wir[kb * 64], Wiz[kb * 64],
                              Win[kb * 64], Whr[kb * 64], Whz[kb * 64], Whn[kb * 64]};
        },
        [&](const GruBwdFrag& f) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                DX = mfma16(f.ar[j], f.bir[j], DX);
                DH = mfma16(f.ar[j], f.bhr[j], DH);
                DX2 = mfma16(f.az[j], f.biz[j], DX2);
                DH2 = mfma16(f.az[j], f.bhz[j], DH2);
                DX = mfma16(f.ai[j], f.bin[j], DX);
                DH = mfma16(f.ah[j], f.bhn[j], DH);
            }
        });
    epi(nb, DX + DX2, DH + DH2, pf);
}

struct GruBwdHalfFrag {
    floatx4 ax, ah, bx, bh;
};

// `scratch` (>= 6 x 64 float4 of LDS, optional): as in gru_tile, with 8 waves and 13 column blocks the waves 5, 6, 7 take
// block 12 by gate gradient (d r, d z, d n: one product into DX and one into DH each) and wave 4 sums the three pairs in
// fixed order and runs the block's epilogue.
template <class Pre, class Epi>
__device__ __forceinline__ void gru_tile_bwd(const float* __restrict__ dR, const float* __restrict__ dZ,
                                             const float* __restrict__ dNI, const float* __restrict__ dNH, int Kb,
                                             int Be, const GruWT& w, Pre&& pre, Epi&& epi, float* __restrict__ scratch = nullptr) {
    const int lane = bd_tid() & 63, wave = bd_wave(bd_tid());
    const int Nb = (Be + 15) >> 4;
    const floatx4* __restrict__ R4 = reinterpret_cast<const floatx4*>(dR) + lane;
    const floatx4* __restrict__ Z4 = reinterpret_cast<const floatx4*>(dZ) + lane;
    const floatx4* __restrict__ I4 = reinterpret_cast<const floatx4*>(dNI) + lane;
    const floatx4* __restrict__ H4 = reinterpret_cast<const floatx4*>(dNH) + lane;
    if (kWaves == 8 && Nb == 13 && scratch != nullptr) {        // workgroup-uniform
        decltype(pre(0)) pf12{};
        if (wave == 4) pf12 = pre(12);                             // its epilogue operands: in flight through both rounds
        gru_bwd_block(R4, Z4, I4, H4, Kb, w, wave, lane, pre(wave), epi);            // blocks 0..7
        floatx4* __restrict__ S4 = reinterpret_cast<floatx4*>(scratch);
        if (wave < 4) {
            gru_bwd_block(R4, Z4, I4, H4, Kb, w, wave + 8, lane, pre(wave + 8), epi);   // blocks 8..11
        } else if (wave > 4) {
            const int g = wave - 5;                                // 0: d r, 1: d z, 2: d n
            const size_t off = (size_t)12 * Kb * 64 + lane;
            const floatx4* __restrict__ Ax = g == 0 ? R4 : (g == 1 ? Z4 : I4);
            const floatx4* __restrict__ Ah = g == 0 ? R4 : (g == 1 ? Z4 : H4);
            const floatx4* __restrict__ Wx = reinterpret_cast<const floatx4*>(g == 0 ? w.wt_ir : (g == 1 ? w.wt_iz : w.wt_in)) + off;
            const floatx4* __restrict__ Wh = reinterpret_cast<const floatx4*>(g == 0 ? w.wt_hr : (g == 1 ? w.wt_hz : w.wt_hn)) + off;
            floatx4 DX = floatx4{0.f, 0.f, 0.f, 0.f}, DH = DX;
            pipelined_k<2>(
                Kb, [&](int kb) { return GruBwdHalfFrag{Ax[kb * 64], Ah[kb * 64], Wx[kb * 64], Wh[kb * 64]}; },
                [&](const GruBwdHalfFrag& f) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        DX = mfma16(f.ax[j], f.bx[j], DX);
                        DH = mfma16(f.ah[j], f.bh[j], DH);
                    }
                });
            S4[(2 * g) * 64 + lane] = DX;
            S4[(2 * g + 1) * 64 + lane] = DH;
        }
        lds_barrier();
        if (wave == 4)
            epi(12, S4[lane] + S4[2 * 64 + lane] + S4[4 * 64 + lane], S4[64 + lane] + S4[3 * 64 + lane] + S4[5 * 64 + lane], pf12);
        return;
    }
    for (int nb = wave; nb < Nb; nb += kWaves) gru_bwd_block(R4, Z4, I4, H4, Kb, w, nb, lane, pre(nb), epi);
}

// element offset inside a fragment tile for the accumulator element (row = 4*(lane>>4)+r, col = nb*16+(lane&15))
__device__ __forceinline__ int acc_frag_off(int nb, int lane, int r) {
    const int c = lane & 15, row = 4 * (lane >> 4) + r;
    return nb * kFragFloats + ((c >> 2) * 16 + row) * 4 + (c & 3);
}

}  // namespace bd
