// chain_ws.hip -- weight-stationary form of the dense chains (DenseModel / build_mlp: src/models.py:365-408) for TALL
// inputs: the reward / value / critic heads over the 34 300 imagined rows of a train step (src/dreamer.py:320-322,370-391).
//
// mlp.hip walks a chain per 16-row tile and re-streams every layer's weights from L2 for every tile: a wave's prefetch
// distance (two K blocks) is shorter than the loaded L2 round trip, so the matrix pipes wait (~40 % of peak, DESIGN.md
// section 7).  Here the roles are swapped inside ONE persistent launch: a workgroup (8 waves, one CU) owns a contiguous
// range of row tiles and processes it in batches of up to four tiles; per batch it walks the layers, and per layer wave w
// loads column blocks w and w + 8 of the weights ONCE into registers (the packed layout [nb][kb][lane] float4 is already
// the MFMA B operand: 13-15 float4 per lane and block; two blocks share every A fragment read from LDS) and sweeps the
// batch's tiles out of LDS two at a time (four independent MFMA chains) -- no weight traffic and no exposed global
// latency inside the sweep; the next layer's block is requested before the layer barrier.  Activations stay in LDS in
// fragment order across layers (two ping-pong images of 4 tiles), exactly as in mlp.hip; saved activations / outputs go
// to HBM from the epilogues.  Narrow layers (N <= 16: the heads' 200 -> 1 output) give one TILE to each wave instead.
// Same arguments, same results (fp32 MFMA, K order unchanged) as bd_mlp_forward / bd_mlp_backward, which dispatch here
// when the form is switched on (BD_CHAIN_WS=1: forward, 2: forward and backward; default off -- see cw_enabled).
#include "bd_device.h"
#include "bd_host.h"
#include <stdlib.h>

namespace bd {

constexpr int kCwWaves = 8;        // 2 per SIMD: 256 VGPRs each, enough for two register-resident column blocks
constexpr int kCwThreads = kCwWaves * 64;
constexpr int kCwR = 4;            // row tiles per batch
constexpr int kCwKb = 15;          // K blocks a layer may have (K <= 240) = fragment blocks per tile image

// Diagnostic (-DBD_STAMPS builds only): s_memtime of wave 0 of workgroup 0 at the phase boundaries of its second batch.
#ifdef BD_STAMPS
__device__ unsigned long long g_cwstamps[64];
#define CW_STAMP(slot)                                                                                           \
    do {                                                                                                         \
        if (blockIdx.x == 0 && threadIdx.x == 0 && b == 1 && (slot) < 64) g_cwstamps[slot] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define CW_STAMP(slot)
#endif

__device__ __forceinline__ float* cw_tile(float* img, int i) { return img + (size_t)i * kCwKb * kFragFloats; }

// this workgroup's tile range and its split into balanced batches of <= kCwR tiles
struct CwRange {
    int t0, nt, nbatch;
    __device__ CwRange(int tiles) {
        const int G = gridDim.x, base = tiles / G, extra = tiles % G, b = blockIdx.x;
        t0 = b * base + (b < extra ? b : extra);
        nt = base + (b < extra ? 1 : 0);
        nbatch = (nt + kCwR - 1) / kCwR;
    }
    __device__ int batch(int b, int done) const { return (nt - done + (nbatch - b) - 1) / (nbatch - b); }
};

// column block `nb` of packed weights (N x K) into registers; kb >= Kb stays zero
__device__ __forceinline__ void cw_load_w(floatx4 (&w)[kCwKb], const float* __restrict__ packed, int nb, int Kb, int lane) {
    const floatx4* __restrict__ W4 = reinterpret_cast<const floatx4*>(packed) + (size_t)nb * Kb * 64 + lane;
#pragma unroll
    for (int kb = 0; kb < kCwKb; ++kb) w[kb] = kb < Kb ? W4[kb * 64] : floatx4{0.f, 0.f, 0.f, 0.f};
}

// NT tiles (1 or 2) against the wave's register-resident column blocks (NB = 1 or 2): every A fragment read from LDS
// feeds NB blocks, every B fragment NT tiles
template <int NT, int NB>
__device__ __forceinline__ void cw_sweep(const float* __restrict__ A0, const float* __restrict__ A1,
                                         const floatx4 (&w0)[kCwKb], const floatx4 (&w1)[kCwKb], int Kb, int lane,
                                         floatx4 (&acc)[2][2]) {
    const floatx4* __restrict__ X0 = reinterpret_cast<const floatx4*>(A0) + lane;
    const floatx4* __restrict__ X1 = reinterpret_cast<const floatx4*>(A1) + lane;
    floatx4 alt = floatx4{0.f, 0.f, 0.f, 0.f};     // NT * NB == 1: second chain so that consecutive MFMAs stay independent
#pragma unroll
    for (int kb = 0; kb < kCwKb; ++kb) {
        if (kb < Kb) {
            const floatx4 x0 = X0[kb * 64];
            floatx4 x1 = x0;
            if constexpr (NT == 2) x1 = X1[kb * 64];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (NT * NB == 1) {
                    if (j & 1) alt = mfma16(x0[j], w0[kb][j], alt);
                    else acc[0][0] = mfma16(x0[j], w0[kb][j], acc[0][0]);
                } else {
                    acc[0][0] = mfma16(x0[j], w0[kb][j], acc[0][0]);
                    if constexpr (NT == 2) acc[0][1] = mfma16(x1[j], w0[kb][j], acc[0][1]);
                    if constexpr (NB == 2) {
                        acc[1][0] = mfma16(x0[j], w1[kb][j], acc[1][0]);
                        if constexpr (NT == 2) acc[1][1] = mfma16(x1[j], w1[kb][j], acc[1][1]);
                    }
                }
            }
        }
    }
    if constexpr (NT * NB == 1) acc[0][0] += alt;
}
// The same with a compile-time K-block count: no branch per K block, so the A fragments of block kb+1 are requested
// before the MFMAs of block kb (the branchy form above exposes one LDS round trip per K block).
template <int NT, int NB, int KB>
__device__ __forceinline__ void cw_sweep_k(const float* __restrict__ A0, const float* __restrict__ A1,
                                           const floatx4 (&w0)[kCwKb], const floatx4 (&w1)[kCwKb], int lane,
                                           floatx4 (&acc)[2][2]) {
    const floatx4* __restrict__ X0 = reinterpret_cast<const floatx4*>(A0) + lane;
    const floatx4* __restrict__ X1 = reinterpret_cast<const floatx4*>(A1) + lane;
    floatx4 x0 = X0[0], x1 = x0;
    if constexpr (NT == 2) x1 = X1[0];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        floatx4 n0 = x0, n1 = x1;
        if (kb + 1 < KB) {
            n0 = X0[(kb + 1) * 64];
            if constexpr (NT == 2) n1 = X1[(kb + 1) * 64];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[0][0] = mfma16(x0[j], w0[kb][j], acc[0][0]);
            if constexpr (NT == 2) acc[0][1] = mfma16(x1[j], w0[kb][j], acc[0][1]);
            if constexpr (NB == 2) {
                acc[1][0] = mfma16(x0[j], w1[kb][j], acc[1][0]);
                if constexpr (NT == 2) acc[1][1] = mfma16(x1[j], w1[kb][j], acc[1][1]);
            }
        }
        x0 = n0;
        x1 = n1;
    }
}

template <int NT, int NB>
__device__ __forceinline__ void cw_sweep_d(const float* A0, const float* A1, const floatx4 (&w0)[kCwKb],
                                           const floatx4 (&w1)[kCwKb], int Kb, int lane, floatx4 (&acc)[2][2]) {
    if constexpr (NT * NB > 1) {
        if (Kb == 13) return cw_sweep_k<NT, NB, 13>(A0, A1, w0, w1, lane, acc);
        if (Kb == 15) return cw_sweep_k<NT, NB, 15>(A0, A1, w0, w1, lane, acc);
    }
    cw_sweep<NT, NB>(A0, A1, w0, w1, Kb, lane, acc);
}

__device__ __forceinline__ void cw_sweep_any(bool two_tiles, bool two_blocks, const float* A0, const float* A1,
                                             const floatx4 (&w0)[kCwKb], const floatx4 (&w1)[kCwKb], int Kb, int lane,
                                             floatx4 (&acc)[2][2]) {
    if (two_tiles) {
        if (two_blocks) cw_sweep_d<2, 2>(A0, A1, w0, w1, Kb, lane, acc);
        else cw_sweep_d<2, 1>(A0, A1, w0, w1, Kb, lane, acc);
    } else {
        if (two_blocks) cw_sweep_d<1, 2>(A0, A1, w0, w1, Kb, lane, acc);
        else cw_sweep_d<1, 1>(A0, A1, w0, w1, Kb, lane, acc);
    }
}

// rows [row0, row0 + 16*rb) x K of a row-major matrix (ld % 2 == 0) into fragment-order tile images: a lane moves 4
// consecutive k of one row (two 8-byte loads, one conflict-free ds_write_b128; rows vary fastest across lanes)
__device__ __forceinline__ void cw_load_rows(float* __restrict__ img, const float* __restrict__ src, int ld, int K, int row0,
                                             int rb, int M, float scale) {
    const int Kb = cdiv(K, 16), K4 = Kb * 4;
    const int total = rb * 16 * K4;
    constexpr int NIT = 8;          // loads of NIT iterations in flight before the first LDS write (kCwR*16*60/512 = 7.5)
    for (int base = 0; base < total; base += NIT * kCwThreads) {
        floatx4 v[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = base + it * kCwThreads + (int)threadIdx.x;
            v[it] = floatx4{0.f, 0.f, 0.f, 0.f};
            if (i < total) {
                const int r = i & 15, rest = i >> 4;
                const int k4 = rest % K4, t = rest / K4;
                const int grow = row0 + t * 16 + r, k = 4 * k4;
                if (grow < M) {
                    const float* p = src + (size_t)grow * ld + k;
                    if (k + 3 < K) {
                        const float2 a = *reinterpret_cast<const float2*>(p), b = *reinterpret_cast<const float2*>(p + 2);
                        v[it] = floatx4{a.x, a.y, b.x, b.y};
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (k + j < K) v[it][j] = p[j];
                    }
                }
            }
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = base + it * kCwThreads + (int)threadIdx.x;
            if (i < total) {
                const int r = i & 15, rest = i >> 4;
                const int k4 = rest % K4, t = rest / K4;
                *reinterpret_cast<floatx4*>(cw_tile(img, t) + (((k4 >> 2) * 64 + (k4 & 3) * 16 + r) << 2)) = v[it] * scale;
            }
        }
    }
}

// One accumulator's epilogue, branch-light: the activation, four LDS writes into the next layer's fragment image (16 bytes
// apart), four global stores per destination (a row stride apart).  nrows = valid rows of the four this lane holds.
__device__ __forceinline__ void cw_finish(const floatx4& acc, bool act, float* __restrict__ lds, float* __restrict__ g0,
                                          size_t st0, float* __restrict__ g1, size_t st1, int nrows) {
    floatx4 v = acc;
    if (act) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = elu(acc[r]);
    }
    if (lds) {
#pragma unroll
        for (int r = 0; r < 4; ++r) lds[4 * r] = v[r];
    }
    if (nrows >= 4) {
        if (g0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) g0[r * st0] = v[r];
        }
        if (g1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) g1[r * st1] = v[r];
        }
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (r < nrows) {
                if (g0) g0[r * st0] = v[r];
                if (g1) g1[r * st1] = v[r];
            }
        }
    }
}

// ---- forward -------------------------------------------------------------------------------------------------------
// which column blocks of an Nb-block layer a wave holds: block `wave` (if < Nb) and block `wave + kCwWaves` (if < Nb)
__device__ __forceinline__ void cw_fetch_blocks(floatx4 (&w0)[kCwKb], floatx4 (&w1)[kCwKb], const float* __restrict__ packed,
                                                int Nb, int Kb, int wave, int, int lane) {
    // K block by K block for both column blocks: vmcnt retires in order, so the sweep can start on K block 0 as soon as
    // its two fragments have landed and then consumes the stream at its arrival rate (block-after-block order made every
    // phase wait for 16 of the 30 loads: ~3 us per layer with all 256 workgroups pulling the same 240 KB through L2)
    if (Nb > 1 && wave < Nb) {
        const bool two = wave + kCwWaves < Nb;
        const floatx4* __restrict__ A4 = reinterpret_cast<const floatx4*>(packed) + (size_t)wave * Kb * 64 + lane;
        const floatx4* __restrict__ B4 = reinterpret_cast<const floatx4*>(packed) + (size_t)(wave + kCwWaves) * Kb * 64 + lane;
#pragma unroll
        for (int kb = 0; kb < kCwKb; ++kb) {
            w0[kb] = kb < Kb ? A4[kb * 64] : floatx4{0.f, 0.f, 0.f, 0.f};
            if (two) w1[kb] = kb < Kb ? B4[kb * 64] : floatx4{0.f, 0.f, 0.f, 0.f};
        }
    }
}

// A single-block layer (N <= 16: the heads' 200 -> 1 output) gives one TILE to each of the first rb waves; its one column
// block is small enough to be fetched inside the sweep (all Kb loads first, then the MFMAs) -- keeping it out of the
// register-resident arrays also keeps those arrays in registers (a third writer demoted them to scratch)
__device__ __forceinline__ floatx4 cw_narrow_tile(const float* __restrict__ A0, const float* __restrict__ packed, int Kb, int lane,
                                                  float bias) {
    const floatx4* __restrict__ X0 = reinterpret_cast<const floatx4*>(A0) + lane;
    const floatx4* __restrict__ W4 = reinterpret_cast<const floatx4*>(packed) + lane;
    floatx4 acc0 = floatx4{bias, bias, bias, bias}, acc1 = floatx4{0.f, 0.f, 0.f, 0.f};
    constexpr int CH = 5;                        // K blocks in flight (20 registers beside the resident arrays)
#pragma unroll
    for (int k0 = 0; k0 < kCwKb; k0 += CH) {
        floatx4 wv[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) wv[i] = (k0 + i < Kb) ? W4[(k0 + i) * 64] : floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            if (k0 + i < Kb) {
                const floatx4 x0 = X0[(k0 + i) * 64];
                acc0 = mfma16(x0[0], wv[i][0], acc0);
                acc1 = mfma16(x0[1], wv[i][1], acc1);
                acc0 = mfma16(x0[2], wv[i][2], acc0);
                acc1 = mfma16(x0[3], wv[i][3], acc1);
            }
        }
    }
    return acc0 + acc1;
}

__global__ __launch_bounds__(kCwThreads) void chain_ws_fwd_kernel(bd_mlp_fwd_args a, int tiles, int dbg) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* const img0 = smem;
    float* const img1 = smem + (size_t)kCwR * kCwKb * kFragFloats;
    const CwRange R(tiles);
    floatx4 w0[kCwKb], w1[kCwKb];
    for (int b = 0, done = 0; b < R.nbatch; ++b) {
        const int rb = R.batch(b, done);
        const int row0 = (R.t0 + done) * 16;
        done += rb;
        CW_STAMP(0);
        if (!(dbg & 2)) cw_load_rows(img0, a.in0, a.ld0, a.w0, row0, rb, a.M, 1.f);
        CW_STAMP(1);
        // the first layer's weights travel beside the input tiles
        cw_fetch_blocks(w0, w1, a.layer[0].w, cdiv(a.layer[0].N, 16), cdiv(a.layer[0].K, 16), wave, rb, lane);
        lds_barrier();
        CW_STAMP(2);
        for (int l = 0; l < a.n_layers; ++l) {
            const bd_layer L = a.layer[l];
            const bool last = (l == a.n_layers - 1);
            const int Kb = cdiv(L.K, 16), Nb = cdiv(L.N, 16);
            float* in = (l & 1) ? img1 : img0;
            float* out = (l & 1) ? img0 : img1;
            bool fetched = false;
            if (Nb > 1) {                   // column blocks per wave, the batch's tiles swept two at a time
                if (wave < Nb) {
                    const bool two_b = wave + kCwWaves < Nb;
                    const int c = lane & 15;
                    const int c0 = wave * 16 + c, c1 = c0 + 16 * kCwWaves;
                    const float bias0 = (L.bias != nullptr && c0 < L.N) ? L.bias[c0] : 0.f;
                    const float bias1 = (L.bias != nullptr && two_b && c1 < L.N) ? L.bias[c1] : 0.f;
                    for (int t = 0; t < rb; t += 2) {
                        const bool two_t = t + 1 < rb;
                        floatx4 acc[2][2];
                        acc[0][0] = acc[0][1] = floatx4{bias0, bias0, bias0, bias0};
                        acc[1][0] = acc[1][1] = floatx4{bias1, bias1, bias1, bias1};
                        CW_STAMP(3 + 8 * l + (t >> 1) * 2);
                        if (!(dbg & 1)) cw_sweep_any(two_t, two_b, cw_tile(in, t), cw_tile(in, two_t ? t + 1 : t), w0, w1, Kb, lane, acc);
                        else acc[0][0] += w0[0] + w1[0] + w0[12] + w1[12];
                        CW_STAMP(4 + 8 * l + (t >> 1) * 2);
                        // The wave's blocks are dead after its last sweep: request the next layer's blocks BEFORE this
                        // sweep's epilogue.  vmcnt retires in order, so issued behind the epilogue's stores the loads
                        // could only be waited for together with every store of the layer.
                        if (t + 2 >= rb && !last) {
                            cw_fetch_blocks(w0, w1, a.layer[l + 1].w, cdiv(a.layer[l + 1].N, 16), cdiv(a.layer[l + 1].K, 16), wave, rb,
                                            lane);
                            fetched = true;
                        }
                        if (dbg & 4) continue;
#pragma unroll
                        for (int bi = 0; bi < 2; ++bi) {
                            if (bi == 1 && !two_b) break;
                            const int nb = wave + bi * kCwWaves, col = nb * 16 + c;
                            const bool col_ok = col < L.N;
#pragma unroll
                            for (int u = 0; u < 2; ++u) {
                                if (u == 1 && !two_t) break;
                                const int grow0 = row0 + (t + u) * 16 + 4 * (lane >> 4);
                                float* lds = last ? nullptr
                                                  : cw_tile(out, t + u) + nb * kFragFloats + ((c >> 2) * 16 + 4 * (lane >> 4)) * 4 + (c & 3);
                                float* g0 = (col_ok && L.save) ? L.save + (size_t)grow0 * L.N + col : nullptr;
                                float* g1 = (col_ok && last) ? a.out + (size_t)grow0 * a.ldo + col : nullptr;
                                cw_finish(acc[bi][u], L.act != 0, lds, g0, (size_t)L.N, g1, (size_t)a.ldo, a.M - grow0);
                            }
                        }
                    }
                }
            } else if (wave < rb) {         // one column block in all: a tile per wave
                const int c = lane & 15;
                const float bias = (L.bias != nullptr && c < L.N) ? L.bias[c] : 0.f;
                const floatx4 accn = cw_narrow_tile(cw_tile(in, wave), L.w, Kb, lane, bias);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 4 * (lane >> 4) + r;
                    const int grow = row0 + wave * 16 + row;
                    const float v = act_apply(L.act, accn[r]);
                    if (!last) cw_tile(out, wave)[((c >> 2) * 16 + row) * 4 + (c & 3)] = v;
                    if (grow < a.M && c < L.N) {
                        if (L.save) L.save[(size_t)grow * L.N + c] = v;
                        if (last) a.out[(size_t)grow * a.ldo + c] = v;
                    }
                }
            }
            if (!last && !fetched)          // (waves that swept nothing in this layer)
                cw_fetch_blocks(w0, w1, a.layer[l + 1].w, cdiv(a.layer[l + 1].N, 16), cdiv(a.layer[l + 1].K, 16), wave, rb, lane);
            CW_STAMP(7 + 8 * l);
            lds_barrier();
            CW_STAMP(8 + 8 * l);
        }
    }
}

// ---- backward (dgrad chain) ----------------------------------------------------------------------------------------
// d pre_{l-1} = (d pre_l W_l) * ELU'(saved_{l-1}) from the last layer down; optionally d in = d pre_0 W_0.
__global__ __launch_bounds__(kCwThreads) void chain_ws_bwd_kernel(bd_mlp_bwd_args a, int tiles) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* const img0 = smem;
    float* const img1 = smem + (size_t)kCwR * kCwKb * kFragFloats;
    const CwRange R(tiles);
    floatx4 w0[kCwKb], w1[kCwKb];
    const int nl = a.n_layers;
    const bool want_din = a.din0 != nullptr || a.din1 != nullptr;
    const int nstage = want_din ? nl : nl - 1;
    for (int b = 0, done = 0; b < R.nbatch; ++b) {
        const int rb = R.batch(b, done);
        const int row0 = (R.t0 + done) * 16;
        done += rb;
        // d(pre-activation) of the last layer into img0
        {
            const bd_layer_bwd L = a.layer[nl - 1];
            const int Kb = cdiv(L.N, 16), Kp = Kb * 16;
            for (int idx = threadIdx.x; idx < rb * 16 * Kp; idx += blockDim.x) {
                const int r = idx & 15, rest = idx >> 4;
                const int k = rest % Kp, t = rest / Kp;
                const int grow = row0 + t * 16 + r;
                float v = 0.f;
                if (grow < a.M && k < L.N) {
                    v = a.dout[(size_t)grow * a.lddo + k] * a.dout_scale;
                    if (L.act) v *= elu_grad_from_out(L.saved[(size_t)grow * L.N + k]);
                    if (L.dpre) L.dpre[(size_t)grow * L.N + k] = v;
                }
                cw_tile(img0, t)[frag_idx(r, k)] = v;
            }
        }
        // stage s = 0 .. nstage-1: contraction through layer l = nl-1-s (its transposed pack: out = K_l, in = N_l);
        // every stage is wider than one block here (K_l >= 17), so the tile-per-wave form is not needed
        if (nstage > 0)
            cw_fetch_blocks(w0, w1, a.layer[nl - 1].wt, cdiv(a.layer[nl - 1].K, 16), cdiv(a.layer[nl - 1].N, 16), wave, 0, lane);
        lds_barrier();
        for (int s = 0; s < nstage; ++s) {
            const int l = nl - 1 - s;
            const bd_layer_bwd L = a.layer[l];
            const int Kb = cdiv(L.N, 16), Nb = cdiv(L.K, 16);      // contraction over this layer's outputs -> its inputs
            float* in = (s & 1) ? img1 : img0;
            float* out = (s & 1) ? img0 : img1;
            bool fetched = false;
            if (wave < Nb) {
                const bool two_b = wave + kCwWaves < Nb;
                const int c = lane & 15;
                for (int t = 0; t < rb; t += 2) {
                    const bool two_t = t + 1 < rb;
                    // epilogue operands (saved activations of the layer below) requested before the contraction
                    float sv[2][2][4];
                    const bool act = l >= 1 && a.layer[l - 1].act;
#pragma unroll
                    for (int bi = 0; bi < 2; ++bi)
#pragma unroll
                        for (int u = 0; u < 2; ++u)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int col = (wave + bi * kCwWaves) * 16 + c;
                                const int grow = row0 + (t + u) * 16 + 4 * (lane >> 4) + r;
                                const bool ok = act && (bi == 0 || two_b) && (u == 0 || two_t) && grow < a.M && col < L.K;
                                sv[bi][u][r] = ok ? a.layer[l - 1].saved[(size_t)grow * L.K + col] : 1.f;
                            }
                    floatx4 acc[2][2];
                    acc[0][0] = acc[0][1] = acc[1][0] = acc[1][1] = floatx4{0.f, 0.f, 0.f, 0.f};
                    cw_sweep_any(two_t, two_b, cw_tile(in, t), cw_tile(in, two_t ? t + 1 : t), w0, w1, Kb, lane, acc);
                    if (t + 2 >= rb && s + 1 < nstage) {        // next stage's blocks ahead of this epilogue's stores
                        cw_fetch_blocks(w0, w1, a.layer[l - 1].wt, cdiv(a.layer[l - 1].K, 16), cdiv(a.layer[l - 1].N, 16), wave, 0,
                                        lane);
                        fetched = true;
                    }
#pragma unroll
                    for (int bi = 0; bi < 2; ++bi) {
                        if (bi == 1 && !two_b) break;
                        const int nb = wave + bi * kCwWaves, col = nb * 16 + c;
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            if (u == 1 && !two_t) break;
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int row = 4 * (lane >> 4) + r;
                                const int grow = row0 + (t + u) * 16 + row;
                                if (l >= 1) {
                                    const bd_layer_bwd P = a.layer[l - 1];
                                    float v = 0.f;
                                    if (grow < a.M && col < L.K) {
                                        v = acc[bi][u][r];
                                        if (P.act) v *= elu_grad_from_out(sv[bi][u][r]);
                                        if (P.dpre) P.dpre[(size_t)grow * L.K + col] = v;
                                    }
                                    cw_tile(out, t + u)[nb * kFragFloats + ((c >> 2) * 16 + row) * 4 + (c & 3)] = v;
                                } else if (grow < a.M) {            // d in = d pre_0 W_0, split like the forward's inputs
                                    float* p = nullptr;
                                    if (col < a.w0) { if (a.din0) p = a.din0 + (size_t)grow * a.ld0 + col; }
                                    else if (col < a.w0 + a.w1) { if (a.din1) p = a.din1 + (size_t)grow * a.ld1 + (col - a.w0); }
                                    if (p) *p = a.accumulate ? *p + acc[bi][u][r] : acc[bi][u][r];
                                }
                            }
                        }
                    }
                }
            }
            if (s + 1 < nstage && !fetched)
                cw_fetch_blocks(w0, w1, a.layer[l - 1].wt, cdiv(a.layer[l - 1].K, 16), cdiv(a.layer[l - 1].N, 16), wave, 0, lane);
            lds_barrier();
        }
    }
}

// -1: BD_CHAIN_WS from the environment, 0 / 1 / 2: forced (bd_chain_ws_set_mode).  DEFAULT OFF: measured on MI355X
// (tools/chain_probe.py, tools/cw_stamps.py) the forward form runs the 34 300-row head chain in
// 178 us against 194 us for the per-tile kernel when alone on the GPU, but inside the three-stream train step it changes
// nothing (3.46 vs 3.48 ms/step): one 120 KiB workgroup per CU keeps the other streams' workgroups off the chip.
static int cw_mode = -1;

static bool cw_enabled() {
    if (cw_mode >= 0) return cw_mode != 0;
    static const char* e = getenv("BD_CHAIN_WS");
    return e && (e[0] == '1' || e[0] == '2');
}

static int cw_grid(int tiles) {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return 0;
        cus = p.multiProcessorCount;
    }
    return tiles < cus ? tiles : cus;
}

constexpr size_t kCwLds = (size_t)2 * kCwR * kCwKb * kFragFloats * sizeof(float);     // 120 KiB

// Tall chains only: below ~64 rows per CU the per-tile kernel's finer granularity wins.
static bool cw_rows_ok(int M) { return M >= 16384; }

bool chain_ws_forward_ok(const bd_mlp_fwd_args* a) {
    if (!cw_enabled() || !cw_rows_ok(a->M) || a->w1 != 0 || a->gD > 0 || (a->ld0 & 1) || ((uintptr_t)a->in0 & 7)) return false;
    for (int l = 0; l < a->n_layers; ++l)
        if (a->layer[l].K > 16 * kCwKb || a->layer[l].N > 16 * 2 * kCwWaves) return false;
    return true;
}

int chain_ws_forward(const bd_mlp_fwd_args* a, hipStream_t s) {
    const int tiles = cdiv(a->M, 16), grid = cw_grid(tiles);
    if (!grid) return fail("bd_mlp_forward(ws): no device");
    static bool ok = false;
    if (!ok) {
        if (allow_big_lds(chain_ws_fwd_kernel)) return -1;
        ok = true;
    }
    static const char* dbg = getenv("BD_CW_DBG");            // diagnostic: 1 no MFMA sweeps, 2 no input load, 4 no epilogue
    hipLaunchKernelGGL(chain_ws_fwd_kernel, dim3(grid), dim3(kCwThreads), kCwLds, s, *a, tiles, dbg ? atoi(dbg) : 0);
    BD_CHECK_LAUNCH("bd_mlp_forward(ws)");
    return 0;
}

bool chain_ws_backward_ok(const bd_mlp_bwd_args* a) {
    // measured (tools/chain_probe.py, 34 300 rows): the backward form is SLOWER than the per-tile kernel (203 vs 176 us
    // with d/d features, 182 vs 134 us in the critic's form) -- it stays off unless forced (mode 2 / BD_CHAIN_WS=2)
    static const char* e = getenv("BD_CHAIN_WS");
    const bool bwd_on = cw_mode >= 0 ? cw_mode >= 2 : (e && e[0] == '2');
    if (!bwd_on || !cw_rows_ok(a->M)) return false;
    for (int l = 0; l < a->n_layers; ++l)
        if (a->layer[l].N > 16 * kCwKb || a->layer[l].K > 16 * 2 * kCwWaves || a->layer[l].K <= 16) return false;
    return true;
}

int chain_ws_backward(const bd_mlp_bwd_args* a, hipStream_t s) {
    const int tiles = cdiv(a->M, 16), grid = cw_grid(tiles);
    if (!grid) return fail("bd_mlp_backward(ws): no device");
    static bool ok = false;
    if (!ok) {
        if (allow_big_lds(chain_ws_bwd_kernel)) return -1;
        ok = true;
    }
    hipLaunchKernelGGL(chain_ws_bwd_kernel, dim3(grid), dim3(kCwThreads), kCwLds, s, *a, tiles);
    BD_CHECK_LAUNCH("bd_mlp_backward(ws)");
    return 0;
}

}  // namespace bd

#ifdef BD_STAMPS
extern "C" int bd_debug_cwstamps(unsigned long long* out64) {
    return hipMemcpyFromSymbol(out64, HIP_SYMBOL(bd::g_cwstamps), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int bd_chain_ws_set_mode(int mode) {
    bd::cw_mode = mode;
    return 0;
}
