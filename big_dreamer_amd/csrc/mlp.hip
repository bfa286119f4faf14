// mlp.hip -- dense chains (DenseModel / build_mlp: src/models.py:365-408, src/utils.py:368-404):
// forward with saved activations, and the dgrad backward producing pre-activation gradients for
// bd_wgrad.  One workgroup = 16*RT rows; the whole chain runs out of LDS, weights stream from L2.
#include "bd_device.h"
#include "bd_host.h"
#include <stdlib.h>

namespace bd {

// ---- forward --------------------------------------------------------------------------------------
template <int RT>
__global__ __launch_bounds__(kThreads) void mlp_fwd_kernel(bd_mlp_fwd_args a, int KbA, int KbB) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int row0 = blockIdx.x * 16 * RT;
    float* cur = smem;                                   // inputs of even layers
    float* nxt = smem + (size_t)RT * KbA * kFragFloats;  // inputs of odd layers
    float* scratch = nxt + (size_t)RT * KbB * kFragFloats;   // split-K partials for narrow layers
    float* xs = scratch + kSplitScratchFloats;               // [16*RT][N0] gathered one-hot columns of layer 0 (gD > 0)
    load_tile_concat<RT>(cur, cdiv(a.w0 + a.w1, 16), row0, a.M, a.in0, a.ld0, a.w0, a.in1, a.ld1, a.w1);
    if (a.gD > 0) {       // layer 0's one-hot input segment: sum of gD rows of the transposed weights per row
        const int N0 = a.layer[0].N, N4 = N0 >> 2;           // N0 % 4 == 0 (host)
        for (int i = bd_tid(); i < 16 * RT * N4; i += blockDim.x) {
            const int row = i / N4, c4 = i - row * N4;
            const int grow = row0 + row;
            floatx4 s = floatx4{0.f, 0.f, 0.f, 0.f};
            if (grow < a.M) {
                const floatx4* __restrict__ W4 = reinterpret_cast<const floatx4*>(a.gWT) + c4;
                const unsigned char* __restrict__ ix = a.gidx + (size_t)grow * a.gD;
                int f = 0;
                for (; f + 8 <= a.gD; f += 8) {
                    floatx4 t[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) t[j] = W4[(size_t)((f + j) * a.gC + ix[f + j]) * N4];
#pragma unroll
                    for (int j = 0; j < 8; ++j) s += t[j];
                }
                for (; f < a.gD; ++f) s += W4[(size_t)(f * a.gC + ix[f]) * N4];
            }
            reinterpret_cast<floatx4*>(xs)[i] = s;
        }
    }
    lds_barrier();
    // Transposed accumulators (bd_device.h, linear_sweep<TR>): a lane holds row lane&15, columns nb*16 + 4*(lane>>4) + r, so
    // the next layer's fragment tile takes one ds_write_b128 per block and the saved activations one 16-byte store where the
    // width and the buffer allow it (4 | N, 16-byte aligned base); the scalar path serves narrow / odd layers.
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    for (int l = 0; l < a.n_layers; ++l) {
        const bd_layer L = a.layer[l];
        const bool last = (l == a.n_layers - 1);
        const int Kb = cdiv(L.K, 16), Nb = cdiv(L.N, 16);
        const bool gather0 = l == 0 && a.gD > 0;
        const bool vec = (L.N & 3) == 0 && al16(L.save) && (!last || ((a.ldo & 3) == 0 && al16(a.out)));
        const Seg seg[1] = {{cur, L.w, Kb}};
        tile_linear_pre<RT, 1, true>(seg, L.bias, L.N, NoPre{}, [&](int rt, int nb, floatx4 acc, NoPreVal) {
            const int ln = bd_tid() & 63, m = ln & 15, col0 = nb * 16 + 4 * (ln >> 4);
            const int grow = row0 + rt * 16 + m;
            if (gather0 && col0 < L.N) acc += *reinterpret_cast<const floatx4*>(xs + (rt * 16 + m) * L.N + col0);   // N0 % 4 == 0
            floatx4 v = acc;
            if (L.act) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = elu(acc[r]);
            }
            if (!last) *reinterpret_cast<floatx4*>(nxt + ((rt * Nb + nb) * 64 + ln) * 4) = v;
            if (grow < a.M && col0 < L.N) {
                if (vec) {
                    if (L.save) *reinterpret_cast<floatx4*>(L.save + (size_t)grow * L.N + col0) = v;
                    if (last) *reinterpret_cast<floatx4*>(a.out + (size_t)grow * a.ldo + col0) = v;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (col0 + r < L.N) {
                            if (L.save) L.save[(size_t)grow * L.N + col0 + r] = v[r];
                            if (last) a.out[(size_t)grow * a.ldo + col0 + r] = v[r];
                        }
                }
            }
        }, scratch);
        lds_barrier();
        float* t = cur; cur = nxt; nxt = t;
    }
}

// ---- backward (dgrad chain) -------------------------------------------------------------------------
template <int RT>
__global__ __launch_bounds__(kThreads) void mlp_bwd_kernel(bd_mlp_bwd_args a, int KbA, int KbB) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int row0 = blockIdx.x * 16 * RT;
    float* cur = smem;
    float* nxt = smem + (size_t)RT * KbA * kFragFloats;
    float* scratch = nxt + (size_t)RT * KbB * kFragFloats;   // split-K partials for narrow layers
    // d(pre-activation) of the last layer
    {
        const bd_layer_bwd L = a.layer[a.n_layers - 1];
        const int Kb = cdiv(L.N, 16), Kp = Kb * 16;
        for (int idx = threadIdx.x; idx < RT * 16 * Kp; idx += blockDim.x) {
            const int r = idx / Kp, k = idx - r * Kp;
            const int grow = row0 + r;
            float v = 0.f;
            if (grow < a.M && k < L.N) {
                v = a.dout[(size_t)grow * a.lddo + k] * a.dout_scale;
                if (L.act) v *= elu_grad_from_out(L.saved[(size_t)grow * L.N + k]);
                if (L.dpre) L.dpre[(size_t)grow * L.N + k] = v;
            }
            cur[(r >> 4) * Kb * kFragFloats + frag_idx(r & 15, k)] = v;
        }
    }
    lds_barrier();
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    for (int l = a.n_layers - 1; l >= 1; --l) {
        const bd_layer_bwd L = a.layer[l];       // contraction over this layer's outputs (N) -> its inputs (K)
        const bd_layer_bwd P = a.layer[l - 1];   // whose outputs those inputs are
        const int Kb = cdiv(L.N, 16), Nb = cdiv(L.K, 16);
        const Seg segs[1] = {{cur, L.wt, Kb}};
        const bool vec = (P.N & 3) == 0 && al16(P.saved) && al16(P.dpre);     // transposed accumulators: 16-byte loads / stores
        tile_linear_pre<RT, 1, true>(
            segs, nullptr, L.K,
            [&](int rt, int nb) {            // saved activations of the previous layer: in flight before the K loop
                Pre4 p;
                const int ln = bd_tid() & 63, col0 = nb * 16 + 4 * (ln >> 4);
                const int grow = row0 + rt * 16 + (ln & 15);
                const bool in = P.act && grow < a.M && col0 < P.N;
                if (in && vec) {
                    const floatx4 t = *reinterpret_cast<const floatx4*>(P.saved + (size_t)grow * P.N + col0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) p.v[r] = t[r];
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) p.v[r] = (in && col0 + r < P.N) ? P.saved[(size_t)grow * P.N + col0 + r] : 1.f;
                }
                return p;
            },
            [&](int rt, int nb, floatx4 acc, const Pre4& p) {
                const int ln = bd_tid() & 63, col0 = nb * 16 + 4 * (ln >> 4);
                const int grow = row0 + rt * 16 + (ln & 15);
                floatx4 v = floatx4{0.f, 0.f, 0.f, 0.f};
                if (grow < a.M && col0 < P.N) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (col0 + r < P.N) v[r] = P.act ? acc[r] * elu_grad_from_out(p.v[r]) : acc[r];
                    if (P.dpre) {
                        if (vec) *reinterpret_cast<floatx4*>(P.dpre + (size_t)grow * P.N + col0) = v;
                        else {
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (col0 + r < P.N) P.dpre[(size_t)grow * P.N + col0 + r] = v[r];
                        }
                    }
                }
                *reinterpret_cast<floatx4*>(nxt + ((rt * Nb + nb) * 64 + ln) * 4) = v;
            },
            scratch);
        lds_barrier();
        float* t = cur; cur = nxt; nxt = t;
    }
    if (a.din0 != nullptr || a.din1 != nullptr) {
        const bd_layer_bwd L = a.layer[0];
        const Seg segs[1] = {{cur, L.wt, cdiv(L.N, 16)}};
        tile_linear_pre<RT, 1, true>(segs, nullptr, L.K, NoPre{}, [&](int rt, int nb, floatx4 acc, NoPreVal) {
            const int ln = bd_tid() & 63, col0 = nb * 16 + 4 * (ln >> 4);
            const int grow = row0 + rt * 16 + (ln & 15);
            if (grow >= a.M) return;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int col = col0 + r;
                float* p = nullptr;
                if (col < a.w0) { if (a.din0) p = a.din0 + (size_t)grow * a.ld0 + col; }
                else if (col < a.w0 + a.w1) { if (a.din1) p = a.din1 + (size_t)grow * a.ld1 + (col - a.w0); }
                if (p) *p = a.accumulate ? *p + acc[r] : acc[r];
            }
        }, scratch);
    }
}

#ifdef BD_STAMPS
__device__ unsigned long long g_tallstamps[64];
#define TALL_STAMP(slot)                                                                                         \
    do {                                                                                                         \
        if (blockIdx.x == 0 && threadIdx.x == 0 && (slot) < 64) g_tallstamps[slot] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define TALL_STAMP(slot)
#endif

// ---- tall form: 16*RT rows per workgroup, ONE in-place LDS image, balanced pairs (bd_device.h: tall_sweep) ----------
// For chains over tens of thousands of rows (the reward / value heads over every imagined transition).  Per layer:
// sweep (accumulators stay in registers) -> barrier -> epilogue overwrites the image -> barrier.
// The sweeps run in the transposed-accumulator form (bd_device.h, linear_sweep<TR>): a lane holds four consecutive
// columns of one row, so a pair's epilogue is 4 x ELU, one ds_write_b128 into the (in-place) fragment image and one
// 16-byte global store, under one row/column predicate per lane.  With the row-per-register accumulator layout the same
// epilogue was 4 conflicting ds_write_b32 + 4 dword stores + per-value predicates: 10k cycles per 200-wide layer and wave
// against 20k for its sweep (s_memtime stamps, tools/tall_stamps.py) -- and on gfx950 a wave's VALU / memory instructions
// do NOT overlap the fp32 MFMAs of the other waves of its SIMD (tools/probes/issue_probe.hip), so every epilogue cycle is
// a matrix-pipe cycle lost.  Element offsets are 32-bit (host-checked), widths are multiples of 4 floats.
template <int RT>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(3, 3))) void mlp_fwd_tall_kernel(bd_mlp_fwd_args a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int row0 = blockIdx.x * 16 * RT;
    float* img = smem;
    TALL_STAMP(0);
    if (tile_pairs_ok(a.in0, a.ld0, a.w0, a.w1 ? a.in1 : nullptr, a.ld1))
        load_tile_concat_pairs<RT>(img, cdiv(a.w0 + a.w1, 16), row0, a.M, a.in0, a.ld0, a.w0, a.in1, a.ld1, a.w1);
    else
        load_tile_concat<RT>(img, cdiv(a.w0 + a.w1, 16), row0, a.M, a.in0, a.ld0, a.w0, a.in1, a.ld1, a.w1);
    TALL_STAMP(1);
    lds_barrier();
    TALL_STAMP(2);
    for (int l = 0; l < a.n_layers; ++l) {
        const bd_layer L = a.layer[l];
        const bool last = (l == a.n_layers - 1);
        const int Kb = cdiv(L.K, 16), Nb = cdiv(L.N, 16);
        const Seg seg[1] = {{img, L.w, Kb}};
        TallAcc<RT> t;
        tall_sweep<RT, 1>(seg, L.bias, L.N, t);
        TALL_STAMP(3 + 4 * l);
        if (!last) lds_barrier();            // every wave has read the layer's input
        TALL_STAMP(4 + 4 * l);
        const int lane = bd_tid() & 63, m = lane & 15, g = lane >> 4;
        const unsigned N = (unsigned)L.N;
        const unsigned g_lane = (unsigned)(row0 + m) * N + 4u * g;           // element offset at rt = 0, nb = 0
        const bool gather0 = l == 0 && a.gD > 0 && !last;
        if (gather0) {
            // layer 0's one-hot input segment: the sum of gD rows of the plain transposed weights per output row, built in
            // the (now free) image in the accumulators' own layout -- lane (row, column group) of block nb holds four
            // columns -- so that the epilogue below reads its 16 bytes from the address it then overwrites
            const int N4 = L.N >> 2;
            for (int i = bd_tid(); i < 16 * RT * N4; i += blockDim.x) {
                const int row = i / N4, c4 = i - row * N4;
                const int grow = row0 + row;
                floatx4 sum = floatx4{0.f, 0.f, 0.f, 0.f};
                if (grow < a.M) {
                    const floatx4* __restrict__ W4 = reinterpret_cast<const floatx4*>(a.gWT) + c4;
                    const unsigned char* __restrict__ ix = a.gidx + (size_t)grow * a.gD;
                    int f = 0;
                    for (; f + 8 <= a.gD; f += 8) {
                        floatx4 tt[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) tt[j] = W4[(size_t)((f + j) * a.gC + ix[f + j]) * N4];
#pragma unroll
                        for (int j = 0; j < 8; ++j) sum += tt[j];
                    }
                    for (; f < a.gD; ++f) sum += W4[(size_t)(f * a.gC + ix[f]) * N4];
                }
                const int rt = row >> 4, nb = c4 >> 2;
                *reinterpret_cast<floatx4*>(img + ((rt * Nb + nb) * 64 + (c4 & 3) * 16 + (row & 15)) * 4) = sum;
            }
            lds_barrier();
        }
        tall_foreach<RT>(L.N, t, [&](int rt, int nb, floatx4 acc) {
            if (gather0 && nb * 16 + 4 * g < L.N) acc += *reinterpret_cast<const floatx4*>(img + ((rt * Nb + nb) * 64 + lane) * 4);
            floatx4 v = acc;
            if (L.act) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = elu(acc[r]);
            }
            if (!last) {      // (columns >= N of the last block are ELU(0 + 0) = 0: bias and packed weights are zero there)
                *reinterpret_cast<floatx4*>(img + ((rt * Nb + nb) * 64 + lane) * 4) = v;
                if (L.save && row0 + rt * 16 + m < a.M && nb * 16 + 4 * g < L.N)
                    *reinterpret_cast<floatx4*>(L.save + (g_lane + (unsigned)(rt * 16) * N + (unsigned)(nb * 16))) = v;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int grow = row0 + rt * 16 + m, col = nb * 16 + 4 * g + r;
                    if (grow < a.M && col < L.N) {
                        if (L.save) L.save[(unsigned)grow * N + (unsigned)col] = v[r];
                        a.out[(unsigned)grow * (unsigned)a.ldo + (unsigned)col] = v[r];
                    }
                }
            }
        });
        TALL_STAMP(5 + 4 * l);
        if (!last) lds_barrier();
        TALL_STAMP(6 + 4 * l);
    }
}

template <int RT>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(3, 3))) void mlp_bwd_tall_kernel(bd_mlp_bwd_args a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int row0 = blockIdx.x * 16 * RT;
    float* img = smem;
    {
        const bd_layer_bwd L = a.layer[a.n_layers - 1];
        const int Kb = cdiv(L.N, 16), Kp = Kb * 16;
        for (int idx = threadIdx.x; idx < RT * 16 * Kp; idx += blockDim.x) {
            const int r = idx / Kp, k = idx - r * Kp;
            const int grow = row0 + r;
            float v = 0.f;
            if (grow < a.M && k < L.N) {
                v = a.dout[(size_t)grow * a.lddo + k] * a.dout_scale;
                if (L.act) v *= elu_grad_from_out(L.saved[(size_t)grow * L.N + k]);
                if (L.dpre) L.dpre[(size_t)grow * L.N + k] = v;
            }
            img[(r >> 4) * Kb * kFragFloats + frag_idx(r & 15, k)] = v;
        }
    }
    lds_barrier();
    for (int l = a.n_layers - 1; l >= 1; --l) {
        const bd_layer_bwd L = a.layer[l];       // contraction over this layer's outputs (N) -> its inputs (K)
        const bd_layer_bwd P = a.layer[l - 1];   // whose outputs those inputs are
        const int Kb = cdiv(L.N, 16), Nb = cdiv(L.K, 16);
        const Seg seg[1] = {{img, L.wt, Kb}};
        TallAcc<RT> t;
        tall_sweep<RT, 1>(seg, nullptr, L.K, t);
        const int lane = bd_tid() & 63, m = lane & 15, g = lane >> 4;
        const unsigned N = (unsigned)P.N;
        const unsigned g_lane = (unsigned)(row0 + m) * N + 4u * g;
        // saved activations of the previous layer: requested before the barrier, consumed after it
        TallAcc<RT> sv;
        tall_fill<RT>(L.K, sv, [&](int rt, int nb) {
            floatx4 p = floatx4{1.f, 1.f, 1.f, 1.f};
            if (P.act && row0 + rt * 16 + m < a.M && nb * 16 + 4 * g < P.N)
                p = *reinterpret_cast<const floatx4*>(P.saved + (g_lane + (unsigned)(rt * 16) * N + (unsigned)(nb * 16)));
            return p;
        });
        lds_barrier();
        tall_foreach2<RT>(L.K, t, sv, [&](int rt, int nb, floatx4 acc, floatx4 p) {
            const bool in = row0 + rt * 16 + m < a.M && nb * 16 + 4 * g < P.N;
            floatx4 v = floatx4{0.f, 0.f, 0.f, 0.f};
            if (in) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = P.act ? acc[r] * elu_grad_from_out(p[r]) : acc[r];
                if (P.dpre) *reinterpret_cast<floatx4*>(P.dpre + (g_lane + (unsigned)(rt * 16) * N + (unsigned)(nb * 16))) = v;
            }
            *reinterpret_cast<floatx4*>(img + ((rt * Nb + nb) * 64 + lane) * 4) = v;
        });
        lds_barrier();
    }
    if (a.din0 != nullptr || a.din1 != nullptr) {
        const bd_layer_bwd L = a.layer[0];
        const bool rows_full = row0 + 16 * RT <= a.M;
        tile_linear<RT>(img, cdiv(L.N, 16), L.wt, nullptr, L.K, [&](int rt, int nb, floatx4 acc) {
            const int ln = bd_tid() & 63;
            const int col = nb * 16 + (ln & 15);
            const unsigned r0 = (unsigned)(row0 + rt * 16 + 4 * (ln >> 4));
            if (rows_full && a.din1 == nullptr && nb * 16 + 16 <= a.w0) {        // the common single-destination case
                float* __restrict__ p = a.din0 + (r0 * (unsigned)a.ld0 + (unsigned)col);
                if (a.accumulate) {
                    float o[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = p[r * (unsigned)a.ld0];
#pragma unroll
                    for (int r = 0; r < 4; ++r) p[r * (unsigned)a.ld0] = o[r] + acc[r];
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) p[r * (unsigned)a.ld0] = acc[r];
                }
                return;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int grow = (int)r0 + r;
                if (grow >= a.M) continue;
                float* p = nullptr;
                if (col < a.w0) { if (a.din0) p = a.din0 + (size_t)grow * a.ld0 + col; }
                else if (col < a.w0 + a.w1) { if (a.din1) p = a.din1 + (size_t)grow * a.ld1 + (col - a.w0); }
                if (p) *p = a.accumulate ? *p + acc[r] : acc[r];
            }
        });
    }
}

constexpr int kTallRT = 3;
constexpr int kTallMinTiles = 512;   // below this the 16-row workgroups fill the chip better

static int tall_mode = -1;   // -1: BD_MLP_TALL from the environment (unset = on), 0 / 1: forced (bd_mlp_set_tall),
                             // 2: on for every M (diagnostics)

static bool tall_enabled() {
    if (tall_mode >= 0) return tall_mode != 0;
    static const char* e = getenv("BD_MLP_TALL");
    return !(e && atoi(e) == 0);
}

template <class K, class Args>
static int launch_tall(K kernel, const char* name, int M, int KbMax, hipStream_t s, const Args& args) {
    const size_t lds = (size_t)kTallRT * KbMax * kFragFloats * sizeof(float);
    hipLaunchKernelGGL(kernel, dim3(cdiv(M, 16 * kTallRT)), dim3(kThreads), lds, s, args);
    BD_CHECK_LAUNCH(name);
    return 0;
}

template <class K, class Args>
static int launch_chain(K kernel, const char* name, int M, int RT, int KbA, int KbB, hipStream_t s, const Args& args,
                        size_t extra_floats = 0) {
    const size_t lds = ((size_t)RT * (KbA + KbB) * kFragFloats + kSplitScratchFloats + extra_floats) * sizeof(float);
    BD_REQUIRE(lds <= (size_t)kMaxLds, "%s: chain needs %zu B of LDS (> %d)", name, lds, kMaxLds);
    if (lds > 64 * 1024 && allow_big_lds(kernel)) return -1;
    hipLaunchKernelGGL(kernel, dim3(cdiv(M, 16 * RT)), dim3(kThreads), lds, s, args, KbA, KbB);
    BD_CHECK_LAUNCH(name);
    return 0;
}

static int pick_rt(int M, int KbA, int KbB, size_t extra_per_rt = 0) {
    const int tiles = cdiv(M, 16);
    int rt = tiles >= 1024 ? 2 : 1;
    static const char* force = getenv("BD_MLP_RT");          // tuning experiments only
    if (force) rt = atoi(force) >= 2 && tiles >= 1024 ? 2 : 1;
    const size_t cap = force ? 80 * 1024 : 64 * 1024;         // 2 workgroups per CU
    while (rt > 1 && ((size_t)rt * ((KbA + KbB) * kFragFloats + extra_per_rt) + kSplitScratchFloats) * sizeof(float) > cap)
        rt >>= 1;
    return rt;
}

}  // namespace bd

extern "C" {

#ifdef BD_STAMPS
int bd_debug_tallstamps(unsigned long long* out64) {
    return hipMemcpyFromSymbol(out64, HIP_SYMBOL(bd::g_tallstamps), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : -1;
}
#endif

int bd_mlp_set_tall(int mode) {
    bd::tall_mode = mode;
    return 0;
}

int bd_mlp_forward(const bd_mlp_fwd_args* a, void* stream) {
    using namespace bd;
    BD_REQUIRE(a && a->M > 0 && a->n_layers >= 1 && a->n_layers <= BD_MAX_LAYERS, "bd_mlp_forward: bad M/n_layers");
    BD_REQUIRE(a->in0 && a->w0 > 0 && a->ld0 >= a->w0, "bd_mlp_forward: bad input 0");
    BD_REQUIRE(a->w1 == 0 || (a->in1 && a->ld1 >= a->w1), "bd_mlp_forward: bad input 1");
    BD_REQUIRE(a->out && a->ldo >= a->layer[a->n_layers - 1].N, "bd_mlp_forward: bad output");
    int KbA = 0, KbB = 0, k = a->w0 + a->w1;
    for (int l = 0; l < a->n_layers; ++l) {
        const bd_layer& L = a->layer[l];
        BD_REQUIRE(L.w && L.N > 0 && L.K == k, "bd_mlp_forward: layer %d has K=%d, expected %d", l, L.K, k);
        int& kb = (l & 1) ? KbB : KbA;
        kb = cdiv(L.K, 16) > kb ? cdiv(L.K, 16) : kb;
        k = L.N;
    }
    if (a->gD > 0)
        BD_REQUIRE(a->gidx && a->gWT && a->gC > 0 && a->gC <= 256 && a->layer[0].N % 4 == 0,
                   "bd_mlp_forward: one-hot segment needs gidx, gWT, 0 < gC <= 256 and N0 %% 4 == 0");
    if (tall_enabled() && cdiv(a->M, 16) >= (tall_mode == 2 ? 1 : kTallMinTiles)) {
        bool ok = !(a->gD > 0 && a->n_layers == 1);                // (the one-hot segment is added in a hidden layer's epilogue)
        size_t widest = (size_t)a->ldo;                            // 32-bit element offsets inside the kernel
        auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
        for (int l = 0; l < a->n_layers; ++l) {
            const bd_layer& L = a->layer[l];
            ok = ok && tall_shape_ok(L.N, kTallRT);
            if (l + 1 < a->n_layers) ok = ok && (L.N & 3) == 0 && al16(L.save);
            widest = (size_t)L.N > widest ? (size_t)L.N : widest;
        }
        ok = ok && (size_t)a->M * widest < ((size_t)1 << 31);
        if (ok) return launch_tall(mlp_fwd_tall_kernel<kTallRT>, "bd_mlp_forward(tall)", a->M, KbA > KbB ? KbA : KbB,
                                   (hipStream_t)stream, *a);
    }
    const size_t xs = a->gD > 0 ? (size_t)16 * a->layer[0].N : 0;
    const int rt = pick_rt(a->M, KbA, KbB, xs);
    if (rt == 2)
        return launch_chain(mlp_fwd_kernel<2>, "bd_mlp_forward", a->M, 2, KbA, KbB, (hipStream_t)stream, *a, 2 * xs);
    return launch_chain(mlp_fwd_kernel<1>, "bd_mlp_forward", a->M, 1, KbA, KbB, (hipStream_t)stream, *a, xs);
}

int bd_mlp_backward(const bd_mlp_bwd_args* a, void* stream) {
    using namespace bd;
    BD_REQUIRE(a && a->M > 0 && a->n_layers >= 1 && a->n_layers <= BD_MAX_LAYERS, "bd_mlp_backward: bad M/n_layers");
    BD_REQUIRE(a->dout && a->lddo >= a->layer[a->n_layers - 1].N, "bd_mlp_backward: bad dout");
    const bool want_din = a->din0 || a->din1;
    // buffer A holds d(out) of layers L-1, L-3, ...; buffer B the others
    int KbA = 0, KbB = 0;
    for (int l = a->n_layers - 1, j = 0; l >= 0; --l, ++j) {
        const bd_layer_bwd& L = a->layer[l];
        BD_REQUIRE(L.N > 0 && L.K > 0, "bd_mlp_backward: layer %d has bad dims", l);
        BD_REQUIRE(!L.act || L.saved, "bd_mlp_backward: layer %d needs its saved output", l);
        BD_REQUIRE(l == 0 ? (!want_din || L.wt) : (L.wt != nullptr), "bd_mlp_backward: layer %d needs packed W^T", l);
        BD_REQUIRE(l == 0 || a->layer[l - 1].N == L.K, "bd_mlp_backward: layer %d K mismatch", l);
        int& kb = (j & 1) ? KbB : KbA;
        kb = cdiv(L.N, 16) > kb ? cdiv(L.N, 16) : kb;
    }
    if (want_din) BD_REQUIRE(a->w0 + a->w1 == a->layer[0].K, "bd_mlp_backward: din widths != K of layer 0");
    if (tall_enabled() && cdiv(a->M, 16) >= (tall_mode == 2 ? 1 : kTallMinTiles)) {
        bool ok = true;
        size_t widest = (size_t)(a->ld0 > a->ld1 ? a->ld0 : a->ld1);
        auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
        for (int l = a->n_layers - 1; l >= 1; --l) {
            const bd_layer_bwd& P = a->layer[l - 1];
            ok = ok && tall_shape_ok(a->layer[l].K, kTallRT) && (P.N & 3) == 0 && al16(P.saved) && al16(P.dpre);
            widest = (size_t)a->layer[l].K > widest ? (size_t)a->layer[l].K : widest;
        }
        ok = ok && (size_t)a->M * widest < ((size_t)1 << 31);
        int kb = KbA > KbB ? KbA : KbB;
        if (want_din) kb = cdiv(a->layer[0].N, 16) > kb ? cdiv(a->layer[0].N, 16) : kb;
        if (ok) return launch_tall(mlp_bwd_tall_kernel<kTallRT>, "bd_mlp_backward(tall)", a->M, kb, (hipStream_t)stream, *a);
    }
    const int rt = pick_rt(a->M, KbA, KbB);
    if (rt == 2) return launch_chain(mlp_bwd_kernel<2>, "bd_mlp_backward", a->M, 2, KbA, KbB, (hipStream_t)stream, *a);
    return launch_chain(mlp_bwd_kernel<1>, "bd_mlp_backward", a->M, 1, KbA, KbB, (hipStream_t)stream, *a);
}

}  // extern "C"
