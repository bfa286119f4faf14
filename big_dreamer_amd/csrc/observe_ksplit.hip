// observe_ksplit.hip -- the Gaussian RSSM observe scan on a cluster of C = Be/16 workgroups per 16-row tile, round-3 form:
// every layer of the recurrence is split over the members along K ("column-parallel then row-parallel", alternating), the
// member's weight slices live in REGISTERS for the whole launch, and nothing is computed redundantly.
//
// Why a second cluster form.  observe_cluster.hip (round 1) splits only the GRU by output columns and all-gathers the new
// belief; embed, posterior hidden layer and head are recomputed by all 13 members and every phase re-streams its weights
// from L2.  rocprof / s_memtime (profiles/r02h, tools/stamps.py): 40k cycles per step, 48 % of it the redundant posterior
// layers, 4.9 x the algorithmic MFMA work, 14.1 + 14.4 % of the GPU time of a training step at 1 % of the MFMA peak.
//
// The chain of one step is  s -> x -> h' -> q -> (mean, raw) -> s'.  Member c owns 16-column block c of x, h', q:
//   F1  x_c  = ELU(W_e[block c, :] [s; a] + b)            needs the full state (every member has it)          local
//   F2  gate partials  R, Z, NI, NH [16 x Be] = x_c W_i*[:, block c]^T (+ h_c W_h*[:, block c]^T)     -> REDUCE-SCATTER
//   F3  h'_c = GRU gates of block c                        (the belief never exists in one place: feat is written by blocks)
//   F4  q partial [16 x Hd] = h'_c W_q1h[:, block c]^T                                                 -> REDUCE-SCATTER
//   F5  q_c  = ELU(sum + pre_emb_c + b)
//   F6  (mean, raw) partial [16 x 2S] = q_c W_q2[:, block c]^T                                         -> ALL-REDUCE
//   F7  s' = mean + (softplus(raw) + min_std) eps          every member, elementwise
// Backward, mirrored:  (dm, draw) -> dQ_c (N-split, local) -> dh partial (RS) -> gate gradients of block c ->
// (DX, DH) partials (RS) -> dE_c, carry_c (the belief-gradient carry stays with its member) -> ds partial (all-reduce).
// Three hand-offs per step and direction instead of one, each a few KB per member pair; per member and step 4 + 13 + 4
// (forward) weight fragments of 1 KiB feed 300 MFMAs -- all register-resident, so a phase is: LDS read, MFMAs,
// write-through stores, flag.  Hand-off protocol, flags, sticky error word: bd_cluster.h.  Reductions run in member order:
// results are bit-identical on every member and from launch to launch.
#include "bd_cluster.h"

namespace bd {

#ifdef BD_STAMPS
__device__ unsigned long long g_kstamps[64];
#define BD_KSTAMP(slot)                                                                                  \
    do {                                                                                                 \
        if (blockIdx.x == 0 && threadIdx.x == 0 && t == 5) g_kstamps[slot] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define BD_KSTAMP(slot)
#endif

constexpr int kKsMaxS = 4;        // state blocks (S <= 64)
constexpr int kKsMaxA = 2;        // action blocks (A <= 32)

__device__ __forceinline__ void ks_store4(float* p, floatx4 v) {     // two 8-byte write-through stores
    unsigned long long lo = (unsigned long long)__float_as_uint(v[0]) | ((unsigned long long)__float_as_uint(v[1]) << 32);
    unsigned long long hi = (unsigned long long)__float_as_uint(v[2]) | ((unsigned long long)__float_as_uint(v[3]) << 32);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p) + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ floatx4 ks_load4(const float* p) {
    const unsigned long long lo = ld_sc1_u64(p), hi = ld_sc1_u64(p + 2);
    return floatx4{__uint_as_float((unsigned)lo), __uint_as_float((unsigned)(lo >> 32)), __uint_as_float((unsigned)hi),
                   __uint_as_float((unsigned)(hi >> 32))};
}
// sum over the members src = first, first + stride, ... < C of one 1 KiB accumulator image (this lane's float4), in order
__device__ __forceinline__ floatx4 ks_sum(const float* base, size_t src_stride, int first, int stride, int C, int lane) {
    floatx4 s = floatx4{0.f, 0.f, 0.f, 0.f};
    const float* p = base + lane * 4;
    int src = first;
    for (; src + 3 * stride < C; src += 4 * stride) {          // four members' loads in flight
        const floatx4 v0 = ks_load4(p + (size_t)src * src_stride), v1 = ks_load4(p + (size_t)(src + stride) * src_stride);
        const floatx4 v2 = ks_load4(p + (size_t)(src + 2 * stride) * src_stride), v3 = ks_load4(p + (size_t)(src + 3 * stride) * src_stride);
        s += v0; s += v1; s += v2; s += v3;
    }
    for (; src < C; src += stride) s += ks_load4(p + (size_t)src * src_stride);
    return s;
}
__device__ __forceinline__ floatx4 mfma4(floatx4 a, floatx4 b, floatx4 acc) {
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = mfma16(a[j], b[j], acc);
    return acc;
}
// packed weight fragment (out block nb, in block kb) of a matrix packed with Kb input blocks; zero when `ok` is false
__device__ __forceinline__ floatx4 ks_frag(const float* w, int nb, int Kb, int kb, int lane, bool ok) {
    return ok ? reinterpret_cast<const floatx4*>(w)[((size_t)nb * Kb + kb) * 64 + lane] : floatx4{0.f, 0.f, 0.f, 0.f};
}

// exchange buffers of one tile and parity (floats)
struct KsBuf {
    size_t g, q, s, total;
    __host__ __device__ explicit KsBuf(int C) {
        g = 0;
        q = g + (size_t)C * C * 4 * 256;     // [dest][src][4 accumulators][64 lanes x 4]
        s = q + (size_t)C * C * 256;         // [dest][src][256]
        total = s + (size_t)C * 8 * 256;     // [src][<= 8 (block, mean | raw) pairs][256]
    }
};

// ---- forward ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void observe_kfwd_kernel(bd_observe_fwd_args a_, float* __restrict__ ws, int C, int tiles,
                                                                unsigned spin_limit) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    BD_KARGS(bd_observe_fwd_args, ap);
#define a (*ap)
    const int Kb_h = cdiv(a.Be, 16), Kb_s = cdiv(a.S, 16), Kb_a = cdiv(a.A, 16), Kb_hd = cdiv(a.Hd, 16);
    const int tile = blockIdx.x / C, c = blockIdx.x - tile * C;
    const int row0 = tile * 16, F = a.Be + a.S, Np = Kb_s * 16;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float* hc = smem;                               // own belief block (fragment block)
    float* xc = hc + 256;                           // own embed block
    float* qc = xc + 256;                           // own posterior-hidden block
    float* sf = qc + 256;                           // masked state, fragment tiles
    float* af = sf + Kb_s * kFragFloats;
    float* s_plain = af + Kb_a * kFragFloats;       // [16][S]
    float* red = s_plain + ((16 * a.S + 3) & ~3);   // [8 waves][64][4]
    float* plain = red + kWaves * 256;              // [2][16][Np]

    unsigned* flags = reinterpret_cast<unsigned*>(ws) + tile * kMaxCluster;
    unsigned* err = reinterpret_cast<unsigned*>(ws) + tiles * kMaxCluster;
    const KsBuf kb_(C);
    float* xbase = ws + cluster_ws_header_floats(tiles) + (size_t)tile * 2 * kb_.total;

    // ---- resident weight slices (K block c of every layer) ----
    floatx4 we_s[kKsMaxS], we_a[kKsMaxA];           // embed: output block c (wave 0 uses them)
#pragma unroll
    for (int kb = 0; kb < kKsMaxS; ++kb) we_s[kb] = ks_frag(a.w_embed_s, c, Kb_s, kb, lane, kb < Kb_s);
#pragma unroll
    for (int kb = 0; kb < kKsMaxA; ++kb) we_a[kb] = ks_frag(a.w_embed_a, c, Kb_a, kb, lane, kb < Kb_a);
    floatx4 wg[2][6], wq[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int nbo = wave + kWaves * i;
        const bool ok = nbo < Kb_h;
        wg[i][0] = ks_frag(a.w_ir, nbo, Kb_h, c, lane, ok); wg[i][1] = ks_frag(a.w_iz, nbo, Kb_h, c, lane, ok);
        wg[i][2] = ks_frag(a.w_in, nbo, Kb_h, c, lane, ok); wg[i][3] = ks_frag(a.w_hr, nbo, Kb_h, c, lane, ok);
        wg[i][4] = ks_frag(a.w_hz, nbo, Kb_h, c, lane, ok); wg[i][5] = ks_frag(a.w_hn, nbo, Kb_h, c, lane, ok);
        wq[i] = ks_frag(a.w_q1h, nbo, Kb_h, c, lane, nbo < Kb_hd);
    }
    const int pair_nb = wave >> 1, pair_raw = wave & 1;              // head pair of this wave: (state block, mean | raw)
    const bool has_pair = pair_nb < Kb_s;
    const floatx4 wh = ks_frag(pair_raw ? a.w_q2s : a.w_q2m, pair_nb, Kb_hd, c, lane, has_pair && c < Kb_hd);
    // biases of the reducing wave (wave 0): GRU block c, posterior hidden block c; of the head pair's wave
    const int colc = c * 16 + (lane & 15);
    float br = 0.f, bz = 0.f, bni = 0.f, bnh = 0.f, bq = 0.f, be = 0.f;
    if (colc < a.Be) {
        br = a.b_ih[colc] + a.b_hh[colc];
        bz = a.b_ih[a.Be + colc] + a.b_hh[a.Be + colc];
        bni = a.b_ih[2 * a.Be + colc];
        bnh = a.b_hh[2 * a.Be + colc];
        be = a.b_embed[colc];
    }
    if (colc < a.Hd) bq = a.b_q1[colc];
    float bh = 0.f;
    if (has_pair && pair_nb * 16 + (lane & 15) < a.S) bh = a.b_q2[pair_raw * a.S + pair_nb * 16 + (lane & 15)];

    // ---- initial carries ----
    for (int i = threadIdx.x; i < 256; i += blockDim.x) {
        const int l = i >> 2, r = i & 3;                           // fragment element -> (row, col) of block c
        const int row = l & 15, col = c * 16 + 4 * (l >> 4) + r;
        hc[i] = (row0 + row < a.B && col < a.Be) ? a.init_belief[(size_t)(row0 + row) * a.Be + col] : 0.f;
    }
    for (int i = threadIdx.x; i < 16 * a.S; i += blockDim.x) {
        const int r = i / a.S, k = i - r * a.S;
        s_plain[i] = (row0 + r < a.B) ? a.init_state[(size_t)(row0 + r) * a.S + k] : 0.f;
    }
    lds_barrier();

    const bool lead = (c == 0);
    const floatx4* __restrict__ HC4 = reinterpret_cast<const floatx4*>(hc) + lane;
    const floatx4* __restrict__ XC4 = reinterpret_cast<const floatx4*>(xc) + lane;
    const floatx4* __restrict__ QC4 = reinterpret_cast<const floatx4*>(qc) + lane;
    floatx4* __restrict__ RED4 = reinterpret_cast<floatx4*>(red);

    for (int t = 0; t < a.T; ++t) {
        const size_t tb = (size_t)t * a.B;
        const int tid = bd_tid();
        float* xb = xbase + (size_t)(t & 1) * kb_.total;
        BD_KSTAMP(0);
        BD_KARGS_FRESH(ap);
        // ---- A: masked state / action fragments (every member; K of the embed layer is tiny) ----
        for (int i = tid; i < 16 * Kb_s * 16; i += blockDim.x) {
            const int r = i / (Kb_s * 16), k = i - r * (Kb_s * 16);
            const int grow = row0 + r;
            float v = 0.f;
            if (grow < a.B && k < a.S) {
                v = s_plain[r * a.S + k];
                if (a.nonterm) v *= a.nonterm[tb + grow];
                if (lead && a.sv_s) a.sv_s[(tb + grow) * a.S + k] = v;
            }
            sf[frag_idx(r, k)] = v;
        }
        for (int i = tid; i < 16 * Kb_a * 16; i += blockDim.x) {
            const int r = i / (Kb_a * 16), k = i - r * (Kb_a * 16);
            const int grow = row0 + r;
            af[frag_idx(r, k)] = (grow < a.B && k < a.A) ? a.actions[(tb + grow) * a.A + k] : 0.f;
        }
        // operands of the later epilogues: requested now, consumed after the hand-offs
        float pre[4] = {0.f, 0.f, 0.f, 0.f};
        if (wave == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int grow = row0 + 4 * (lane >> 4) + r;
                if (grow < a.B && colc < a.Hd) pre[r] = a.pre_emb[(tb + grow) * a.Hd + colc];
            }
        }
        float eps = 0.f;
        {
            const int row = tid / a.S, col = tid - row * a.S;
            if (tid < 16 * a.S && row0 + row < a.B) eps = a.eps_post[(tb + row0 + row) * a.S + col];
        }
        lds_barrier();
        BD_KSTAMP(1);
        // ---- F1: x_c = ELU(W_e[block c] [s; a] + b) (wave 0) ----
        if (wave == 0) {
            floatx4 acc = floatx4{be, be, be, be};
            const floatx4* __restrict__ S4 = reinterpret_cast<const floatx4*>(sf) + lane;
            const floatx4* __restrict__ A4 = reinterpret_cast<const floatx4*>(af) + lane;
#pragma unroll
            for (int kb = 0; kb < kKsMaxS; ++kb)
                if (kb < Kb_s) acc = mfma4(S4[kb * 64], we_s[kb], acc);
#pragma unroll
            for (int kb = 0; kb < kKsMaxA; ++kb)
                if (kb < Kb_a) acc = mfma4(A4[kb * 64], we_a[kb], acc);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int grow = row0 + 4 * (lane >> 4) + r;
                const bool ok = grow < a.B && colc < a.Be;
                const float v = ok ? elu(acc[r]) : 0.f;
                xc[acc_frag_off(0, lane, r)] = v;
                if (ok && a.sv_x) a.sv_x[(tb + grow) * a.Be + colc] = v;
            }
        }
        lds_barrier();
        BD_KSTAMP(2);
        // ---- F2: gate partials over K block c, for every output block; reduce-scatter #1 ----
        {
            const floatx4 ax = XC4[0], ah = HC4[0];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int nbo = wave + kWaves * i;
                if (nbo < Kb_h) {
                    const floatx4 z4 = floatx4{0.f, 0.f, 0.f, 0.f};
                    floatx4 R = mfma4(ax, wg[i][0], z4), Z = mfma4(ax, wg[i][1], z4), NI = mfma4(ax, wg[i][2], z4);
                    R = mfma4(ah, wg[i][3], R);
                    Z = mfma4(ah, wg[i][4], Z);
                    const floatx4 NH = mfma4(ah, wg[i][5], z4);
                    float* dst = xb + kb_.g + ((size_t)(nbo * C + c) * 4) * 256 + lane * 4;
                    ks_store4(dst, R); ks_store4(dst + 256, Z); ks_store4(dst + 512, NI); ks_store4(dst + 768, NH);
                }
            }
        }
        publish(flags + c, (unsigned)(3 * t + 1));
        BD_KSTAMP(3);
        wait_all(flags, C, (unsigned)(3 * t + 1), err, spin_limit, kErrFwd);
        BD_KSTAMP(4);
        // ---- F3: sum the C partials of block c (wave = (gate, half of the members)), GRU gates, h'_c ----
        {
            const int g = wave & 3, half = wave >> 2;
            RED4[wave * 64 + lane] = ks_sum(xb + kb_.g + ((size_t)(c * C) * 4 + g) * 256, (size_t)4 * 256, half, 2, C, lane);
        }
        lds_barrier();
        if (wave == 0) {
            const floatx4 R = RED4[0 * 64 + lane] + RED4[4 * 64 + lane], Z = RED4[1 * 64 + lane] + RED4[5 * 64 + lane];
            const floatx4 NI = RED4[2 * 64 + lane] + RED4[6 * 64 + lane], NH = RED4[3 * 64 + lane] + RED4[7 * 64 + lane];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int grow = row0 + 4 * (lane >> 4) + r;
                const int off = acc_frag_off(0, lane, r);
                const float rr = sigmoidf(R[r] + br), zz = sigmoidf(Z[r] + bz);
                const float nh = NH[r] + bnh;
                const float nn = tanh_act(NI[r] + bni + rr * nh);
                const bool ok = grow < a.B && colc < a.Be;
                const float hn = ok ? (1.f - zz) * nn + zz * hc[off] : 0.f;
                hc[off] = hn;
                if (ok) {
                    a.feat[(tb + grow) * F + colc] = hn;
                    if (a.sv_gates) {
                        float* gg = a.sv_gates + (tb + grow) * 4 * a.Be + colc;
                        gg[0] = rr; gg[a.Be] = zz; gg[2 * a.Be] = nn; gg[3 * a.Be] = nh;
                    }
                }
            }
        }
        lds_barrier();
        BD_KSTAMP(5);
        // ---- F4: posterior-hidden partials over K block c; reduce-scatter #2 ----
        {
            const floatx4 ah = HC4[0];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int nbo = wave + kWaves * i;
                if (nbo < Kb_hd) ks_store4(xb + kb_.q + (size_t)(nbo * C + c) * 256 + lane * 4, mfma4(ah, wq[i], floatx4{0.f, 0.f, 0.f, 0.f}));
            }
        }
        publish(flags + c, (unsigned)(3 * t + 2));
        BD_KSTAMP(6);
        wait_all(flags, C, (unsigned)(3 * t + 2), err, spin_limit, kErrFwd);
        BD_KSTAMP(7);
        // ---- F5: q_c = ELU(sum + pre_emb_c + b) ----
        if (c < Kb_hd) RED4[wave * 64 + lane] = ks_sum(xb + kb_.q + (size_t)(c * C) * 256, 256, wave, kWaves, C, lane);
        lds_barrier();
        if (wave == 0) {
            floatx4 q4 = floatx4{0.f, 0.f, 0.f, 0.f};
            if (c < Kb_hd)
                for (int w = 0; w < kWaves; ++w) q4 += RED4[w * 64 + lane];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int grow = row0 + 4 * (lane >> 4) + r;
                const bool ok = grow < a.B && colc < a.Hd;
                const float v = ok ? elu(q4[r] + pre[r] + bq) : 0.f;
                qc[acc_frag_off(0, lane, r)] = v;
                if (ok && a.sv_q) a.sv_q[(tb + grow) * a.Hd + colc] = v;
            }
        }
        lds_barrier();
        BD_KSTAMP(8);
        // ---- F6: (mean, raw) partials over K block c; all-reduce ----
        if (has_pair) ks_store4(xb + kb_.s + (size_t)(c * 8 + wave) * 256 + lane * 4, mfma4(QC4[0], wh, floatx4{0.f, 0.f, 0.f, 0.f}));
        publish(flags + c, (unsigned)(3 * t + 3));
        BD_KSTAMP(9);
        wait_all(flags, C, (unsigned)(3 * t + 3), err, spin_limit, kErrFwd);
        BD_KSTAMP(10);
        // ---- F7: every member sums the head partials, then samples s' elementwise ----
        if (has_pair) {
            const floatx4 v = ks_sum(xb + kb_.s + (size_t)wave * 256, (size_t)8 * 256, 0, 1, C, lane);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                plain[pair_raw * 16 * Np + (4 * (lane >> 4) + r) * Np + pair_nb * 16 + (lane & 15)] = v[r] + bh;
        }
        lds_barrier();
        for (int e = tid; e < 16 * a.S; e += blockDim.x) {
            const int row = e / a.S, col = e - row * a.S;
            const int grow = row0 + row;
            float st = 0.f;
            if (grow < a.B) {
                const float Mn = plain[row * Np + col], Rw = plain[16 * Np + row * Np + col];
                const float ee = e == tid ? eps : a.eps_post[(tb + grow) * a.S + col];
                const float sd = softplusf(Rw) + a.min_std;
                st = Mn + sd * ee;
                if (lead) {
                    const size_t i = (tb + grow) * a.S + col;
                    a.post_mean[i] = Mn;
                    a.post_std[i] = sd;
                    a.feat[(tb + grow) * F + a.Be + col] = st;
                }
            }
            s_plain[row * a.S + col] = st;
        }
        lds_barrier();
        BD_KSTAMP(11);
    }
#undef a
}

// ---- backward --------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void observe_kbwd_kernel(bd_observe_bwd_args a_, float* __restrict__ ws, int C, int tiles,
                                                                unsigned spin_limit) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    BD_KARGS(bd_observe_bwd_args, ap);
#define a (*ap)
    const int Kb_h = cdiv(a.Be, 16), Kb_s = cdiv(a.S, 16), Kb_hd = cdiv(a.Hd, 16);
    const int tile = blockIdx.x / C, c = blockIdx.x - tile * C;
    const int row0 = tile * 16, F = a.Be + a.S;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float* dhc = smem;                              // own belief-gradient carry block (fragment block)
    float* dQc = dhc + 256;
    float* dRc = dQc + 256;
    float* dZc = dRc + 256;
    float* dNIc = dZc + 256;
    float* dNHc = dNIc + 256;
    float* dEc = dNHc + 256;
    float* dM = dEc + 256;                          // Kb_s fragment tiles
    float* dRaw = dM + Kb_s * kFragFloats;
    float* ds_plain = dRaw + Kb_s * kFragFloats;    // [16][S]
    float* red = ds_plain + ((16 * a.S + 3) & ~3);  // [8 waves][64][4]

    unsigned* flags = reinterpret_cast<unsigned*>(ws) + tile * kMaxCluster;
    unsigned* err = reinterpret_cast<unsigned*>(ws) + tiles * kMaxCluster;
    const KsBuf kb_(C);                             // backward uses: g as [dest][src][2][256] (DX, DH), q as [dest][src][256] (dh),
    float* xbase = ws + cluster_ws_header_floats(tiles) + (size_t)tile * 2 * kb_.total;      //   s as [src][Kb_s][256] (ds)

    // ---- resident weight slices ----
    floatx4 w2m[kKsMaxS], w2s[kKsMaxS];             // dQ block c: wt_q2m / wt_q2s (out = Hd, in = S)
#pragma unroll
    for (int kb = 0; kb < kKsMaxS; ++kb) {
        w2m[kb] = ks_frag(a.wt_q2m, c, Kb_s, kb, lane, kb < Kb_s && c < Kb_hd);
        w2s[kb] = ks_frag(a.wt_q2s, c, Kb_s, kb, lane, kb < Kb_s && c < Kb_hd);
    }
    floatx4 w1[2], wg[2][6];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int nbo = wave + kWaves * i;
        const bool ok = nbo < Kb_h;
        w1[i] = ks_frag(a.wt_q1h, nbo, Kb_hd, c, lane, ok && c < Kb_hd);       // (out = Be, in = Hd)
        wg[i][0] = ks_frag(a.wt_ir, nbo, Kb_h, c, lane, ok); wg[i][1] = ks_frag(a.wt_iz, nbo, Kb_h, c, lane, ok);
        wg[i][2] = ks_frag(a.wt_in, nbo, Kb_h, c, lane, ok); wg[i][3] = ks_frag(a.wt_hr, nbo, Kb_h, c, lane, ok);
        wg[i][4] = ks_frag(a.wt_hz, nbo, Kb_h, c, lane, ok); wg[i][5] = ks_frag(a.wt_hn, nbo, Kb_h, c, lane, ok);
    }
    const floatx4 wes = ks_frag(a.wt_embed_s, wave, Kb_h, c, lane, wave < Kb_s);  // (out = S, in = Be): state block `wave`
    const int colc = c * 16 + (lane & 15);

    for (int i = threadIdx.x; i < 256; i += blockDim.x) dhc[i] = 0.f;
    for (int i = threadIdx.x; i < 16 * a.S; i += blockDim.x) ds_plain[i] = 0.f;
    lds_barrier();

    const bool lead = (c == 0);
    floatx4* __restrict__ RED4 = reinterpret_cast<floatx4*>(red);
    unsigned epoch = 0;

    for (int t = a.T - 1; t >= 0; --t) {
        const size_t tb = (size_t)t * a.B;
        const int tid = bd_tid();
        float* xb = xbase + (size_t)(t & 1) * kb_.total;
        BD_KSTAMP(16);
        BD_KARGS_FRESH(ap);
        // ---- B1: through the sample / softplus into (mean, raw) (every member, elementwise) ----
        for (int i = tid; i < 16 * Kb_s * 16; i += blockDim.x) {
            const int r = i / (Kb_s * 16), k = i - r * (Kb_s * 16);
            const int grow = row0 + r;
            float dm = 0.f, dr = 0.f;
            if (grow < a.B && k < a.S) {
                const size_t idx = (tb + grow) * a.S + k;
                const float dst = ds_plain[r * a.S + k] + a.dfeat[(tb + grow) * F + a.Be + k];
                dm = dst + (a.dpost_mean ? a.dpost_mean[idx] : 0.f);
                const float dsd = dst * a.eps_post[idx] + (a.dpost_std ? a.dpost_std[idx] : 0.f);
                dr = dsd * one_minus_exp_neg(a.post_std[idx] - a.min_std);
                if (lead) {
                    a.d_q2_out[(tb + grow) * 2 * a.S + k] = dm;
                    a.d_q2_out[(tb + grow) * 2 * a.S + a.S + k] = dr;
                }
            }
            dM[frag_idx(r, k)] = dm;
            dRaw[frag_idx(r, k)] = dr;
        }
        // saved operands of wave 0's epilogues: requested now
        float svq[4] = {1.f, 1.f, 1.f, 1.f}, svx[4] = {1.f, 1.f, 1.f, 1.f};
        float gr[4], gz[4], gn[4], gh[4], hprev[4], dft[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) gr[r] = gz[r] = gn[r] = gh[r] = hprev[r] = dft[r] = 0.f;
        if (wave == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int grow = row0 + 4 * (lane >> 4) + r;
                if (grow < a.B && colc < a.Hd) svq[r] = a.sv_q[(tb + grow) * a.Hd + colc];
                if (grow < a.B && colc < a.Be) {
                    svx[r] = a.sv_x[(tb + grow) * a.Be + colc];
                    const float* gg = a.sv_gates + (tb + grow) * 4 * a.Be + colc;
                    gr[r] = gg[0]; gz[r] = gg[a.Be]; gn[r] = gg[2 * a.Be]; gh[r] = gg[3 * a.Be];
                    hprev[r] = t > 0 ? a.feat[(tb - a.B + grow) * F + colc] : a.init_belief[(size_t)grow * a.Be + colc];
                    dft[r] = a.dfeat[(tb + grow) * F + colc];
                }
            }
        }
        lds_barrier();
        BD_KSTAMP(17);
        // ---- B2: dQ block c = ([dm | draw] W_q2)[:, block c] * ELU'(q_c) (wave 0) ----
        if (wave == 0) {
            floatx4 acc = floatx4{0.f, 0.f, 0.f, 0.f};
            const floatx4* __restrict__ M4 = reinterpret_cast<const floatx4*>(dM) + lane;
            const floatx4* __restrict__ R4 = reinterpret_cast<const floatx4*>(dRaw) + lane;
#pragma unroll
            for (int kb = 0; kb < kKsMaxS; ++kb)
                if (kb < Kb_s) {
                    acc = mfma4(M4[kb * 64], w2m[kb], acc);
                    acc = mfma4(R4[kb * 64], w2s[kb], acc);
                }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int grow = row0 + 4 * (lane >> 4) + r;
                float v = 0.f;
                if (grow < a.B && colc < a.Hd) {
                    v = acc[r] * elu_grad_from_out(svq[r]);
                    a.d_q1_pre[(tb + grow) * a.Hd + colc] = v;
                }
                dQc[acc_frag_off(0, lane, r)] = v;
            }
        }
        lds_barrier();
        BD_KSTAMP(18);
        // ---- B3: d belief partials over K block c; reduce-scatter #1 ----
        ++epoch;
        {
            const floatx4 aq = reinterpret_cast<const floatx4*>(dQc)[lane];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int nbo = wave + kWaves * i;
                if (nbo < Kb_h) ks_store4(xb + kb_.q + (size_t)(nbo * C + c) * 256 + lane * 4, mfma4(aq, w1[i], floatx4{0.f, 0.f, 0.f, 0.f}));
            }
        }
        publish(flags + c, epoch);
        BD_KSTAMP(19);
        wait_all(flags, C, epoch, err, spin_limit, kErrBwd);
        BD_KSTAMP(20);
        // ---- B4: total d belief of block c, gate gradients ----
        RED4[wave * 64 + lane] = ks_sum(xb + kb_.q + (size_t)(c * C) * 256, 256, wave, kWaves, C, lane);
        lds_barrier();
        if (wave == 0) {
            floatx4 dh4 = floatx4{0.f, 0.f, 0.f, 0.f};
            for (int w = 0; w < kWaves; ++w) dh4 += RED4[w * 64 + lane];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int grow = row0 + 4 * (lane >> 4) + r;
                const int off = acc_frag_off(0, lane, r);
                float vr = 0.f, vz = 0.f, vni = 0.f, vnh = 0.f, carry = 0.f;
                if (grow < a.B && colc < a.Be) {
                    const float dh = dh4[r] + dhc[off] + dft[r];
                    const float rr = gr[r], zz = gz[r], nn = gn[r], hn = gh[r];
                    const float dn = dh * (1.f - zz);
                    const float dz = dh * (hprev[r] - nn);
                    vni = dn * (1.f - nn * nn);
                    vnh = vni * rr;
                    vr = vni * hn * rr * (1.f - rr);
                    vz = dz * zz * (1.f - zz);
                    carry = dh * zz;
                    float* gi = a.d_gi + (tb + grow) * 3 * a.Be + colc;
                    float* ghh = a.d_gh + (tb + grow) * 3 * a.Be + colc;
                    gi[0] = vr; gi[a.Be] = vz; gi[2 * a.Be] = vni;
                    ghh[0] = vr; ghh[a.Be] = vz; ghh[2 * a.Be] = vnh;
                }
                dRc[off] = vr; dZc[off] = vz; dNIc[off] = vni; dNHc[off] = vnh;
                dhc[off] = carry;          // direct path dh * z; B6 adds the W_hh^T term
            }
        }
        lds_barrier();
        BD_KSTAMP(21);
        // ---- B5: (DX, DH) partials over K block c through W_ih^T / W_hh^T; reduce-scatter #2 ----
        ++epoch;
        {
            const floatx4 aR = reinterpret_cast<const floatx4*>(dRc)[lane], aZ = reinterpret_cast<const floatx4*>(dZc)[lane];
            const floatx4 aI = reinterpret_cast<const floatx4*>(dNIc)[lane], aH = reinterpret_cast<const floatx4*>(dNHc)[lane];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int nbo = wave + kWaves * i;
                if (nbo < Kb_h) {
                    const floatx4 z4 = floatx4{0.f, 0.f, 0.f, 0.f};
                    floatx4 DX = mfma4(aR, wg[i][0], z4), DH = mfma4(aR, wg[i][3], z4);
                    DX = mfma4(aZ, wg[i][1], DX); DH = mfma4(aZ, wg[i][4], DH);
                    DX = mfma4(aI, wg[i][2], DX); DH = mfma4(aH, wg[i][5], DH);
                    float* dst = xb + kb_.g + ((size_t)(nbo * C + c) * 2) * 256 + lane * 4;
                    ks_store4(dst, DX); ks_store4(dst + 256, DH);
                }
            }
        }
        publish(flags + c, epoch);
        BD_KSTAMP(22);
        wait_all(flags, C, epoch, err, spin_limit, kErrBwd);
        BD_KSTAMP(23);
        // ---- B6: d embed pre-activation of block c, the carry's W_hh^T term ----
        {
            const int g = wave & 1, quarter = wave >> 1;
            RED4[wave * 64 + lane] = ks_sum(xb + kb_.g + ((size_t)(c * C) * 2 + g) * 256, (size_t)2 * 256, quarter, 4, C, lane);
        }
        lds_barrier();
        if (wave == 0) {
            floatx4 DX = floatx4{0.f, 0.f, 0.f, 0.f}, DH = DX;
            for (int qd = 0; qd < 4; ++qd) { DX += RED4[(2 * qd) * 64 + lane]; DH += RED4[(2 * qd + 1) * 64 + lane]; }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int grow = row0 + 4 * (lane >> 4) + r;
                const int off = acc_frag_off(0, lane, r);
                float de = 0.f;
                if (grow < a.B && colc < a.Be) {
                    de = DX[r] * elu_grad_from_out(svx[r]);
                    a.d_embed_pre[(tb + grow) * a.Be + colc] = de;
                    dhc[off] += DH[r];
                }
                dEc[off] = de;
            }
        }
        lds_barrier();
        BD_KSTAMP(24);
        // ---- B7: d state partials over K block c; all-reduce ----
        ++epoch;
        if (wave < Kb_s)
            ks_store4(xb + kb_.s + (size_t)(c * 8 + wave) * 256 + lane * 4,
                      mfma4(reinterpret_cast<const floatx4*>(dEc)[lane], wes, floatx4{0.f, 0.f, 0.f, 0.f}));
        publish(flags + c, epoch);
        BD_KSTAMP(25);
        wait_all(flags, C, epoch, err, spin_limit, kErrBwd);
        BD_KSTAMP(26);
        // ---- B8: d posterior_state_t (through the nonterminal mask of this step's input) ----
        if (wave < Kb_s) {
            const floatx4 v = ks_sum(xb + kb_.s + (size_t)wave * 256, (size_t)8 * 256, 0, 1, C, lane);
            const int col = wave * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * (lane >> 4) + r, grow = row0 + row;
                if (col < a.S) ds_plain[row * a.S + col] = grow < a.B ? v[r] * (a.nonterm ? a.nonterm[tb + grow] : 1.f) : 0.f;
            }
        }
        lds_barrier();
        BD_KSTAMP(27);
    }
#undef a
}

size_t ksplit_ws_floats_per_tile(int C) { return 2 * KsBuf(C).total; }

int& ksplit_mode() {               // -1: as the environment says (BD_OBS_KSPLIT=0 switches it off), 0: off, 1: on
    static int m = -1;
    return m;
}

bool ksplit_ok(int Be, int S, int A, int Hd, int C) {
    static const char* e = getenv("BD_OBS_KSPLIT");
    if (ksplit_mode() == 0 || (ksplit_mode() < 0 && e && e[0] == '0')) return false;
    return C == cdiv(Be, 16) && C <= 2 * kWaves && cdiv(Hd, 16) <= C && cdiv(S, 16) <= kKsMaxS && cdiv(A, 16) <= kKsMaxA &&
           S <= kHeadMaxN && 16 * S <= kThreads && 2 * cdiv(S, 16) <= kWaves;
}

static size_t ks_lds_fwd(int S, int A) {
    const int Kb_s = cdiv(S, 16), Kb_a = cdiv(A, 16);
    return ((size_t)3 * 256 + (size_t)(Kb_s + Kb_a) * kFragFloats + ((16 * S + 3) & ~3) + kWaves * 256 + 2 * 16 * Kb_s * 16) * sizeof(float);
}
static size_t ks_lds_bwd(int S) {
    const int Kb_s = cdiv(S, 16);
    return ((size_t)7 * 256 + (size_t)2 * Kb_s * kFragFloats + ((16 * S + 3) & ~3) + kWaves * 256) * sizeof(float);
}

int launch_observe_kfwd(const bd_observe_fwd_args* a, float* ws, int C, int tiles, hipStream_t stream) {
    if (allow_big_lds(observe_kfwd_kernel)) return -1;
    const size_t dyn = launch_lds(observe_kfwd_kernel, ks_lds_fwd(a->S, a->A), "bd_observe_forward_cluster");
    if (!dyn) return -1;
    hipLaunchKernelGGL(observe_kfwd_kernel, dim3(tiles * C), dim3(kThreads), dyn, stream, *a, ws, C, tiles, cluster_spin_limit());
    BD_CHECK_LAUNCH("bd_observe_forward_cluster");
    return 0;
}

int launch_observe_kbwd(const bd_observe_bwd_args* a, float* ws, int C, int tiles, hipStream_t stream) {
    if (allow_big_lds(observe_kbwd_kernel)) return -1;
    const size_t dyn = launch_lds(observe_kbwd_kernel, ks_lds_bwd(a->S), "bd_observe_backward_cluster");
    if (!dyn) return -1;
    hipLaunchKernelGGL(observe_kbwd_kernel, dim3(tiles * C), dim3(kThreads), dyn, stream, *a, ws, C, tiles, cluster_spin_limit());
    BD_CHECK_LAUNCH("bd_observe_backward_cluster");
    return 0;
}

}  // namespace bd

#ifdef BD_STAMPS
extern "C" int bd_debug_kstamps(unsigned long long* out64) {
    return hipMemcpyFromSymbol(out64, HIP_SYMBOL(bd::g_kstamps), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : -1;
}
#endif
