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

// one 16-byte write-through (sc1) store: the hand-off tables of MI355X_MICROARCH.md list 4-, 8- and 16-byte sc1 stores as
// equivalent for visibility, and the wide form is one instruction (and one fabric write) instead of two.  Stores need no
// result register, so the instruction is safe as inline asm: publish() drains vmcnt explicitly before the flag.
// The stored value is an MFMA accumulator: the hazard recogniser does not see an inline-asm READ of a register an MFMA is
// still writing (v_mfma_f32_16x16x4_f32 = 8 passes: 11 wait states before a VMEM read of its result; without them one
// component of the float4 left stale -- every fourth belief column wrong, tools/ks_debug.py), so the wait states are spelled out.
__device__ __forceinline__ void ks_store4(float* p, floatx4 v) {
    asm volatile("s_nop 7\n\ts_nop 3\n\tglobal_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ floatx4 ks_load4(const float* p) {
    const unsigned long long lo = ld_sc1_u64(p), hi = ld_sc1_u64(p + 2);
    return floatx4{__uint_as_float((unsigned)lo), __uint_as_float((unsigned)(lo >> 32)), __uint_as_float((unsigned)hi),
                   __uint_as_float((unsigned)(hi >> 32))};
}
// sum over the members src = first, first + stride, ... < C of one 1 KiB accumulator image (this lane's float4), in member
// order.  All loads are issued before the first add (up to 16 members: C <= kMaxCluster): one L2 round trip, not one per batch.
__device__ __forceinline__ floatx4 ks_sum(const float* base, size_t src_stride, int first, int stride, int C, int lane) {
    const float* p = base + lane * 4;
    floatx4 v[kMaxCluster];
#pragma unroll
    for (int i = 0; i < kMaxCluster; ++i) {
        const int src = first + i * stride;
        v[i] = floatx4{0.f, 0.f, 0.f, 0.f};
        if (src < C) v[i] = ks_load4(p + (size_t)src * src_stride);
    }
    floatx4 s = v[0];
#pragma unroll
    for (int i = 1; i < kMaxCluster; ++i) s += v[i];
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

// Register layout of every activation block in these kernels ("fragment layout"): element i of a lane's float4 is
// (row = lane & 15, column = 16 c + 4 (lane >> 4) + i).  Products are formed TRANSPOSED -- D^T = W X^T, the packed weight
// fragment as the MFMA's A operand and the activation float4 as its B operand (bd_device.h, linear_sweep<TR>) -- so an
// accumulator is already the next layer's operand: x_c, h'_c, q_c (and their gradients) never touch LDS, and every wave
// finishes every reduction itself (the eight waves hold identical copies), which leaves ONE workgroup barrier per reduction.
__device__ __forceinline__ floatx4 mfmaT(floatx4 w, floatx4 x, floatx4 acc) {
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = mfma16(w[j], x[j], acc);
    return acc;
}
// four consecutive floats of a row-major row (16-byte load when the row start allows it)
__device__ __forceinline__ floatx4 ks_row4(const float* p, int n_valid) {
    floatx4 v = floatx4{0.f, 0.f, 0.f, 0.f};
    if (n_valid >= 4 && (reinterpret_cast<uintptr_t>(p) & 15) == 0) return *reinterpret_cast<const floatx4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (i < n_valid) v[i] = p[i];
    return v;
}
__device__ __forceinline__ void ks_put4(float* p, floatx4 v, int n_valid) {
    if (n_valid >= 4 && (reinterpret_cast<uintptr_t>(p) & 15) == 0) {
        *reinterpret_cast<floatx4*>(p) = v;
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (i < n_valid) p[i] = v[i];
}

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
    float* sf = smem;                               // masked state, fragment tiles
    float* af = sf + Kb_s * kFragFloats;
    float* s_plain = af + Kb_a * kFragFloats;       // [16][S]
    float* red = s_plain + ((16 * a.S + 3) & ~3);   // [8 waves][64][4]
    float* plain = red + kWaves * 256;              // [2][16][Np]

    unsigned* flags = reinterpret_cast<unsigned*>(ws) + tile * kMaxCluster;
    unsigned* err = reinterpret_cast<unsigned*>(ws) + tiles * kMaxCluster;
    const KsBuf kb_(C);
    float* xbase = ws + cluster_ws_header_floats(tiles) + (size_t)tile * 2 * kb_.total;

    // ---- resident weight slices (K block c of every layer) ----
    floatx4 we_s[kKsMaxS], we_a[kKsMaxA];           // embed: output block c
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
    // this lane's elements: row frow, columns fcol0 .. fcol0 + 3 of block c
    const int frow = lane & 15, fcol0 = c * 16 + 4 * (lane >> 4);
    const int grow = row0 + frow;
    const int nbe = a.Be - fcol0 < 0 ? 0 : (a.Be - fcol0 < 4 ? a.Be - fcol0 : 4);      // valid belief columns of this lane
    const int nhd = a.Hd - fcol0 < 0 ? 0 : (a.Hd - fcol0 < 4 ? a.Hd - fcol0 : 4);
    const bool rok = grow < a.B;
    floatx4 br4 = floatx4{0.f, 0.f, 0.f, 0.f}, bz4 = br4, bni4 = br4, bnh4 = br4, be4 = br4, bq4 = br4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i < nbe) {
            const int col = fcol0 + i;
            br4[i] = a.b_ih[col] + a.b_hh[col];
            bz4[i] = a.b_ih[a.Be + col] + a.b_hh[a.Be + col];
            bni4[i] = a.b_ih[2 * a.Be + col];
            bnh4[i] = a.b_hh[2 * a.Be + col];
            be4[i] = a.b_embed[col];
        }
        if (i < nhd) bq4[i] = a.b_q1[fcol0 + i];
    }
    floatx4 bh4 = floatx4{0.f, 0.f, 0.f, 0.f};
    if (has_pair) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int col = pair_nb * 16 + 4 * (lane >> 4) + i;
            if (col < a.S) bh4[i] = a.b_q2[pair_raw * a.S + col];
        }
    }
    // ---- initial carries ----
    floatx4 h4 = rok ? ks_row4(a.init_belief + (size_t)grow * a.Be + fcol0, nbe) : floatx4{0.f, 0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < 16 * a.S; i += blockDim.x) {
        const int r = i / a.S, k = i - r * a.S;
        s_plain[i] = (row0 + r < a.B) ? a.init_state[(size_t)(row0 + r) * a.S + k] : 0.f;
    }
    lds_barrier();

    const bool lead = (c == 0);
    floatx4* __restrict__ RED4 = reinterpret_cast<floatx4*>(red);
    const floatx4 z4 = floatx4{0.f, 0.f, 0.f, 0.f};

    for (int t = 0; t < a.T; ++t) {
        const size_t tb = (size_t)t * a.B;
        const int tid = bd_tid();
        float* xb = xbase + (size_t)(t & 1) * kb_.total;
        BD_KSTAMP(0);
        BD_KARGS_FRESH(ap);
        // ---- A: masked state / action fragments (every member; K of the embed layer is tiny) ----
        for (int i = tid; i < 16 * Kb_s * 16; i += blockDim.x) {
            const int r = i / (Kb_s * 16), k = i - r * (Kb_s * 16);
            const int gr = row0 + r;
            float v = 0.f;
            if (gr < a.B && k < a.S) {
                v = s_plain[r * a.S + k];
                if (a.nonterm) v *= a.nonterm[tb + gr];
                if (lead && a.sv_s) a.sv_s[(tb + gr) * a.S + k] = v;
            }
            sf[frag_idx(r, k)] = v;
        }
        for (int i = tid; i < 16 * Kb_a * 16; i += blockDim.x) {
            const int r = i / (Kb_a * 16), k = i - r * (Kb_a * 16);
            const int gr = row0 + r;
            af[frag_idx(r, k)] = (gr < a.B && k < a.A) ? a.actions[(tb + gr) * a.A + k] : 0.f;
        }
        // operands of the later epilogues: requested now, consumed after the hand-offs
        const floatx4 pre4 = rok ? ks_row4(a.pre_emb + (tb + grow) * a.Hd + fcol0, nhd) : z4;
        float eps = 0.f;
        {
            const int row = tid / a.S, col = tid - row * a.S;
            if (tid < 16 * a.S && row0 + row < a.B) eps = a.eps_post[(tb + row0 + row) * a.S + col];
        }
        lds_barrier();
        BD_KSTAMP(1);
        // ---- F1: x_c = ELU(W_e[block c] [s; a] + b): every wave, in registers ----
        floatx4 x4;
        {
            floatx4 acc = be4;
            const floatx4* __restrict__ S4 = reinterpret_cast<const floatx4*>(sf) + lane;
            const floatx4* __restrict__ A4 = reinterpret_cast<const floatx4*>(af) + lane;
#pragma unroll
            for (int kb = 0; kb < kKsMaxS; ++kb)
                if (kb < Kb_s) acc = mfmaT(we_s[kb], S4[kb * 64], acc);
#pragma unroll
            for (int kb = 0; kb < kKsMaxA; ++kb)
                if (kb < Kb_a) acc = mfmaT(we_a[kb], A4[kb * 64], acc);
#pragma unroll
            for (int i = 0; i < 4; ++i) x4[i] = (rok && i < nbe) ? elu(acc[i]) : 0.f;
            if (wave == 0 && rok && a.sv_x) ks_put4(a.sv_x + (tb + grow) * a.Be + fcol0, x4, nbe);
        }
        BD_KSTAMP(2);
        BD_KARGS_FRESH(ap);
        // ---- F2: gate partials over K block c, for every output block; reduce-scatter #1 ----
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int nbo = wave + kWaves * i;
            if (nbo < Kb_h) {
                floatx4 R = mfmaT(wg[i][0], x4, z4), Z = mfmaT(wg[i][1], x4, z4);
                const floatx4 NI = mfmaT(wg[i][2], x4, z4);
                R = mfmaT(wg[i][3], h4, R);
                Z = mfmaT(wg[i][4], h4, Z);
                const floatx4 NH = mfmaT(wg[i][5], h4, z4);
                float* dst = xb + kb_.g + ((size_t)(nbo * C + c) * 4) * 256 + lane * 4;
                ks_store4(dst, R); ks_store4(dst + 256, Z); ks_store4(dst + 512, NI); ks_store4(dst + 768, NH);
            }
        }
        publish(flags + c, (unsigned)(3 * t + 1));
        BD_KSTAMP(3);
        wait_all(flags, C, (unsigned)(3 * t + 1), err, spin_limit, kErrFwd);
        BD_KSTAMP(4);
        // ---- F3: sum the C partials of block c (wave = (gate, half of the members)); every wave finishes the GRU gates ----
        {
            const int g = wave & 3, half = wave >> 2;
            RED4[wave * 64 + lane] = ks_sum(xb + kb_.g + ((size_t)(c * C) * 4 + g) * 256, (size_t)4 * 256, half, 2, C, lane);
        }
        lds_barrier();
        {
            const floatx4 R = RED4[0 * 64 + lane] + RED4[4 * 64 + lane] + br4, Z = RED4[1 * 64 + lane] + RED4[5 * 64 + lane] + bz4;
            const floatx4 NI = RED4[2 * 64 + lane] + RED4[6 * 64 + lane] + bni4, NH = RED4[3 * 64 + lane] + RED4[7 * 64 + lane] + bnh4;
            floatx4 rr4, zz4, nn4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                rr4[i] = sigmoidf(R[i]);
                zz4[i] = sigmoidf(Z[i]);
                nn4[i] = tanh_act(NI[i] + rr4[i] * NH[i]);
                h4[i] = (rok && i < nbe) ? (1.f - zz4[i]) * nn4[i] + zz4[i] * h4[i] : 0.f;
            }
            if (wave == 0 && rok) {
                ks_put4(a.feat + (tb + grow) * F + fcol0, h4, nbe);
                if (a.sv_gates) {
                    float* gg = a.sv_gates + (tb + grow) * 4 * a.Be + fcol0;
                    ks_put4(gg, rr4, nbe); ks_put4(gg + a.Be, zz4, nbe); ks_put4(gg + 2 * a.Be, nn4, nbe); ks_put4(gg + 3 * a.Be, NH, nbe);
                }
            }
        }
        BD_KSTAMP(5);
        BD_KARGS_FRESH(ap);
        // ---- F4: posterior-hidden partials over K block c; reduce-scatter #2 ----
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int nbo = wave + kWaves * i;
            if (nbo < Kb_hd) ks_store4(xb + kb_.q + (size_t)(nbo * C + c) * 256 + lane * 4, mfmaT(wq[i], h4, z4));
        }
        publish(flags + c, (unsigned)(3 * t + 2));       // (its barrier also orders the RED reads above before the next writes)
        BD_KSTAMP(6);
        wait_all(flags, C, (unsigned)(3 * t + 2), err, spin_limit, kErrFwd);
        BD_KSTAMP(7);
        // ---- F5: q_c = ELU(sum + pre_emb_c + b): every wave ----
        RED4[wave * 64 + lane] = c < Kb_hd ? ks_sum(xb + kb_.q + (size_t)(c * C) * 256, 256, wave, kWaves, C, lane) : z4;
        lds_barrier();
        floatx4 q4 = bq4 + pre4;
        for (int w = 0; w < kWaves; ++w) q4 += RED4[w * 64 + lane];
#pragma unroll
        for (int i = 0; i < 4; ++i) q4[i] = (rok && i < nhd) ? elu(q4[i]) : 0.f;
        if (wave == 0 && rok && a.sv_q) ks_put4(a.sv_q + (tb + grow) * a.Hd + fcol0, q4, nhd);
        BD_KSTAMP(8);
        BD_KARGS_FRESH(ap);
        // ---- F6: (mean, raw) partials over K block c; all-reduce ----
        if (has_pair) ks_store4(xb + kb_.s + (size_t)(c * 8 + wave) * 256 + lane * 4, mfmaT(wh, q4, z4));
        publish(flags + c, (unsigned)(3 * t + 3));
        BD_KSTAMP(9);
        wait_all(flags, C, (unsigned)(3 * t + 3), err, spin_limit, kErrFwd);
        BD_KSTAMP(10);
        BD_KARGS_FRESH(ap);
        // ---- F7: every member sums the head partials, then samples s' elementwise ----
        if (has_pair) {
            const floatx4 v = ks_sum(xb + kb_.s + (size_t)wave * 256, (size_t)8 * 256, 0, 1, C, lane) + bh4;
#pragma unroll
            for (int i = 0; i < 4; ++i) plain[pair_raw * 16 * Np + frow * Np + pair_nb * 16 + 4 * (lane >> 4) + i] = v[i];
        }
        lds_barrier();
        for (int e = tid; e < 16 * a.S; e += blockDim.x) {
            const int row = e / a.S, col = e - row * a.S;
            const int gr = row0 + row;
            float st = 0.f;
            if (gr < a.B) {
                const float Mn = plain[row * Np + col], Rw = plain[16 * Np + row * Np + col];
                const float ee = e == tid ? eps : a.eps_post[(tb + gr) * a.S + col];
                const float sd = softplusf(Rw) + a.min_std;
                st = Mn + sd * ee;
                if (lead) {
                    const size_t ix = (tb + gr) * a.S + col;
                    a.post_mean[ix] = Mn;
                    a.post_std[ix] = sd;
                    a.feat[(tb + gr) * F + a.Be + col] = st;
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
    float* dM = smem;                               // Kb_s fragment tiles
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
    const int frow = lane & 15, fcol0 = c * 16 + 4 * (lane >> 4);
    const int grow = row0 + frow;
    const int nbe = a.Be - fcol0 < 0 ? 0 : (a.Be - fcol0 < 4 ? a.Be - fcol0 : 4);
    const int nhd = a.Hd - fcol0 < 0 ? 0 : (a.Hd - fcol0 < 4 ? a.Hd - fcol0 : 4);
    const bool rok = grow < a.B;

    for (int i = threadIdx.x; i < 16 * a.S; i += blockDim.x) ds_plain[i] = 0.f;
    lds_barrier();

    const bool lead = (c == 0);
    floatx4* __restrict__ RED4 = reinterpret_cast<floatx4*>(red);
    const floatx4 z4 = floatx4{0.f, 0.f, 0.f, 0.f};
    floatx4 dhc4 = z4;                               // belief-gradient carry of block c (every wave holds a copy)
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
            const int gr = row0 + r;
            float dm = 0.f, dr = 0.f;
            if (gr < a.B && k < a.S) {
                const size_t idx = (tb + gr) * a.S + k;
                const float dst = ds_plain[r * a.S + k] + a.dfeat[(tb + gr) * F + a.Be + k];
                dm = dst + (a.dpost_mean ? a.dpost_mean[idx] : 0.f);
                const float dsd = dst * a.eps_post[idx] + (a.dpost_std ? a.dpost_std[idx] : 0.f);
                dr = dsd * one_minus_exp_neg(a.post_std[idx] - a.min_std);
                if (lead) {
                    a.d_q2_out[(tb + gr) * 2 * a.S + k] = dm;
                    a.d_q2_out[(tb + gr) * 2 * a.S + a.S + k] = dr;
                }
            }
            dM[frag_idx(r, k)] = dm;
            dRaw[frag_idx(r, k)] = dr;
        }
        // saved operands of this lane's elements (every wave keeps its own copy): requested now
        floatx4 svq4 = floatx4{1.f, 1.f, 1.f, 1.f}, svx4 = svq4, gr4 = z4, gz4 = z4, gn4 = z4, gh4 = z4, hp4 = z4, dft4 = z4;
        if (rok) {
            svq4 = ks_row4(a.sv_q + (tb + grow) * a.Hd + fcol0, nhd);
            svx4 = ks_row4(a.sv_x + (tb + grow) * a.Be + fcol0, nbe);
            const float* gg = a.sv_gates + (tb + grow) * 4 * a.Be + fcol0;
            gr4 = ks_row4(gg, nbe); gz4 = ks_row4(gg + a.Be, nbe); gn4 = ks_row4(gg + 2 * a.Be, nbe); gh4 = ks_row4(gg + 3 * a.Be, nbe);
            hp4 = t > 0 ? ks_row4(a.feat + (tb - a.B + grow) * F + fcol0, nbe) : ks_row4(a.init_belief + (size_t)grow * a.Be + fcol0, nbe);
            dft4 = ks_row4(a.dfeat + (tb + grow) * F + fcol0, nbe);
        }
        lds_barrier();
        BD_KSTAMP(17);
        // ---- B2: dQ block c = ([dm | draw] W_q2)[:, block c] * ELU'(q_c): every wave, in registers ----
        floatx4 dq4;
        {
            floatx4 acc = z4;
            const floatx4* __restrict__ M4 = reinterpret_cast<const floatx4*>(dM) + lane;
            const floatx4* __restrict__ R4 = reinterpret_cast<const floatx4*>(dRaw) + lane;
#pragma unroll
            for (int kb = 0; kb < kKsMaxS; ++kb)
                if (kb < Kb_s) {
                    acc = mfmaT(w2m[kb], M4[kb * 64], acc);
                    acc = mfmaT(w2s[kb], R4[kb * 64], acc);
                }
#pragma unroll
            for (int i = 0; i < 4; ++i) dq4[i] = (rok && i < nhd) ? acc[i] * elu_grad_from_out(svq4[i]) : 0.f;
            if (wave == 0 && rok) ks_put4(a.d_q1_pre + (tb + grow) * a.Hd + fcol0, dq4, nhd);
        }
        BD_KSTAMP(18);
        BD_KARGS_FRESH(ap);
        // ---- B3: d belief partials over K block c; reduce-scatter #1 ----
        ++epoch;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int nbo = wave + kWaves * i;
            if (nbo < Kb_h) ks_store4(xb + kb_.q + (size_t)(nbo * C + c) * 256 + lane * 4, mfmaT(w1[i], dq4, z4));
        }
        publish(flags + c, epoch);
        BD_KSTAMP(19);
        wait_all(flags, C, epoch, err, spin_limit, kErrBwd);
        BD_KSTAMP(20);
        // ---- B4: total d belief of block c, gate gradients: every wave ----
        RED4[wave * 64 + lane] = ks_sum(xb + kb_.q + (size_t)(c * C) * 256, 256, wave, kWaves, C, lane);
        lds_barrier();
        floatx4 vr4, vz4, vni4, vnh4, carry4;
        {
            floatx4 dh4 = dhc4 + dft4;
            for (int w = 0; w < kWaves; ++w) dh4 += RED4[w * 64 + lane];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool ok = rok && i < nbe;
                const float dh = ok ? dh4[i] : 0.f;
                const float rr = gr4[i], zz = gz4[i], nn = gn4[i], hn = gh4[i];
                const float dn = dh * (1.f - zz);
                const float dz = dh * (hp4[i] - nn);
                vni4[i] = dn * (1.f - nn * nn);
                vnh4[i] = vni4[i] * rr;
                vr4[i] = vni4[i] * hn * rr * (1.f - rr);
                vz4[i] = dz * zz * (1.f - zz);
                carry4[i] = dh * zz;
            }
            if (wave == 0 && rok) {
                float* gi = a.d_gi + (tb + grow) * 3 * a.Be + fcol0;
                float* ghh = a.d_gh + (tb + grow) * 3 * a.Be + fcol0;
                ks_put4(gi, vr4, nbe); ks_put4(gi + a.Be, vz4, nbe); ks_put4(gi + 2 * a.Be, vni4, nbe);
                ks_put4(ghh, vr4, nbe); ks_put4(ghh + a.Be, vz4, nbe); ks_put4(ghh + 2 * a.Be, vnh4, nbe);
            }
        }
        BD_KSTAMP(21);
        BD_KARGS_FRESH(ap);
        // ---- B5: (DX, DH) partials over K block c through W_ih^T / W_hh^T; reduce-scatter #2 ----
        ++epoch;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int nbo = wave + kWaves * i;
            if (nbo < Kb_h) {
                floatx4 DX = mfmaT(wg[i][0], vr4, z4), DH = mfmaT(wg[i][3], vr4, z4);
                DX = mfmaT(wg[i][1], vz4, DX); DH = mfmaT(wg[i][4], vz4, DH);
                DX = mfmaT(wg[i][2], vni4, DX); DH = mfmaT(wg[i][5], vnh4, DH);
                float* dst = xb + kb_.g + ((size_t)(nbo * C + c) * 2) * 256 + lane * 4;
                ks_store4(dst, DX); ks_store4(dst + 256, DH);
            }
        }
        publish(flags + c, epoch);
        BD_KSTAMP(22);
        wait_all(flags, C, epoch, err, spin_limit, kErrBwd);
        BD_KSTAMP(23);
        // ---- B6: d embed pre-activation of block c, the carry's W_hh^T term: every wave ----
        {
            const int g = wave & 1, quarter = wave >> 1;
            RED4[wave * 64 + lane] = ks_sum(xb + kb_.g + ((size_t)(c * C) * 2 + g) * 256, (size_t)2 * 256, quarter, 4, C, lane);
        }
        lds_barrier();
        floatx4 de4;
        {
            floatx4 DX = z4, DH = z4;
            for (int qd = 0; qd < 4; ++qd) { DX += RED4[(2 * qd) * 64 + lane]; DH += RED4[(2 * qd + 1) * 64 + lane]; }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool ok = rok && i < nbe;
                de4[i] = ok ? DX[i] * elu_grad_from_out(svx4[i]) : 0.f;
                dhc4[i] = ok ? carry4[i] + DH[i] : 0.f;
            }
            if (wave == 0 && rok) ks_put4(a.d_embed_pre + (tb + grow) * a.Be + fcol0, de4, nbe);
        }
        BD_KSTAMP(24);
        BD_KARGS_FRESH(ap);
        // ---- B7: d state partials over K block c; all-reduce ----
        ++epoch;
        if (wave < Kb_s) ks_store4(xb + kb_.s + (size_t)(c * 8 + wave) * 256 + lane * 4, mfmaT(wes, de4, z4));
        publish(flags + c, epoch);
        BD_KSTAMP(25);
        wait_all(flags, C, epoch, err, spin_limit, kErrBwd);
        BD_KSTAMP(26);
        BD_KARGS_FRESH(ap);
        // ---- B8: d posterior_state_t (through the nonterminal mask of this step's input) ----
        if (wave < Kb_s) {
            const floatx4 v = ks_sum(xb + kb_.s + (size_t)wave * 256, (size_t)8 * 256, 0, 1, C, lane);
            const float mk = rok ? (a.nonterm ? a.nonterm[tb + grow] : 1.f) : 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int col = wave * 16 + 4 * (lane >> 4) + i;
                if (col < a.S) ds_plain[frow * a.S + col] = v[i] * mk;
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
    return ((size_t)(Kb_s + Kb_a) * kFragFloats + ((16 * S + 3) & ~3) + kWaves * 256 + 2 * 16 * Kb_s * 16) * sizeof(float);
}
static size_t ks_lds_bwd(int S) {
    const int Kb_s = cdiv(S, 16);
    return ((size_t)2 * Kb_s * kFragFloats + ((16 * S + 3) & ~3) + kWaves * 256) * sizeof(float);
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
