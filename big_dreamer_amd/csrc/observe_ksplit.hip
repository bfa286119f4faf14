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
// (The DEFAULT forward is observe_kfwd_ns_kernel further down: it splits the GRU by output columns and needs two hand-offs
// per step; the form described here stays selectable and is what the backward mirrors.)
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
// The other direction too: a store of more than 8 bytes keeps reading its data registers for two wait states after issue
// (the compiler's ">64-bit store data" hazard, which it cannot apply to an asm block); the granule form rewrites the staging
// tuple with the very next v_mov, and without the trailing s_nop the last lanes of each quarter sent the NEXT value.
__device__ __forceinline__ void ks_store4(float* p, floatx4 v) {
    asm volatile("s_nop 7\n\ts_nop 3\n\tglobal_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ floatx4 ks_load4(const float* p) {
    const unsigned long long lo = ld_sc1_u64(p), hi = ld_sc1_u64(p + 2);
    return floatx4{__uint_as_float((unsigned)lo), __uint_as_float((unsigned)(lo >> 32)), __uint_as_float((unsigned)hi),
                   __uint_as_float((unsigned)(hi >> 32))};
}
// ---- hand-off forms ---------------------------------------------------------------------------------------------------------
// GR = false: form R1 of bd_cluster.h (sc1 payload -> vmcnt(0) -> barrier -> flag; one wave polls the flags -> barrier -> sc1
//   loads).  s_memtime: a hand-off costs ~7k cycles end to end (publish 2k, poll 2.5-3.4k, the payload's own round trip 2k).
// GR = true (BD_OBS_KSPLIT=2; parity-tested, NOT the default -- measured below): form R2 of the guide ("the data IS the flag", cdna_hip_programming.md Guideline 16): every value travels in a
//   naturally aligned 8-byte granule {tag = epoch, value}; a float4 of a lane becomes two 16-byte sc1 stores
//   {tag, v0, tag, v1}, {tag, v2, tag, v3} (each 8-byte half of a 16-byte sc1 store is observed untorn on gfx950:
//   MI355X_MICROARCH.md, Valid forms), and the consumer polls the granules it is about to sum until every tag carries the
//   epoch -- no drain, no flag, no barrier, no second round trip.  Tags restart at 1 in every launch, so the launch function
//   zeroes the exchange buffer (a memset node ahead of the kernel); spins are bounded and end in the sticky error word.
//   Buffer reuse: every hand-off is all-to-all (a member passes hand-off k only when ALL members have stored for k, which
//   they do after finishing their reads of hand-off k - 1), so a region is rewritten at the next time step at the earliest
//   two full hand-offs after its last reader finished: one copy, no parity double-buffering.
//   Measured (tools/ks_stamps.py, member 0, cycles per step): forward 34.0k (R1) vs 36.8k (R2), backward 32.9k vs 31.9k; the
//   gate hand-off (52 KB of partials per member, 104 KB as granules) went from 12.3k to 15.1k, the two small ones gained 0.5k
//   each.  So the floor of a hand-off here is NOT the flag's extra round trip: it is the ~6k cycles (2.5 us) that a
//   write-through store takes to become readable from another CU while eight waves per CU keep the memory queue busy
//   (the guide's handoff-1to1 row under load), and doubling the bytes costs more than the saved round trip returns.
// An image = the float4 of 64 lanes: 256 floats (R1) or 512 (R2: chunk 0 = {tag, v0, tag, v1} of every lane, chunk 1 the rest).
template <bool GR> struct KsImg { static constexpr int floats = GR ? 512 : 256; };

template <bool GR>
__device__ __forceinline__ void ks_emit(float* img, int lane, floatx4 v, unsigned epoch) {
    if constexpr (GR) {
        const float tg = __uint_as_float(epoch);
        ks_store4(img + lane * 4, floatx4{tg, v[0], tg, v[1]});
        ks_store4(img + 256 + lane * 4, floatx4{tg, v[2], tg, v[3]});
    } else {
        ks_store4(img + lane * 4, v);
    }
}

constexpr int kKsChunk = 8;        // members whose loads one poll / sum pass keeps in flight (64 VGPRs either way)

// sum over the members src = first, first + stride, ... < C of one image (this lane's float4), in member order.
// R1: the caller has passed wait_all; all loads of a chunk are issued before the first add.
// R2: polls.  A probe pass re-reads ONE granule per member until its tag matches (cheap while the producers are still
// busy), then the full sweep loads the other three and checks every tag again (a mismatch there just repeats the pass).
template <bool GR>
__device__ __forceinline__ floatx4 ks_reduce(const float* base, size_t src_stride, int first, int stride, int C, int lane,
                                             unsigned epoch, unsigned* err, unsigned limit, unsigned code, bool& dead) {
    floatx4 s = floatx4{0.f, 0.f, 0.f, 0.f};
    if constexpr (!GR) {
        const float* p = base + lane * 4;
        for (int m0 = first; m0 < C; m0 += kKsChunk * stride) {
            floatx4 v[kKsChunk];
#pragma unroll
            for (int i = 0; i < kKsChunk; ++i) {
                const int src = m0 + i * stride;
                v[i] = floatx4{0.f, 0.f, 0.f, 0.f};
                if (src < C) v[i] = ks_load4(p + (size_t)src * src_stride);
            }
#pragma unroll
            for (int i = 0; i < kKsChunk; ++i) s += v[i];
        }
        return s;
    } else {
        const float* p = base + lane * 4;
        unsigned spins = 0;       // `dead` is the wave's: after one time-out it never spins again (the launch still terminates)
        auto timed_out = [&]() {
            if (++spins <= limit) return false;
            if (lane == 0) __hip_atomic_fetch_or(err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            dead = true;
            return true;
        };
        for (int m0 = first; m0 < C; m0 += kKsChunk * stride) {
            unsigned long long g0[kKsChunk];
            // probe: granule 0 of every member of the chunk
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int i = 0; i < kKsChunk; ++i)
                    if (m0 + i * stride < C) g0[i] = ld_sc1_u64(p + (size_t)(m0 + i * stride) * src_stride);
#pragma unroll
                for (int i = 0; i < kKsChunk; ++i)
                    if (m0 + i * stride < C) ok &= (unsigned)g0[i] == epoch;
                if (__all(ok) || dead || timed_out()) break;
                __builtin_amdgcn_s_sleep(1);
            }
            // sweep, four members at a time: the other three granules (granule 0 matched above)
#pragma unroll
            for (int j0 = 0; j0 < kKsChunk; j0 += 4) {
                if (m0 + j0 * stride >= C) break;
                unsigned long long g[4][3];
                for (;;) {
                    bool ok = true;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int src = m0 + (j0 + j) * stride;
                        if (src < C) {
                            const float* q = p + (size_t)src * src_stride;
                            g[j][0] = ld_sc1_u64(q + 2);
                            g[j][1] = ld_sc1_u64(q + 256);
                            g[j][2] = ld_sc1_u64(q + 258);
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (m0 + (j0 + j) * stride < C)
                            ok &= (unsigned)g[j][0] == epoch && (unsigned)g[j][1] == epoch && (unsigned)g[j][2] == epoch;
                    if (__all(ok) || dead || timed_out()) break;
                    __builtin_amdgcn_s_sleep(1);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (m0 + (j0 + j) * stride < C)
                        s += floatx4{__uint_as_float((unsigned)(g0[j0 + j] >> 32)), __uint_as_float((unsigned)(g[j][0] >> 32)),
                                     __uint_as_float((unsigned)(g[j][1] >> 32)), __uint_as_float((unsigned)(g[j][2] >> 32))};
            }
        }
        return s;
    }
}
// one hand-off: R1 publishes and waits; R2 has nothing to do here (ks_reduce polls)
template <bool GR>
__device__ __forceinline__ void ks_handoff(unsigned* flags, int c, int C, unsigned epoch, unsigned* err, unsigned limit,
                                           unsigned code) {
    if constexpr (!GR) {
        publish(flags + c, epoch);
        wait_all(flags, C, epoch, err, limit, code);
    }
}
__device__ __forceinline__ floatx4 mfma4(floatx4 a, floatx4 b, floatx4 acc) {
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = mfma16(a[j], b[j], acc);
    return acc;
}
// Loop-invariant registers that were LOADED before the time loop (weight slices, biases, the initial carry): the compiler's
// waitcnt pass cannot prove across the back edge that those loads have completed, so it puts `s_waitcnt vmcnt(0)` in front
// of their first use in EVERY iteration -- which drains whatever the step has in flight at that point (the previous phase's
// stores; operand prefetches issued at the top of the step: their whole HBM latency).  After one explicit wait the values
// are passed through an empty asm: the loop then sees registers defined by the asm, not by a load.
__device__ __forceinline__ void ks_settle(floatx4& v) { asm volatile("" : "+v"(v)); }
template <int N>
__device__ __forceinline__ void ks_settle(floatx4 (&v)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) ks_settle(v[i]);
}
#define KS_SETTLE_BEGIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")

// packed weight fragment (out block nb, in block kb) of a matrix packed with Kb input blocks; zero when `ok` is false
__device__ __forceinline__ floatx4 ks_frag(const float* w, int nb, int Kb, int kb, int lane, bool ok) {
    return ok ? reinterpret_cast<const floatx4*>(w)[((size_t)nb * Kb + kb) * 64 + lane] : floatx4{0.f, 0.f, 0.f, 0.f};
}

// exchange buffers of one tile (floats); R1 keeps two copies (step parity), R2 one (see above)
struct KsBuf {
    size_t g, q, s, total;
    int copies;
    __host__ __device__ KsBuf(int C, bool gr) {
        const size_t img = gr ? 512 : 256;
        g = 0;
        q = g + (size_t)C * C * 4 * img;     // [dest][src][4 accumulators][image]
        s = q + (size_t)C * C * img;         // [dest][src][image]
        total = s + (size_t)C * 8 * img;     // [src][<= 8 (block, mean | raw) pairs][image]
        copies = gr ? 1 : 2;
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
// four consecutive floats of a row-major row.  Rows of the feature tensors start at 4-byte granularity (F = Be + S floats), so
// the 16-byte access goes through memcpy: the compiler emits ONE global_load / global_store_dwordx4 for a 4-byte aligned
// address (gfx950 global memory takes unaligned vector accesses).  The earlier form branched on the address and fell back to
// four predicated dword loads into a zero-initialised tuple -- a write-after-write on registers with loads in flight, which the
// compiler guards with `s_waitcnt vmcnt(0)`: the operand prefetches of a step then completed one after the other.
__device__ __forceinline__ floatx4 ks_row4(const float* p, int n_valid) {
    floatx4 v = floatx4{0.f, 0.f, 0.f, 0.f};
    if (n_valid >= 4) {
        __builtin_memcpy(&v, p, 16);
        return v;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (i < n_valid) v[i] = p[i];
    return v;
}
__device__ __forceinline__ void ks_put4(float* p, floatx4 v, int n_valid) {
    if (n_valid >= 4) {
        __builtin_memcpy(p, &v, 16);
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (i < n_valid) p[i] = v[i];
}

// ---- forward ---------------------------------------------------------------------------------------------------------
template <bool GR>
__global__ __launch_bounds__(kThreads) void observe_kfwd_kernel(bd_observe_fwd_args a_, float* __restrict__ ws, int C, int tiles,
                                                                unsigned spin_limit) {
    constexpr int IMG = KsImg<GR>::floats;
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
    float* red = s_plain + ((16 * a.S + 3) & ~3);   // [2][8 waves][64][4]: the two reductions of a step alternate
    float* plain = red + 2 * kWaves * 256;          // [parts][2][16][Np]: head sums, one copy per helper wave group

    unsigned* flags = reinterpret_cast<unsigned*>(ws) + tile * kMaxCluster;
    unsigned* err = reinterpret_cast<unsigned*>(ws) + tiles * kMaxCluster;
    const KsBuf kb_(C, GR);
    float* xbase = ws + cluster_ws_header_floats(tiles) + (size_t)tile * kb_.copies * kb_.total;

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
    // head pairs (state block, mean | raw): wave w works on pair w % npairs; the waves with part = w / npairs == 0 form the
    // partial products, and all nparts groups share the sum over the members (group `part` takes members part, part + nparts, ..)
    const int npairs = 2 * Kb_s, nparts = kWaves / npairs;
    const int pair = wave % npairs, part = wave / npairs;
    const int pair_nb = pair >> 1, pair_raw = pair & 1;
    const bool has_pair = part == 0, sums_pair = part < nparts;
    floatx4 wh = ks_frag(pair_raw ? a.w_q2s : a.w_q2m, pair_nb, Kb_hd, c, lane, has_pair && c < Kb_hd);
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

    KS_SETTLE_BEGIN();
    ks_settle(we_s); ks_settle(we_a); ks_settle(wg[0]); ks_settle(wg[1]); ks_settle(wq);
    ks_settle(wh);
    ks_settle(br4); ks_settle(bz4); ks_settle(bni4); ks_settle(bnh4); ks_settle(be4); ks_settle(bq4); ks_settle(bh4); ks_settle(h4);
    const bool lead = (c == 0);
    floatx4* __restrict__ RED4 = reinterpret_cast<floatx4*>(red);
    floatx4* __restrict__ REDB = RED4 + kWaves * 64;
    const floatx4 z4 = floatx4{0.f, 0.f, 0.f, 0.f};
    bool dead = false;

    // this thread's first elements of phase A (nonterminal flag of its state element, its action element): requested one step
    // ahead, so that a step does not start behind an HBM round trip
    auto load_nt = [&](int t, int i) {
        const int r = i / (Kb_s * 16), k = i - r * (Kb_s * 16);
        return (a.nonterm && i < 16 * Kb_s * 16 && row0 + r < a.B && k < a.S) ? a.nonterm[(size_t)t * a.B + row0 + r] : 1.f;
    };
    auto load_ac = [&](int t, int i) {
        const int r = i / (Kb_a * 16), k = i - r * (Kb_a * 16);
        return (i < 16 * Kb_a * 16 && row0 + r < a.B && k < a.A) ? a.actions[((size_t)t * a.B + row0 + r) * a.A + k] : 0.f;
    };
    float n_nt = load_nt(0, (int)threadIdx.x), n_ac = load_ac(0, (int)threadIdx.x);

    for (int t = 0; t < a.T; ++t) {
        const size_t tb = (size_t)t * a.B;
        const int tid = bd_tid();
        float* xb = xbase + (size_t)(GR ? 0 : (t & 1)) * kb_.total;
        BD_KSTAMP(0);
        BD_KARGS_FRESH(ap);
        const float c_nt = n_nt, c_ac = n_ac;
        if (t + 1 < a.T) {
            n_nt = load_nt(t + 1, tid);
            n_ac = load_ac(t + 1, tid);
        }
        // ---- A: masked state / action fragments (every member; K of the embed layer is tiny) ----
        // (ksplit_ok: 16 * S <= 512 threads, so these element loops are ONE pass and every thread uses the operands it
        //  requested a step ahead -- a fallback load in the loop body would put a vmcnt wait on the common path)
        if (tid < 16 * Kb_s * 16) {
            const int i = tid;
            const int r = i / (Kb_s * 16), k = i - r * (Kb_s * 16);
            const int gr = row0 + r;
            float v = 0.f;
            if (gr < a.B && k < a.S) {
                v = s_plain[r * a.S + k] * c_nt;
                if (lead && a.sv_s) a.sv_s[(tb + gr) * a.S + k] = v;
            }
            sf[frag_idx(r, k)] = v;
        }
        if (tid < 16 * Kb_a * 16) af[frag_idx(tid / (Kb_a * 16), tid % (Kb_a * 16))] = c_ac;
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
                float* dst = xb + kb_.g + ((size_t)(nbo * C + c) * 4) * IMG;
                const unsigned ep = (unsigned)(3 * t + 1);
                ks_emit<GR>(dst, lane, R, ep); ks_emit<GR>(dst + IMG, lane, Z, ep);
                ks_emit<GR>(dst + 2 * IMG, lane, NI, ep); ks_emit<GR>(dst + 3 * IMG, lane, NH, ep);
            }
        }
        BD_KSTAMP(3);
        ks_handoff<GR>(flags, c, C, (unsigned)(3 * t + 1), err, spin_limit, kErrFwd);
        BD_KSTAMP(4);
        // ---- F3: sum the C partials of block c (wave = (gate, half of the members)); every wave finishes the GRU gates ----
        {
            const int g = wave & 3, half = wave >> 2;
            RED4[wave * 64 + lane] = ks_reduce<GR>(xb + kb_.g + ((size_t)(c * C) * 4 + g) * IMG, (size_t)4 * IMG, half, 2, C, lane,
                                                   (unsigned)(3 * t + 1), err, spin_limit, kErrFwd, dead);
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
            if (nbo < Kb_hd) ks_emit<GR>(xb + kb_.q + (size_t)(nbo * C + c) * IMG, lane, mfmaT(wq[i], h4, z4), (unsigned)(3 * t + 2));
        }
        BD_KSTAMP(6);
        ks_handoff<GR>(flags, c, C, (unsigned)(3 * t + 2), err, spin_limit, kErrFwd);
        BD_KSTAMP(7);
        // ---- F5: q_c = ELU(sum + pre_emb_c + b): every wave (second RED buffer: a slow wave may still read the first) ----
        REDB[wave * 64 + lane] = c < Kb_hd ? ks_reduce<GR>(xb + kb_.q + (size_t)(c * C) * IMG, IMG, wave, kWaves, C, lane,
                                                          (unsigned)(3 * t + 2), err, spin_limit, kErrFwd, dead) : z4;
        lds_barrier();
        floatx4 q4 = bq4 + pre4;
        for (int w = 0; w < kWaves; ++w) q4 += REDB[w * 64 + lane];
#pragma unroll
        for (int i = 0; i < 4; ++i) q4[i] = (rok && i < nhd) ? elu(q4[i]) : 0.f;
        if (wave == 0 && rok && a.sv_q) ks_put4(a.sv_q + (tb + grow) * a.Hd + fcol0, q4, nhd);
        BD_KSTAMP(8);
        BD_KARGS_FRESH(ap);
        // ---- F6: (mean, raw) partials over K block c; all-reduce ----
        if (has_pair) ks_emit<GR>(xb + kb_.s + (size_t)(c * 8 + pair) * IMG, lane, mfmaT(wh, q4, z4), (unsigned)(3 * t + 3));
        BD_KSTAMP(9);
        ks_handoff<GR>(flags, c, C, (unsigned)(3 * t + 3), err, spin_limit, kErrFwd);
        BD_KSTAMP(10);
        BD_KARGS_FRESH(ap);
        // ---- F7: every member sums the head partials (wave group `part` its share of the members), then samples s' ----
        if (sums_pair) {
            floatx4 v = ks_reduce<GR>(xb + kb_.s + (size_t)pair * IMG, (size_t)8 * IMG, part, nparts, C, lane, (unsigned)(3 * t + 3), err,
                                      spin_limit, kErrFwd, dead);
            if (part == 0) v += bh4;
            float* pl = plain + (size_t)part * 2 * 16 * Np;
#pragma unroll
            for (int i = 0; i < 4; ++i) pl[pair_raw * 16 * Np + frow * Np + pair_nb * 16 + 4 * (lane >> 4) + i] = v[i];
        }
        lds_barrier();
        if (tid < 16 * a.S) {
            const int e = tid;
            const int row = e / a.S, col = e - row * a.S;
            const int gr = row0 + row;
            float st = 0.f;
            if (gr < a.B) {
                float Mn = plain[row * Np + col], Rw = plain[16 * Np + row * Np + col];
                for (int pt = 1; pt < nparts; ++pt) {
                    Mn += plain[pt * 2 * 16 * Np + row * Np + col];
                    Rw += plain[pt * 2 * 16 * Np + 16 * Np + row * Np + col];
                }
                const float ee = eps;
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

// ---- forward, GRU split by OUTPUT columns (two hand-offs per step) ---------------------------------------------------------
// The K-split forward above pays three hand-offs per step, and the first one carries the gate partials of all 13 blocks
// (52 KB per member).  Here member c forms the four gate pre-activations of ITS block over the full K instead:
//   F1  x (ALL blocks) = ELU(W_e [s; a] + b)      every member, redundantly: K = S + A is tiny (wave w: blocks w, w + 8) -> LDS
//   F2  R, Z, NI, NH of block c = W_i*[block c, :] x + W_h*[block c, :] h      K split over the WAVES (wave w: K blocks w, w + 8),
//       partial sums meet in LDS -- no hand-off; the weights are the same 2 x 6 fragments per wave, other slices of them
//   F3  h'_c = GRU gates of block c; stored into the exchange buffer as well: the belief all-gather
//   F4  q partials over K block c -> reduce-scatter (hand-off 1; its flag also covers the belief block stored before it)
//       ... then every member gathers the 13 belief blocks into LDS for the NEXT step's F2: off the critical path
//   F5 - F7 as above (hand-off 2: head partials all-reduce)
// The backward keeps its three hand-offs (its reduce-scatters carry gradients that no member can form alone).
__global__ __launch_bounds__(kThreads) void observe_kfwd_ns_kernel(bd_observe_fwd_args a_, float* __restrict__ ws, int C, int tiles,
                                                                   unsigned spin_limit) {
    constexpr bool GR = false;
    constexpr int IMG = KsImg<GR>::floats;
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
    float* red = s_plain + ((16 * a.S + 3) & ~3);   // [2][8 waves][64][4]: the two reductions of a step alternate
    float* plain = red + 2 * kWaves * 256;          // [parts][2][16][Np]: head sums, one copy per helper wave group
    float* xf = plain + kWaves * 256;               // x, all Kb_h blocks, fragment tiles
    float* hf = xf + Kb_h * kFragFloats;            // h, all Kb_h blocks, fragment tiles
    float* gpart = hf + Kb_h * kFragFloats;         // [8 waves][4 gates][64][4]: the waves' K-partial gate sums

    unsigned* flags = reinterpret_cast<unsigned*>(ws) + tile * kMaxCluster;
    unsigned* err = reinterpret_cast<unsigned*>(ws) + tiles * kMaxCluster;
    const KsBuf kb_(C, GR);
    float* xbase = ws + cluster_ws_header_floats(tiles) + (size_t)tile * kb_.copies * kb_.total;

    // ---- resident weight slices ----
    floatx4 we_s[2][kKsMaxS], we_a[2][kKsMaxA], be4x[2];      // embed: output blocks wave, wave + 8
    floatx4 wg[2][6], wq[2];                                  // gates: block c, K blocks wave, wave + 8;  q: K block c
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int nbo = wave + kWaves * i;
        const bool ok = nbo < Kb_h;
#pragma unroll
        for (int kb = 0; kb < kKsMaxS; ++kb) we_s[i][kb] = ks_frag(a.w_embed_s, nbo, Kb_s, kb, lane, ok && kb < Kb_s);
#pragma unroll
        for (int kb = 0; kb < kKsMaxA; ++kb) we_a[i][kb] = ks_frag(a.w_embed_a, nbo, Kb_a, kb, lane, ok && kb < Kb_a);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = nbo * 16 + 4 * (lane >> 4) + j;
            be4x[i][j] = (ok && col < a.Be) ? a.b_embed[col] : 0.f;
        }
        wg[i][0] = ks_frag(a.w_ir, c, Kb_h, nbo, lane, ok); wg[i][1] = ks_frag(a.w_iz, c, Kb_h, nbo, lane, ok);
        wg[i][2] = ks_frag(a.w_in, c, Kb_h, nbo, lane, ok); wg[i][3] = ks_frag(a.w_hr, c, Kb_h, nbo, lane, ok);
        wg[i][4] = ks_frag(a.w_hz, c, Kb_h, nbo, lane, ok); wg[i][5] = ks_frag(a.w_hn, c, Kb_h, nbo, lane, ok);
        wq[i] = ks_frag(a.w_q1h, nbo, Kb_h, c, lane, nbo < Kb_hd);
    }
    const int npairs = 2 * Kb_s, nparts = kWaves / npairs;
    const int pair = wave % npairs, part = wave / npairs;
    const int pair_nb = pair >> 1, pair_raw = pair & 1;
    const bool has_pair = part == 0, sums_pair = part < nparts;
    floatx4 wh = ks_frag(pair_raw ? a.w_q2s : a.w_q2m, pair_nb, Kb_hd, c, lane, has_pair && c < Kb_hd);
    const int frow = lane & 15, fcol0 = c * 16 + 4 * (lane >> 4);
    const int grow = row0 + frow;
    const int nbe = a.Be - fcol0 < 0 ? 0 : (a.Be - fcol0 < 4 ? a.Be - fcol0 : 4);
    const int nhd = a.Hd - fcol0 < 0 ? 0 : (a.Hd - fcol0 < 4 ? a.Hd - fcol0 : 4);
    const bool rok = grow < a.B;
    floatx4 br4 = floatx4{0.f, 0.f, 0.f, 0.f}, bz4 = br4, bni4 = br4, bnh4 = br4, bq4 = br4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i < nbe) {
            const int col = fcol0 + i;
            br4[i] = a.b_ih[col] + a.b_hh[col];
            bz4[i] = a.b_ih[a.Be + col] + a.b_hh[a.Be + col];
            bni4[i] = a.b_ih[2 * a.Be + col];
            bnh4[i] = a.b_hh[2 * a.Be + col];
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
    // ---- initial carries: this lane's elements of block c in registers, the whole belief as fragment tiles in LDS ----
    floatx4 h4 = rok ? ks_row4(a.init_belief + (size_t)grow * a.Be + fcol0, nbe) : floatx4{0.f, 0.f, 0.f, 0.f};
    floatx4* __restrict__ XF4 = reinterpret_cast<floatx4*>(xf);
    floatx4* __restrict__ HF4 = reinterpret_cast<floatx4*>(hf);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int nbo = wave + kWaves * i;
        if (nbo < Kb_h) {
            const int col0 = nbo * 16 + 4 * (lane >> 4);
            const int nv = a.Be - col0 < 0 ? 0 : (a.Be - col0 < 4 ? a.Be - col0 : 4);
            HF4[nbo * 64 + lane] = rok ? ks_row4(a.init_belief + (size_t)grow * a.Be + col0, nv) : floatx4{0.f, 0.f, 0.f, 0.f};
        }
    }
    // the first step's masked state / action fragments (later steps: written by the sampling phase of the step before)
    for (int i = threadIdx.x; i < 16 * Kb_s * 16; i += blockDim.x) {
        const int r = i / (Kb_s * 16), k = i - r * (Kb_s * 16);
        float v = 0.f;
        if (row0 + r < a.B && k < a.S) {
            v = a.init_state[(size_t)(row0 + r) * a.S + k] * (a.nonterm ? a.nonterm[row0 + r] : 1.f);
            if (c == 0 && a.sv_s) a.sv_s[(size_t)(row0 + r) * a.S + k] = v;
        }
        sf[frag_idx(r, k)] = v;
    }
    for (int i = threadIdx.x; i < 16 * Kb_a * 16; i += blockDim.x) {
        const int r = i / (Kb_a * 16), k = i - r * (Kb_a * 16);
        af[frag_idx(r, k)] = (row0 + r < a.B && k < a.A) ? a.actions[(size_t)(row0 + r) * a.A + k] : 0.f;
    }
    lds_barrier();

    KS_SETTLE_BEGIN();
    ks_settle(we_s[0]); ks_settle(we_s[1]); ks_settle(we_a[0]); ks_settle(we_a[1]); ks_settle(be4x);
    ks_settle(wg[0]); ks_settle(wg[1]); ks_settle(wq);
    ks_settle(wh);
    ks_settle(br4); ks_settle(bz4); ks_settle(bni4); ks_settle(bnh4); ks_settle(bq4); ks_settle(bh4); ks_settle(h4);
    const bool lead = (c == 0);
    floatx4* __restrict__ RED4 = reinterpret_cast<floatx4*>(red);
    floatx4* __restrict__ REDB = RED4 + kWaves * 64;
    floatx4* __restrict__ GP4 = reinterpret_cast<floatx4*>(gpart);
    const floatx4 z4 = floatx4{0.f, 0.f, 0.f, 0.f};
    bool dead = false;

    // operands of the NEXT step's fragments, in the sampling phase's element order (e -> row e / S), requested a step ahead
    auto load_nt = [&](int t, int e) {
        const int r = e / a.S;
        return (a.nonterm && t < a.T && e < 16 * a.S && row0 + r < a.B) ? a.nonterm[(size_t)t * a.B + row0 + r] : 1.f;
    };
    auto load_ac = [&](int t, int i) {
        const int r = i / (Kb_a * 16), k = i - r * (Kb_a * 16);
        return (t < a.T && i < 16 * Kb_a * 16 && row0 + r < a.B && k < a.A) ? a.actions[((size_t)t * a.B + row0 + r) * a.A + k] : 0.f;
    };

    for (int t = 0; t < a.T; ++t) {
        const size_t tb = (size_t)t * a.B;
        const int tid = bd_tid();
        float* xb = xbase + (size_t)(t & 1) * kb_.total;
        float* hx = xb + kb_.g;                      // [src][256]: the belief blocks (the gate-partial region is free in this form)
        BD_KSTAMP(0);
        BD_KARGS_FRESH(ap);
        // (no phase A: this step's masked state / action fragments were written by the step before)
        BD_KSTAMP(1);
        // ---- F1: x, blocks wave and wave + 8 -> fragment tiles in LDS ----
        // (12 dependent MFMAs per block at S = 30, A = 1: 0.4k cycles of issue each, two waves per SIMD.  A branch-free form
        //  that pads to the compile-time maxima doubles the MFMAs and measured slower: 4.5k against 3.8k cycles.)
        {
            const floatx4* __restrict__ S4 = reinterpret_cast<const floatx4*>(sf) + lane;
            const floatx4* __restrict__ A4 = reinterpret_cast<const floatx4*>(af) + lane;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int nbo = wave + kWaves * i;
                if (nbo < Kb_h) {
                    floatx4 acc = be4x[i];
#pragma unroll
                    for (int kb = 0; kb < kKsMaxS; ++kb)
                        if (kb < Kb_s) acc = mfmaT(we_s[i][kb], S4[kb * 64], acc);
#pragma unroll
                    for (int kb = 0; kb < kKsMaxA; ++kb)
                        if (kb < Kb_a) acc = mfmaT(we_a[i][kb], A4[kb * 64], acc);
                    floatx4 x4;
                    const int col0 = nbo * 16 + 4 * (lane >> 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) x4[j] = (rok && col0 + j < a.Be) ? elu(acc[j]) : 0.f;
                    XF4[nbo * 64 + lane] = x4;
                    if (nbo == c && rok && a.sv_x) ks_put4(a.sv_x + (tb + grow) * a.Be + fcol0, x4, nbe);
                }
            }
        }
        lds_barrier();
        BD_KSTAMP(2);
        BD_KARGS_FRESH(ap);
        // ---- F2: gate pre-activations of block c: this wave's K blocks; the eight partial sums meet in LDS ----
        {
            floatx4 R = z4, Z = z4, NI = z4, NH = z4;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int kb = wave + kWaves * i;
                if (kb < Kb_h) {
                    const floatx4 xk = XF4[kb * 64 + lane], hk = HF4[kb * 64 + lane];
                    R = mfmaT(wg[i][0], xk, R); Z = mfmaT(wg[i][1], xk, Z); NI = mfmaT(wg[i][2], xk, NI);
                    R = mfmaT(wg[i][3], hk, R); Z = mfmaT(wg[i][4], hk, Z); NH = mfmaT(wg[i][5], hk, NH);
                }
            }
            GP4[(wave * 4 + 0) * 64 + lane] = R; GP4[(wave * 4 + 1) * 64 + lane] = Z;
            GP4[(wave * 4 + 2) * 64 + lane] = NI; GP4[(wave * 4 + 3) * 64 + lane] = NH;
        }
        lds_barrier();
        BD_KSTAMP(3);
        BD_KSTAMP(4);
        // ---- F3: wave = (gate, half of the waves' partials), then every wave finishes the GRU gates ----
        {
            const int g = wave & 3, half = wave >> 2;
            floatx4 s4 = z4;
#pragma unroll
            for (int w = 0; w < 4; ++w) s4 += GP4[((half * 4 + w) * 4 + g) * 64 + lane];
            RED4[wave * 64 + lane] = s4;
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
            if (wave == 0) {
                ks_store4(hx + (size_t)c * 256 + lane * 4, h4);        // the belief all-gather: announced by the next publish
                if (rok) {
                    ks_put4(a.feat + (tb + grow) * F + fcol0, h4, nbe);
                    if (a.sv_gates) {
                        float* gg = a.sv_gates + (tb + grow) * 4 * a.Be + fcol0;
                        ks_put4(gg, rr4, nbe); ks_put4(gg + a.Be, zz4, nbe); ks_put4(gg + 2 * a.Be, nn4, nbe); ks_put4(gg + 3 * a.Be, NH, nbe);
                    }
                }
            }
        }
        BD_KSTAMP(5);
        BD_KARGS_FRESH(ap);
        // operands of the later phases and of the next step: requested HERE, behind the phases that issue no loads (a load in
        // flight at the top of the step made the compiler drain vmcnt in front of F1's MFMAs: 3k cycles per step)
        const float n_nt = load_nt(t + 1, tid), n_ac = load_ac(t + 1, tid);
        const floatx4 pre4 = rok ? ks_row4(a.pre_emb + (tb + grow) * a.Hd + fcol0, nhd) : z4;
        float eps = 0.f;
        {
            const int row = tid / a.S, col = tid - row * a.S;
            if (tid < 16 * a.S && row0 + row < a.B) eps = a.eps_post[(tb + row0 + row) * a.S + col];
        }
        // ---- F4: posterior-hidden partials over K block c; reduce-scatter (hand-off 1) ----
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int nbo = wave + kWaves * i;
            if (nbo < Kb_hd) ks_emit<GR>(xb + kb_.q + (size_t)(nbo * C + c) * IMG, lane, mfmaT(wq[i], h4, z4), (unsigned)(3 * t + 2));
        }
        BD_KSTAMP(6);
        ks_handoff<GR>(flags, c, C, (unsigned)(3 * t + 2), err, spin_limit, kErrFwd);
        BD_KSTAMP(7);
        // ---- F5: q_c = ELU(sum + pre_emb_c + b); and the belief blocks of all members into LDS for the next step's F2 ----
        REDB[wave * 64 + lane] = c < Kb_hd ? ks_reduce<GR>(xb + kb_.q + (size_t)(c * C) * IMG, IMG, wave, kWaves, C, lane,
                                                          (unsigned)(3 * t + 2), err, spin_limit, kErrFwd, dead) : z4;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int nbo = wave + kWaves * i;
            if (nbo < Kb_h) HF4[nbo * 64 + lane] = ks_load4(hx + (size_t)nbo * 256 + lane * 4);     // (read again only after barriers)
        }
        lds_barrier();
        floatx4 q4 = bq4 + pre4;
        for (int w = 0; w < kWaves; ++w) q4 += REDB[w * 64 + lane];
#pragma unroll
        for (int i = 0; i < 4; ++i) q4[i] = (rok && i < nhd) ? elu(q4[i]) : 0.f;
        if (wave == 0 && rok && a.sv_q) ks_put4(a.sv_q + (tb + grow) * a.Hd + fcol0, q4, nhd);
        BD_KSTAMP(8);
        BD_KARGS_FRESH(ap);
        // ---- F6: (mean, raw) partials over K block c; all-reduce (hand-off 2) ----
        if (has_pair) ks_emit<GR>(xb + kb_.s + (size_t)(c * 8 + pair) * IMG, lane, mfmaT(wh, q4, z4), (unsigned)(3 * t + 3));
        BD_KSTAMP(9);
        ks_handoff<GR>(flags, c, C, (unsigned)(3 * t + 3), err, spin_limit, kErrFwd);
        BD_KSTAMP(10);
        BD_KARGS_FRESH(ap);
        // ---- F7: every member sums the head partials, then samples s' ----
        if (sums_pair) {
            floatx4 v = ks_reduce<GR>(xb + kb_.s + (size_t)pair * IMG, (size_t)8 * IMG, part, nparts, C, lane, (unsigned)(3 * t + 3), err,
                                      spin_limit, kErrFwd, dead);
            if (part == 0) v += bh4;
            float* pl = plain + (size_t)part * 2 * 16 * Np;
#pragma unroll
            for (int i = 0; i < 4; ++i) pl[pair_raw * 16 * Np + frow * Np + pair_nb * 16 + 4 * (lane >> 4) + i] = v[i];
        }
        lds_barrier();
        if (tid < 16 * a.S) {
            const int e = tid;
            const int row = e / a.S, col = e - row * a.S;
            const int gr = row0 + row;
            float st = 0.f;
            if (gr < a.B) {
                float Mn = plain[row * Np + col], Rw = plain[16 * Np + row * Np + col];
                for (int pt = 1; pt < nparts; ++pt) {
                    Mn += plain[pt * 2 * 16 * Np + row * Np + col];
                    Rw += plain[pt * 2 * 16 * Np + 16 * Np + row * Np + col];
                }
                const float ee = eps;
                const float sd = softplusf(Rw) + a.min_std;
                st = Mn + sd * ee;
                if (lead) {
                    const size_t ix = (tb + gr) * a.S + col;
                    a.post_mean[ix] = Mn;
                    a.post_std[ix] = sd;
                    a.feat[(tb + gr) * F + a.Be + col] = st;
                }
            }
            // the next step's input: s' through that step's nonterminal mask, straight into the fragment tile
            const float mv = st * n_nt;
            sf[frag_idx(row, col)] = mv;
            if (lead && a.sv_s && t + 1 < a.T && gr < a.B) a.sv_s[(tb + a.B + gr) * a.S + col] = mv;
        }
        if (tid < 16 * Kb_a * 16) af[frag_idx(tid / (Kb_a * 16), tid % (Kb_a * 16))] = n_ac;
        lds_barrier();
        BD_KSTAMP(11);
    }
#undef a
}

// ---- backward --------------------------------------------------------------------------------------------------------
template <bool GR>
__global__ __launch_bounds__(kThreads) void observe_kbwd_kernel(bd_observe_bwd_args a_, float* __restrict__ ws, int C, int tiles,
                                                                unsigned spin_limit) {
    constexpr int IMG = KsImg<GR>::floats;
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
    float* ds_plain = dRaw + Kb_s * kFragFloats;    // [parts][16][S]: d state of the next step, one partial sum per wave group
    const int nparts = kWaves / Kb_s;               // wave w: state block w % Kb_s, members part = w / Kb_s, part + nparts, ...
    const int dsp_stride = (16 * a.S + 3) & ~3;
    float* red = ds_plain + (size_t)nparts * dsp_stride;   // [2][8 waves][64][4]

    unsigned* flags = reinterpret_cast<unsigned*>(ws) + tile * kMaxCluster;
    unsigned* err = reinterpret_cast<unsigned*>(ws) + tiles * kMaxCluster;
    const KsBuf kb_(C, GR);                         // backward uses: g as [dest][src][2][image] (DX, DH), q as [dest][src][image] (dh),
    float* xbase = ws + cluster_ws_header_floats(tiles) + (size_t)tile * kb_.copies * kb_.total;   // s as [src][Kb_s][image] (ds)

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
    const int sblk = wave % Kb_s, spart = wave / Kb_s;
    floatx4 wes = ks_frag(a.wt_embed_s, sblk, Kb_h, c, lane, spart == 0);  // (out = S, in = Be): state block of this wave
    const int frow = lane & 15, fcol0 = c * 16 + 4 * (lane >> 4);
    const int grow = row0 + frow;
    const int nbe = a.Be - fcol0 < 0 ? 0 : (a.Be - fcol0 < 4 ? a.Be - fcol0 : 4);
    const int nhd = a.Hd - fcol0 < 0 ? 0 : (a.Hd - fcol0 < 4 ? a.Hd - fcol0 : 4);
    const bool rok = grow < a.B;

    for (int i = threadIdx.x; i < nparts * dsp_stride; i += blockDim.x) ds_plain[i] = 0.f;
    lds_barrier();

    KS_SETTLE_BEGIN();
    ks_settle(w2m); ks_settle(w2s); ks_settle(w1); ks_settle(wg[0]); ks_settle(wg[1]);
    ks_settle(wes);
    const bool lead = (c == 0);
    floatx4* __restrict__ RED4 = reinterpret_cast<floatx4*>(red);
    floatx4* __restrict__ REDB = RED4 + kWaves * 64;
    const floatx4 z4 = floatx4{0.f, 0.f, 0.f, 0.f};
    floatx4 dhc4 = z4;                               // belief-gradient carry of block c (every wave holds a copy)
    unsigned epoch = 0;
    bool dead = false;

    // Operands that do not depend on the recurrence (saved activations, incoming gradients, noise) are requested ONE STEP AHEAD:
    // a step starts with them in registers instead of behind an HBM round trip (B1 was 6.4k of 32.9k cycles, tools/ks_stamps.py).
    struct B1v { float dfs, dpm, eps, dps, pstd; };
    struct Svv { floatx4 svq4, svx4, gr4, gz4, gn4, gh4, hp4, dft4; };
    auto load_b1 = [&](int t, int i) {
        B1v v{0.f, 0.f, 0.f, 0.f, 0.f};
        const int r = i / (Kb_s * 16), k = i - r * (Kb_s * 16);
        const int gr = row0 + r;
        if (i < 16 * Kb_s * 16 && gr < a.B && k < a.S) {
            const size_t tb = (size_t)t * a.B, idx = (tb + gr) * a.S + k;
            v.dfs = a.dfeat[(tb + gr) * F + a.Be + k];
            v.dpm = a.dpost_mean ? a.dpost_mean[idx] : 0.f;
            v.eps = a.eps_post[idx];
            v.dps = a.dpost_std ? a.dpost_std[idx] : 0.f;
            v.pstd = a.post_std[idx];
        }
        return v;
    };
    auto load_sv = [&](int t) {
        Svv v;
        v.svq4 = floatx4{1.f, 1.f, 1.f, 1.f};
        v.svx4 = v.svq4;
        v.gr4 = v.gz4 = v.gn4 = v.gh4 = v.hp4 = v.dft4 = z4;
        if (rok) {
            const size_t tb = (size_t)t * a.B;
            v.svq4 = ks_row4(a.sv_q + (tb + grow) * a.Hd + fcol0, nhd);
            v.svx4 = ks_row4(a.sv_x + (tb + grow) * a.Be + fcol0, nbe);
            const float* gg = a.sv_gates + (tb + grow) * 4 * a.Be + fcol0;
            v.gr4 = ks_row4(gg, nbe); v.gz4 = ks_row4(gg + a.Be, nbe); v.gn4 = ks_row4(gg + 2 * a.Be, nbe); v.gh4 = ks_row4(gg + 3 * a.Be, nbe);
            v.hp4 = t > 0 ? ks_row4(a.feat + (tb - a.B + grow) * F + fcol0, nbe) : ks_row4(a.init_belief + (size_t)grow * a.Be + fcol0, nbe);
            v.dft4 = ks_row4(a.dfeat + (tb + grow) * F + fcol0, nbe);
        }
        return v;
    };
    constexpr bool AHEAD = !GR;          // (the granule form has no registers to spare: it loads at the top of the step)
    B1v nb1{0.f, 0.f, 0.f, 0.f, 0.f};
    Svv nsv{z4, z4, z4, z4, z4, z4, z4, z4};
    if constexpr (AHEAD) {
        nb1 = load_b1(a.T - 1, (int)threadIdx.x);
        nsv = load_sv(a.T - 1);
    }

    for (int t = a.T - 1; t >= 0; --t) {
        const size_t tb = (size_t)t * a.B;
        const int tid = bd_tid();
        float* xb = xbase + (size_t)(GR ? 0 : (t & 1)) * kb_.total;
        BD_KSTAMP(16);
        BD_KARGS_FRESH(ap);
        B1v cb1;
        Svv sv;
        if constexpr (AHEAD) {
            cb1 = nb1;
            sv = nsv;
            if (t > 0) {
                nb1 = load_b1(t - 1, tid);
                nsv = load_sv(t - 1);
            }
        } else {
            cb1 = load_b1(t, tid);
            sv = load_sv(t);
        }
        // ---- B1: through the sample / softplus into (mean, raw) (every member, elementwise) ----
        if (tid < 16 * Kb_s * 16) {          // (one pass: see the forward kernels)
            const int i = tid;
            const int r = i / (Kb_s * 16), k = i - r * (Kb_s * 16);
            const int gr = row0 + r;
            float dm = 0.f, dr = 0.f;
            if (gr < a.B && k < a.S) {
                const B1v v = cb1;
                float dst = v.dfs;
                for (int pt = 0; pt < nparts; ++pt) dst += ds_plain[pt * dsp_stride + r * a.S + k];
                dm = dst + v.dpm;
                const float dsd = dst * v.eps + v.dps;
                dr = dsd * one_minus_exp_neg(v.pstd - a.min_std);
                if (lead) {
                    a.d_q2_out[(tb + gr) * 2 * a.S + k] = dm;
                    a.d_q2_out[(tb + gr) * 2 * a.S + a.S + k] = dr;
                }
            }
            dM[frag_idx(r, k)] = dm;
            dRaw[frag_idx(r, k)] = dr;
        }
        const floatx4 svq4 = sv.svq4, svx4 = sv.svx4, gr4 = sv.gr4, gz4 = sv.gz4, gn4 = sv.gn4, gh4 = sv.gh4, hp4 = sv.hp4, dft4 = sv.dft4;
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
            if (nbo < Kb_h) ks_emit<GR>(xb + kb_.q + (size_t)(nbo * C + c) * IMG, lane, mfmaT(w1[i], dq4, z4), epoch);
        }
        BD_KSTAMP(19);
        ks_handoff<GR>(flags, c, C, epoch, err, spin_limit, kErrBwd);
        BD_KSTAMP(20);
        // ---- B4: total d belief of block c, gate gradients: every wave ----
        RED4[wave * 64 + lane] = ks_reduce<GR>(xb + kb_.q + (size_t)(c * C) * IMG, IMG, wave, kWaves, C, lane, epoch, err, spin_limit, kErrBwd, dead);
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
                float* dst = xb + kb_.g + ((size_t)(nbo * C + c) * 2) * IMG;
                ks_emit<GR>(dst, lane, DX, epoch); ks_emit<GR>(dst + IMG, lane, DH, epoch);
            }
        }
        BD_KSTAMP(22);
        ks_handoff<GR>(flags, c, C, epoch, err, spin_limit, kErrBwd);
        BD_KSTAMP(23);
        // ---- B6: d embed pre-activation of block c, the carry's W_hh^T term: every wave ----
        {
            const int g = wave & 1, quarter = wave >> 1;
            REDB[wave * 64 + lane] = ks_reduce<GR>(xb + kb_.g + ((size_t)(c * C) * 2 + g) * IMG, (size_t)2 * IMG, quarter, 4, C, lane, epoch,
                                                   err, spin_limit, kErrBwd, dead);
        }
        lds_barrier();
        floatx4 de4;
        {
            floatx4 DX = z4, DH = z4;
            for (int qd = 0; qd < 4; ++qd) { DX += REDB[(2 * qd) * 64 + lane]; DH += REDB[(2 * qd + 1) * 64 + lane]; }
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
        if (spart == 0) ks_emit<GR>(xb + kb_.s + (size_t)(c * 8 + sblk) * IMG, lane, mfmaT(wes, de4, z4), epoch);
        BD_KSTAMP(25);
        ks_handoff<GR>(flags, c, C, epoch, err, spin_limit, kErrBwd);
        BD_KSTAMP(26);
        BD_KARGS_FRESH(ap);
        // ---- B8: d posterior_state_t (through the nonterminal mask of this step's input); wave group `spart` sums its members ----
        if (spart < nparts) {
            const floatx4 v = ks_reduce<GR>(xb + kb_.s + (size_t)sblk * IMG, (size_t)8 * IMG, spart, nparts, C, lane, epoch, err, spin_limit,
                                            kErrBwd, dead);
            const float mk = rok ? (a.nonterm ? a.nonterm[tb + grow] : 1.f) : 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int col = sblk * 16 + 4 * (lane >> 4) + i;
                if (col < a.S) ds_plain[spart * dsp_stride + frow * a.S + col] = v[i] * mk;
            }
        }
        lds_barrier();
        BD_KSTAMP(27);
    }
#undef a
}

size_t ksplit_ws_floats_per_tile(int C) {
    const KsBuf r1(C, false), r2(C, true);
    const size_t x = r1.copies * r1.total, y = r2.copies * r2.total;
    return x > y ? x : y;
}

int& ksplit_mode() {               // -1: as the environment says, 0: off (round-1 cluster form), 1: K-split with R1 hand-offs,
    static int m = -1;             //  2: K-split with granule (R2) hand-offs, 3: as 1 with the forward GRU split by output columns
    return m;
}
static int ksplit_form() {         // 0 / 1 / 2 / 3 as above, environment resolved (BD_OBS_KSPLIT, default 3)
    static const char* e = getenv("BD_OBS_KSPLIT");
    if (ksplit_mode() >= 0) return ksplit_mode();
    if (e && e[0] >= '0' && e[0] <= '3') return e[0] - '0';
    return 3;
}

bool ksplit_ok(int Be, int S, int A, int Hd, int C) {
    if (ksplit_form() == 0) return false;
    return C == cdiv(Be, 16) && C <= 2 * kWaves && cdiv(Hd, 16) <= C && cdiv(S, 16) <= kKsMaxS && cdiv(A, 16) <= kKsMaxA &&
           S <= kHeadMaxN && 16 * S <= kThreads && 2 * cdiv(S, 16) <= kWaves;
}

static size_t ks_lds_fwd(int S, int A) {
    const int Kb_s = cdiv(S, 16), Kb_a = cdiv(A, 16);
    const int nparts = kWaves / (2 * Kb_s);
    return ((size_t)(Kb_s + Kb_a) * kFragFloats + ((16 * S + 3) & ~3) + 2 * kWaves * 256 + (size_t)nparts * 2 * 16 * Kb_s * 16) * sizeof(float);
}
static size_t ks_lds_fwd_ns(int S, int A, int Be) {       // + x and h fragment tiles + the waves' gate partials
    return ks_lds_fwd(S, A) + ((size_t)2 * cdiv(Be, 16) * kFragFloats + (size_t)kWaves * 4 * 256) * sizeof(float);
}
static size_t ks_lds_bwd(int S) {
    const int Kb_s = cdiv(S, 16);
    const int nparts = kWaves / Kb_s;
    return ((size_t)2 * Kb_s * kFragFloats + (size_t)nparts * ((16 * S + 3) & ~3) + 2 * kWaves * 256) * sizeof(float);
}

// zero what the form polls: the member flags (R1) or the whole exchange buffer, whose tags restart at 1 (R2)
static int ks_reset(float* ws, int C, int tiles, bool gr, hipStream_t stream, const char* who) {
    const size_t floats = gr ? cluster_ws_flag_floats(tiles) : cluster_ws_flag_floats(tiles);
    if (hipMemsetAsync(ws, 0, floats * sizeof(float), stream) != hipSuccess) return fail("%s: memset failed", who);
    if (gr) {
        const KsBuf kb(C, true);
        float* x = ws + cluster_ws_header_floats(tiles);
        if (hipMemsetAsync(x, 0, (size_t)tiles * kb.copies * kb.total * sizeof(float), stream) != hipSuccess)
            return fail("%s: memset failed", who);
    }
    return 0;
}

template <bool GR>
static int launch_kfwd(const bd_observe_fwd_args* a, float* ws, int C, int tiles, hipStream_t stream) {
    if (allow_big_lds(observe_kfwd_kernel<GR>)) return -1;
    const size_t dyn = launch_lds(observe_kfwd_kernel<GR>, ks_lds_fwd(a->S, a->A), "bd_observe_forward_cluster");
    if (!dyn) return -1;
    if (ks_reset(ws, C, tiles, GR, stream, "bd_observe_forward_cluster")) return -1;
    hipLaunchKernelGGL(observe_kfwd_kernel<GR>, dim3(tiles * C), dim3(kThreads), dyn, stream, *a, ws, C, tiles, cluster_spin_limit());
    BD_CHECK_LAUNCH("bd_observe_forward_cluster");
    return 0;
}
template <bool GR>
static int launch_kbwd(const bd_observe_bwd_args* a, float* ws, int C, int tiles, hipStream_t stream) {
    if (allow_big_lds(observe_kbwd_kernel<GR>)) return -1;
    const size_t dyn = launch_lds(observe_kbwd_kernel<GR>, ks_lds_bwd(a->S), "bd_observe_backward_cluster");
    if (!dyn) return -1;
    if (ks_reset(ws, C, tiles, GR, stream, "bd_observe_backward_cluster")) return -1;
    hipLaunchKernelGGL(observe_kbwd_kernel<GR>, dim3(tiles * C), dim3(kThreads), dyn, stream, *a, ws, C, tiles, cluster_spin_limit());
    BD_CHECK_LAUNCH("bd_observe_backward_cluster");
    return 0;
}

static int launch_kfwd_ns(const bd_observe_fwd_args* a, float* ws, int C, int tiles, hipStream_t stream) {
    if (allow_big_lds(observe_kfwd_ns_kernel)) return -1;
    const size_t dyn = launch_lds(observe_kfwd_ns_kernel, ks_lds_fwd_ns(a->S, a->A, a->Be), "bd_observe_forward_cluster");
    if (!dyn) return -1;
    if (ks_reset(ws, C, tiles, false, stream, "bd_observe_forward_cluster")) return -1;
    hipLaunchKernelGGL(observe_kfwd_ns_kernel, dim3(tiles * C), dim3(kThreads), dyn, stream, *a, ws, C, tiles, cluster_spin_limit());
    BD_CHECK_LAUNCH("bd_observe_forward_cluster");
    return 0;
}

int launch_observe_kfwd(const bd_observe_fwd_args* a, float* ws, int C, int tiles, hipStream_t stream) {
    if (ksplit_form() == 3) return launch_kfwd_ns(a, ws, C, tiles, stream);
    return ksplit_form() == 2 ? launch_kfwd<true>(a, ws, C, tiles, stream) : launch_kfwd<false>(a, ws, C, tiles, stream);
}
int launch_observe_kbwd(const bd_observe_bwd_args* a, float* ws, int C, int tiles, hipStream_t stream) {
    return ksplit_form() == 2 ? launch_kbwd<true>(a, ws, C, tiles, stream) : launch_kbwd<false>(a, ws, C, tiles, stream);
}

}  // namespace bd

#ifdef BD_STAMPS
extern "C" int bd_debug_kstamps(unsigned long long* out64) {
    return hipMemcpyFromSymbol(out64, HIP_SYMBOL(bd::g_kstamps), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : -1;
}
#endif
