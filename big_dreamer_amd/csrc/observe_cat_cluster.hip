// observe_cat_cluster.hip -- the Categorical RSSM observe scan (TransitionModel.forward with
// latent_distribution="Categorical", src/models.py:191-299 + CategoricalBeliefModel :76-117) and its backward with a
// CLUSTER of workgroups per 16-row tile.  BASELINE configs[4] has batch 100 per GPU = 7 row tiles: one workgroup per
// tile (scan_cat.hip) leaves 249 of 256 CUs idle for 8.7 ms of a 14-35 ms step.
//
// Cm workgroups (one per CU) share a tile; what a member owns:
//   * GRU column blocks c, c + Cm (forward: the cell; backward: its W_ih^T / W_hh^T dgrad), K split over the waves;
//   * D / Cm FACTORS of the posterior head: their logit columns (forward: hidden -> logits, sample; backward: the
//     embed-layer dgrad into those columns, the straight-through Jacobian, and the K-slice of d hidden = d logits W2).
// Everything 200 wide (embed, posterior hidden, d belief + gate gradients) is computed redundantly by every member.
// Two hand-offs per time step, both through L2 with the R1 protocol of bd_cluster.h:
//   forward : (1) all-gather of the new belief h'   (2) all-gather of the sampled class indices (16 x D bytes as words)
//   backward: (1) all-REDUCE of d(posterior hidden) -- every member publishes its K-slice partial [16 x Hd], every member
//                 sums the Cm partials in member order (bit-identical on all members)
//             (2) all-gather of [belief-gradient carry | d embed pre-activation], as the Gaussian cluster scan.
// Epoch = 2 * step + phase (monotonic, never 0); payloads are double-buffered by step parity.
#include "bd_cluster.h"
#include "bd_categorical.h"

namespace bd {

#ifdef BD_STAMPS
__device__ unsigned long long g_ccstamps[64];
#define BD_CCSTAMP(slot)                                                                                   \
    do {                                                                                                   \
        if (blockIdx.x == 0 && threadIdx.x == 0 && t == 5) g_ccstamps[slot] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define BD_CCSTAMP(slot)
#endif

constexpr int kCatOwnBlocks = 8;      // logit column blocks a member owns at most (128 columns)

__device__ __forceinline__ void st_sc1_u32(unsigned* p, unsigned v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_sc1_x4(float* p, floatx4 v) {     // two 8-byte write-through stores
    unsigned long long lo = (unsigned long long)__float_as_uint(v[0]) | ((unsigned long long)__float_as_uint(v[1]) << 32);
    unsigned long long hi = (unsigned long long)__float_as_uint(v[2]) | ((unsigned long long)__float_as_uint(v[3]) << 32);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p) + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// what a member owns
struct CatOwn {
    int nfm, f0, ncols, col0, nbl, nb0;
    __host__ __device__ CatOwn(int D, int C, int Cm, int c)
        : nfm(D / Cm), f0(c * (D / Cm)), ncols((D / Cm) * C), col0(c * (D / Cm) * C), nbl((D / Cm) * C / 16),
          nb0(c * (D / Cm) * C / 16) {}
};

// dense one-hot values of the factors [f0, f0 + nf) of 16 tile rows (dst rows global, row stride ld)
__device__ __forceinline__ void write_onehot_range(const CatGeo& g, const int* __restrict__ sidx_l, const float* __restrict__ sw_l,
                                                   const float* __restrict__ scale_l, float* __restrict__ dst_row0, size_t ld,
                                                   int rows_valid, int f0, int nf) {
    const bool vec = (g.C & 3) == 0 && (ld & 3) == 0 && (((uintptr_t)dst_row0) & 15) == 0;
    for (int i = bd_tid(); i < 16 * nf; i += blockDim.x) {
        const int row = i / nf, f = f0 + (i - row * nf);
        if (row >= rows_valid) continue;
        const int hot = sidx_l[row * g.D + f];
        float v = sw_l[row * g.D + f];
        if (scale_l) v *= scale_l[row];
        float* p = dst_row0 + (size_t)row * ld + f * g.C;
        if (vec) {
            for (int cc = 0; cc < g.C; cc += 4) {
                floatx4 o = floatx4{0.f, 0.f, 0.f, 0.f};
                if ((hot & ~3) == cc) o[hot & 3] = v;
                *reinterpret_cast<floatx4*>(p + cc) = o;
            }
        } else {
            for (int cc = 0; cc < g.C; ++cc) p[cc] = (cc == hot) ? v : 0.f;
        }
    }
}

// ---- forward ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void observe_cat_cfwd_kernel(bd_observe_cat_fwd_args a_, float* __restrict__ ws, int Cm,
                                                                    int tiles, unsigned spin_limit) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    BD_KARGS(bd_observe_cat_fwd_args, ap);
#define a (*ap)
    const CatGeo g(a.D, a.C);
    const int Kb_h = cdiv(a.Be, 16), Kb_a = cdiv(a.A, 16), Kb_hd = cdiv(a.Hd, 16);
    const int tile = blockIdx.x / Cm, c = blockIdx.x - tile * Cm;
    const int row0 = tile * 16;
    const int S = g.S, F = a.Be + S;
    const int nh = Kb_h * kFragFloats, nhd = Kb_hd * kFragFloats;
    const int rows_valid = a.B - row0 < 16 ? a.B - row0 : 16;
    const int Nb = Kb_h;
    const CatOwn own(a.D, a.C, Cm, c);
    const CatFull gl(own.nfm, a.C);                // image of the member's OWN logit columns (local factor index)
    float* h_cur = smem;
    float* h_nxt = h_cur + nh;
    float* xf = h_nxt + nh;
    float* qf = xf + nh;
    float* af = qf + nhd;
    float* xs = af + Kb_a * kFragFloats;          // [16][Be] gathered W_es s~
    float* lg = xs + 16 * a.Be;                   // own logits (swizzled image)
    float* sw_l = lg + gl.image_floats();         // [16][D]
    float* mrow = sw_l + 16 * g.D;                // [16]
    int* sidx_l = reinterpret_cast<int*>(mrow + 16);            // [16][D]
    float* scratch = reinterpret_cast<float*>(sidx_l + 16 * g.D);   // GRU partials | logit partials
    floatx4* __restrict__ G4 = reinterpret_cast<floatx4*>(scratch);

    unsigned* flags = reinterpret_cast<unsigned*>(ws) + tile * kMaxCluster;
    unsigned* err = reinterpret_cast<unsigned*>(ws) + tiles * kMaxCluster;
    const size_t per_tile = (size_t)2 * nh + (size_t)2 * 16 * g.D;
    float* xbuf_h = ws + cluster_ws_header_floats(tiles) + (size_t)tile * per_tile;     // [2][nh]
    unsigned* xbuf_s = reinterpret_cast<unsigned*>(xbuf_h + 2 * nh);                    // [2][16][D]

    load_tile_concat<1>(h_cur, Kb_h, row0, a.B, a.init_belief, a.Be, a.Be, nullptr, 0, 0);
    state_to_indices(g, a.init_state, (size_t)S, row0, a.B, sidx_l, sw_l);
    lds_barrier();

    const bool lead = (c == 0);

    for (int t = 0; t < a.T; ++t) {
        const size_t tb = (size_t)t * a.B;
        const int tid = bd_tid();
        const int lane = tid & 63, wave = bd_wave(tid);
        const floatx4* __restrict__ X4 = reinterpret_cast<const floatx4*>(xf) + lane;
        BD_CCSTAMP(0);
        BD_KARGS_FRESH(ap);
        // ---- A: mask, action fragments (every member) ----
        if (tid < 16) mrow[tid] = (a.nonterm && row0 + tid < a.B) ? a.nonterm[tb + row0 + tid] : 1.f;
        for (int i = tid; i < 16 * Kb_a * 16; i += blockDim.x) {
            const int r = i / (Kb_a * 16), k = i - r * (Kb_a * 16);
            const int grow = row0 + r;
            af[frag_idx(r, k)] = (grow < a.B && k < a.A) ? a.actions[(tb + grow) * a.A + k] : 0.f;
        }
        lds_barrier();
        BD_CCSTAMP(1);
        BD_KARGS_FRESH(ap);
        // ---- A2: W_es s~ as a gather (every member); the masked one-hot state for the weight gradient: own factors ----
        state_gather(g, a.w_embed_sT, a.Be, sidx_l, sw_l, mrow, xs);
        if (a.sv_s) write_onehot_range(g, sidx_l, sw_l, mrow, a.sv_s + (tb + row0) * S, (size_t)S, rows_valid, own.f0, own.nfm);
        lds_barrier();
        BD_CCSTAMP(2);
        BD_KARGS_FRESH(ap);
        // ---- B: x = ELU(W_es s~ + W_ea a + b_e) (every member) ----
        {
            const Seg segs[1] = {{af, a.w_embed_a, Kb_a}};
            tile_linear_seg<1>(segs, a.b_embed, a.Be, [&](int nb, floatx4 acc) {
                const int col = nb * 16 + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 4 * (lane >> 4) + r, grow = row0 + row;
                    const bool ok = grow < a.B && col < a.Be;
                    const float v = ok ? elu(acc[r] + xs[row * a.Be + col]) : 0.f;
                    xf[acc_frag_off(nb, lane, r)] = v;
                    if (lead && a.sv_x && ok) a.sv_x[(tb + grow) * a.Be + col] = v;
                }
            });
        }
        lds_barrier();
        BD_CCSTAMP(3);
        BD_KARGS_FRESH(ap);
        // ---- C: GRU, this member's column blocks, K split over the waves ----
        const int my_nb = c + wave * Cm;
        const bool reducer = wave < kLocalBlocks && my_nb < Nb;
        float br = 0.f, bz = 0.f, bni = 0.f, bnh = 0.f;
        if (reducer) {
            const int col = my_nb * 16 + (lane & 15);
            if (col < a.Be) {
                br = a.b_ih[col] + a.b_hh[col];
                bz = a.b_ih[a.Be + col] + a.b_hh[a.Be + col];
                bni = a.b_ih[2 * a.Be + col];
                bnh = a.b_hh[2 * a.Be + col];
            }
        }
        {
            const floatx4* __restrict__ H4 = reinterpret_cast<const floatx4*>(h_cur) + lane;
#pragma unroll
            for (int bi = 0; bi < kLocalBlocks; ++bi) {
                const int nb = c + bi * Cm;
                if (nb < Nb) {
                    floatx4 R = floatx4{0.f, 0.f, 0.f, 0.f}, Z = R, NI = R, NH = R;
                    const size_t off = (size_t)nb * Kb_h * 64 + lane;
                    const floatx4* __restrict__ Wir = reinterpret_cast<const floatx4*>(a.w_ir) + off;
                    const floatx4* __restrict__ Wiz = reinterpret_cast<const floatx4*>(a.w_iz) + off;
                    const floatx4* __restrict__ Win = reinterpret_cast<const floatx4*>(a.w_in) + off;
                    const floatx4* __restrict__ Whr = reinterpret_cast<const floatx4*>(a.w_hr) + off;
                    const floatx4* __restrict__ Whz = reinterpret_cast<const floatx4*>(a.w_hz) + off;
                    const floatx4* __restrict__ Whn = reinterpret_cast<const floatx4*>(a.w_hn) + off;
                    for (int kb = wave; kb < Kb_h; kb += kWaves) {
                        const floatx4 ax = X4[kb * 64], ah = H4[kb * 64];
                        const floatx4 bir = Wir[kb * 64], biz = Wiz[kb * 64], bin = Win[kb * 64];
                        const floatx4 bhr = Whr[kb * 64], bhz = Whz[kb * 64], bhn = Whn[kb * 64];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            R = mfma16(ax[j], bir[j], R);
                            Z = mfma16(ax[j], biz[j], Z);
                            NI = mfma16(ax[j], bin[j], NI);
                            NH = mfma16(ah[j], bhn[j], NH);
                            R = mfma16(ah[j], bhr[j], R);
                            Z = mfma16(ah[j], bhz[j], Z);
                        }
                    }
                    floatx4* gp = G4 + ((wave * kLocalBlocks + bi) * 4) * 64 + lane;
                    gp[0] = R; gp[64] = Z; gp[128] = NI; gp[192] = NH;
                }
            }
        }
        lds_barrier();
        BD_CCSTAMP(4);
        float hn_keep[4] = {0.f, 0.f, 0.f, 0.f}, g_keep[4][4];
        if (reducer) {
            const int col = my_nb * 16 + (lane & 15);
            const bool okc = col < a.Be;
            floatx4 R = floatx4{br, br, br, br}, Z = floatx4{bz, bz, bz, bz};
            floatx4 NI = floatx4{bni, bni, bni, bni}, NH = floatx4{bnh, bnh, bnh, bnh};
            for (int w = 0; w < kWaves; ++w) {
                const floatx4* gp = G4 + ((w * kLocalBlocks + wave) * 4) * 64 + lane;
                R += gp[0]; Z += gp[64]; NI += gp[128]; NH += gp[192];
            }
            float* xb = xbuf_h + (size_t)(t & 1) * nh;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int grow = row0 + 4 * (lane >> 4) + r;
                const int off = acc_frag_off(my_nb, lane, r);
                const float rr = sigmoidf(R[r]), zz = sigmoidf(Z[r]);
                const float nn = tanh_act(NI[r] + rr * NH[r]);
                const float hn = (grow < a.B && okc) ? (1.f - zz) * nn + zz * h_cur[off] : 0.f;
                st_sc1(xb + off, hn);
                hn_keep[r] = hn;
                g_keep[r][0] = rr; g_keep[r][1] = zz; g_keep[r][2] = nn; g_keep[r][3] = NH[r];
            }
        }
        publish(flags + c, (unsigned)(2 * t + 1));
        if (reducer) {                                        // plain stores after the flag: they do not delay it
            const int col = my_nb * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int grow = row0 + 4 * (lane >> 4) + r;
                if (grow < a.B && col < a.Be) {
                    a.feat[(tb + grow) * F + col] = hn_keep[r];
                    if (a.sv_gates) {
                        float* gg = a.sv_gates + (tb + grow) * 4 * a.Be + col;
                        gg[0] = g_keep[r][0]; gg[a.Be] = g_keep[r][1]; gg[2 * a.Be] = g_keep[r][2]; gg[3 * a.Be] = g_keep[r][3];
                    }
                }
            }
        }
        BD_CCSTAMP(5);
        wait_all(flags, Cm, (unsigned)(2 * t + 1), err, spin_limit, kErrFwd);
        BD_CCSTAMP(6);
        gather_payload(xbuf_h + (size_t)(t & 1) * nh, h_nxt, nh);
        lds_barrier();
        BD_CCSTAMP(7);
        BD_KARGS_FRESH(ap);
        // ---- D: posterior hidden (every member, full width) ----
        {
            const Seg segs[1] = {{h_nxt, a.w_q1h, Kb_h}};
            tile_linear_pre<1, 1>(
                segs, a.b_q1, a.Hd,
                [&](int, int nb) {           // hoisted embedding projection: fetched before the contraction
                    Pre4 p;
                    const int col = nb * 16 + (lane & 15);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int grow = row0 + 4 * (lane >> 4) + r;
                        p.v[r] = (grow < a.B && col < a.Hd) ? a.pre_emb[(tb + grow) * a.Hd + col] : 0.f;
                    }
                    return p;
                },
                [&](int, int nb, floatx4 acc, const Pre4& p) {
                    const int col = nb * 16 + (lane & 15);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int grow = row0 + 4 * (lane >> 4) + r;
                        const bool ok = grow < a.B && col < a.Hd;
                        const float v = ok ? elu(acc[r] + p.v[r]) : 0.f;
                        qf[acc_frag_off(nb, lane, r)] = v;
                        if (lead && ok && a.sv_q) a.sv_q[(tb + grow) * a.Hd + col] = v;
                    }
                });
        }
        lds_barrier();
        BD_CCSTAMP(8);
        BD_KARGS_FRESH(ap);
        // ---- E: the member's own logit columns (K split over the waves), sample its factors ----
        {
            const floatx4* __restrict__ Q4 = reinterpret_cast<const floatx4*>(qf) + lane;
            const floatx4* __restrict__ W4 = reinterpret_cast<const floatx4*>(a.w_q2) + (size_t)own.nb0 * Kb_hd * 64 + lane;
            floatx4 acc[kCatOwnBlocks];
#pragma unroll
            for (int bi = 0; bi < kCatOwnBlocks; ++bi) acc[bi] = floatx4{0.f, 0.f, 0.f, 0.f};
            for (int kb = wave; kb < Kb_hd; kb += kWaves) {
                const floatx4 a4 = Q4[kb * 64];
                floatx4 w4[kCatOwnBlocks];
#pragma unroll
                for (int bi = 0; bi < kCatOwnBlocks; ++bi)
                    if (bi < own.nbl) w4[bi] = W4[((size_t)bi * Kb_hd + kb) * 64];
#pragma unroll
                for (int bi = 0; bi < kCatOwnBlocks; ++bi)
                    if (bi < own.nbl) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[bi] = mfma16(a4[j], w4[bi][j], acc[bi]);
                    }
            }
#pragma unroll
            for (int bi = 0; bi < kCatOwnBlocks; ++bi)
                if (bi < own.nbl) G4[(wave * own.nbl + bi) * 64 + lane] = acc[bi];
        }
        lds_barrier();
        for (int bi = wave; bi < own.nbl; bi += kWaves) {
            const int colc = bi * 16 + (lane & 15), col = own.col0 + colc;
            const float b = a.b_q2[col];
            floatx4 r4 = floatx4{b, b, b, b};
            for (int w = 0; w < kWaves; ++w) r4 += G4[(w * own.nbl + bi) * 64 + lane];
            const int fl = colc / g.C, cc = colc - fl * g.C;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * (lane >> 4) + r;
                lg[gl.addr(row, fl, cc)] = r4[r];
                if (row < rows_valid) a.post_logits[(tb + row0 + row) * S + col] = r4[r];
            }
        }
        lds_barrier();
        BD_CCSTAMP(9);
        {
            unsigned* xs_ = xbuf_s + (size_t)(t & 1) * 16 * g.D;
            for (int i = tid; i < 16 * own.nfm; i += blockDim.x) {
                const int row = i / own.nfm, fl = i - row * own.nfm;
                int arg = 0;
                if (row < rows_valid) {
                    const float* qrow = a.q_post + (tb + row0 + row) * S + (own.f0 + fl) * g.C;
                    arg = g.C == 32 ? cat_sample_reg<32>(gl, lg, qrow, row, fl) : cat_sample_any(gl, lg, qrow, row, fl);
                    a.sidx[(tb + row0 + row) * g.D + own.f0 + fl] = (unsigned char)arg;
                }
                st_sc1_u32(xs_ + row * g.D + own.f0 + fl, (unsigned)arg);
            }
        }
        publish(flags + c, (unsigned)(2 * t + 2));
        BD_CCSTAMP(10);
        wait_all(flags, Cm, (unsigned)(2 * t + 2), err, spin_limit, kErrFwd);
        BD_CCSTAMP(11);
        {
            const float* src = reinterpret_cast<const float*>(xbuf_s + (size_t)(t & 1) * 16 * g.D);
            for (int i = tid * 2; i < 16 * g.D; i += blockDim.x * 2) {
                const unsigned long long u = ld_sc1_u64(src + i);
                sidx_l[i] = (int)(unsigned)u;
                sidx_l[i + 1] = (int)(unsigned)(u >> 32);
                sw_l[i] = i / g.D < rows_valid ? 1.f : 0.f;
                sw_l[i + 1] = (i + 1) / g.D < rows_valid ? 1.f : 0.f;
            }
        }
        lds_barrier();
        BD_CCSTAMP(12);
        // the dense one-hot state for the heads / weight gradients: own factors
        write_onehot_range(g, sidx_l, sw_l, nullptr, a.feat + (tb + row0) * F + a.Be, (size_t)F, rows_valid, own.f0, own.nfm);
        float* tmp = h_cur; h_cur = h_nxt; h_nxt = tmp;
        BD_CCSTAMP(13);
    }
#undef a
}

// ---- backward --------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void observe_cat_cbwd_kernel(bd_observe_cat_bwd_args a_, float* __restrict__ ws, int Cm,
                                                                    int tiles, unsigned spin_limit) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    BD_KARGS(bd_observe_cat_bwd_args, ap);
#define a (*ap)
    const CatGeo g(a.D, a.C);
    const int Kb_h = cdiv(a.Be, 16), Kb_hd = cdiv(a.Hd, 16), Kb_S = cdiv(g.S, 16);
    const int tile = blockIdx.x / Cm, c = blockIdx.x - tile * Cm;
    const int row0 = tile * 16;
    const int S = g.S, F = a.Be + S;
    const int nh = Kb_h * kFragFloats, nhd = Kb_hd * kFragFloats;
    const int rows_valid = a.B - row0 < 16 ? a.B - row0 : 16;
    const int Nb = Kb_h;
    const CatOwn own(a.D, a.C, Cm, c);
    const CatGeo go(own.nfm, a.C);                 // the member's own factors as a one-chunk geometry (ld = ncols + 8)
    float* dhc = smem;                             // [dhc | dE] contiguous: one payload of 2*nh floats
    float* dE = dhc + nh;
    float* dR = dE + nh;
    float* dZ = dR + nh;
    float* dNI = dZ + nh;
    float* dNH = dNI + nh;
    float* dQ = dNH + nh;                          // Kb_hd
    float* pl = dQ + nhd;                          // own columns: g, then d logits (swizzled image)
    float* lgs = pl + go.image_floats();           // own logits (swizzled image)
    float* dLf = lgs + go.image_floats();          // own d logits as a fragment tile (nbl K blocks)
    float* mrow = dLf + own.nbl * kFragFloats;     // [16]
    float* scratch = mrow + 16;
    floatx4* __restrict__ G4 = reinterpret_cast<floatx4*>(scratch);

    unsigned* flags = reinterpret_cast<unsigned*>(ws) + tile * kMaxCluster;
    unsigned* err = reinterpret_cast<unsigned*>(ws) + tiles * kMaxCluster;
    const size_t per_tile = (size_t)2 * (2 * nh) + (size_t)2 * Cm * nhd;
    float* xbuf_c = ws + cluster_ws_header_floats(tiles) + (size_t)tile * per_tile;     // [2][2*nh]   carry | d embed
    float* xbuf_q = xbuf_c + 2 * (2 * nh);                                              // [2][Cm][nhd] d hidden partials

    for (int i = threadIdx.x; i < 2 * nh; i += blockDim.x) dhc[i] = 0.f;
    lds_barrier();

    const bool lead = (c == 0);
    unsigned epoch = 0;

    for (int t = a.T - 1; t >= 0; --t) {
        const size_t tb = (size_t)t * a.B;
        const int tid = bd_tid();
        const int lane = tid & 63, wave = bd_wave(tid);
        const bool have_carry = t + 1 < a.T;
        const int par = t & 1;
        BD_CCSTAMP(16);
        BD_KARGS_FRESH(ap);
        // mask of step t+1 (its input state is posterior_state_t * nonterminal_{t+1})
        if (tid < 16) mrow[tid] = (a.nonterm && have_carry && row0 + tid < a.B) ? a.nonterm[tb + a.B + row0 + tid] : 1.f;
        // ---- 1a: g = (d embed pre-activation of step t+1) W_es [own columns] * mask + heads' gradient; own logits staged ----
        {
            // heads' gradient for the reducer's accumulator: requested before the contraction
            float dst[4] = {0.f, 0.f, 0.f, 0.f};
            for (int bi = wave; bi < own.nbl; bi += kWaves) {      // (nbl <= kWaves: at most one block per wave)
                const int colc = bi * 16 + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 4 * (lane >> 4) + r;
                    dst[r] = row < rows_valid ? a.dfeat[(tb + row0 + row) * F + a.Be + own.col0 + colc] : 0.f;
                }
            }
            for (int i = tid; i < 16 * own.ncols; i += blockDim.x) {
                const int row = i / own.ncols, colc = i - row * own.ncols;
                lgs[go.addr_col(row, colc)] = row < rows_valid ? a.post_logits[(tb + row0 + row) * S + own.col0 + colc] : 0.f;
            }
            if (have_carry) {
                const floatx4* __restrict__ E4 = reinterpret_cast<const floatx4*>(dE) + lane;
                const floatx4* __restrict__ W4 = reinterpret_cast<const floatx4*>(a.wt_embed_s) + (size_t)own.nb0 * Kb_h * 64 + lane;
                floatx4 acc[kCatOwnBlocks];
#pragma unroll
                for (int bi = 0; bi < kCatOwnBlocks; ++bi) acc[bi] = floatx4{0.f, 0.f, 0.f, 0.f};
                for (int kb = wave; kb < Kb_h; kb += kWaves) {
                    const floatx4 a4 = E4[kb * 64];
                    floatx4 w4[kCatOwnBlocks];
#pragma unroll
                    for (int bi = 0; bi < kCatOwnBlocks; ++bi)
                        if (bi < own.nbl) w4[bi] = W4[((size_t)bi * Kb_h + kb) * 64];
#pragma unroll
                    for (int bi = 0; bi < kCatOwnBlocks; ++bi)
                        if (bi < own.nbl) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) acc[bi] = mfma16(a4[j], w4[bi][j], acc[bi]);
                        }
                }
#pragma unroll
                for (int bi = 0; bi < kCatOwnBlocks; ++bi)
                    if (bi < own.nbl) G4[(wave * own.nbl + bi) * 64 + lane] = acc[bi];
            }
            lds_barrier();
            for (int bi = wave; bi < own.nbl; bi += kWaves) {
                const int colc = bi * 16 + (lane & 15);
                floatx4 r4 = floatx4{0.f, 0.f, 0.f, 0.f};
                if (have_carry)
                    for (int w = 0; w < kWaves; ++w) r4 += G4[(w * own.nbl + bi) * 64 + lane];
                const int fl = colc / g.C, cc = colc - fl * g.C;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 4 * (lane >> 4) + r;
                    pl[go.addr(row, fl, cc)] = row < rows_valid ? r4[r] * mrow[row] + dst[r] : 0.f;
                }
            }
        }
        lds_barrier();
        BD_CCSTAMP(17);
        // ---- 1b: straight-through Jacobian per own (row, factor), + direct logit gradient (KL); out to HBM and into dLf ----
        {
            const float* dextra = a.dpost_logits ? a.dpost_logits + (tb + row0) * S + own.col0 : nullptr;
            float* dout = a.d_q2_out + (tb + row0) * S + own.col0;
            if (g.C == 32) {
                for (int i = tid; i < 16 * own.nfm * 4; i += blockDim.x) {
                    const int grp = i >> 2, quad = i & 3;
                    const int row = grp / own.nfm, fl = grp - row * own.nfm;
                    const size_t gi = (size_t)row * S + fl * 32;
                    cat_jacobian_quad32(go, lgs, pl, row, fl, quad, row < rows_valid, dextra ? dextra + gi : nullptr, dout + gi, dLf);
                }
            } else {
                for (int i = tid; i < 16 * own.nfm; i += blockDim.x) {
                    const int row = i / own.nfm, fl = i - row * own.nfm;
                    const size_t gi = (size_t)row * S + fl * g.C;
                    cat_jacobian(go, lgs, pl, row, fl);
                    for (int cc = 0; cc < g.C; ++cc) {
                        float v = 0.f;
                        if (row < rows_valid) {
                            v = pl[go.addr(row, fl, cc)] + (dextra ? dextra[gi + cc] : 0.f);
                            dout[gi + cc] = v;
                        }
                        dLf[frag_idx(row, fl * g.C + cc)] = v;
                    }
                }
            }
        }
        lds_barrier();
        BD_CCSTAMP(18);
        BD_KARGS_FRESH(ap);
        // ---- 1c: this member's K-slice of d hidden = d logits W2, published; all-reduce over the members ----
        ++epoch;
        {
            const floatx4* __restrict__ A4 = reinterpret_cast<const floatx4*>(dLf) + lane;
            float* xq = xbuf_q + ((size_t)par * Cm + c) * nhd;
            for (int nb = wave; nb < Kb_hd; nb += kWaves) {
                const floatx4* __restrict__ W4 = reinterpret_cast<const floatx4*>(a.wt_q2) + ((size_t)nb * Kb_S + own.nb0) * 64 + lane;
                floatx4 a0 = floatx4{0.f, 0.f, 0.f, 0.f}, a1 = a0;
                for (int kb = 0; kb < own.nbl; ++kb) {
                    const floatx4 x4 = A4[kb * 64], w4 = W4[(size_t)kb * 64];
                    a0 = mfma16(x4[0], w4[0], a0);
                    a1 = mfma16(x4[1], w4[1], a1);
                    a0 = mfma16(x4[2], w4[2], a0);
                    a1 = mfma16(x4[3], w4[3], a1);
                }
                st_sc1_x4(xq + ((size_t)nb * 64 + lane) * 4, a0 + a1);
            }
        }
        publish(flags + c, epoch);
        BD_CCSTAMP(19);
        wait_all(flags, Cm, epoch, err, spin_limit, kErrBwd);
        BD_CCSTAMP(20);
        for (int nb = wave; nb < Kb_hd; nb += kWaves) {
            const int col = nb * 16 + (lane & 15);
            float svq[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int grow = row0 + 4 * (lane >> 4) + r;
                svq[r] = (grow < a.B && col < a.Hd) ? a.sv_q[(tb + grow) * a.Hd + col] : 1.f;
            }
            const float* base = xbuf_q + (size_t)par * Cm * nhd + ((size_t)nb * 64 + lane) * 4;
            floatx4 sum = floatx4{0.f, 0.f, 0.f, 0.f};
            for (int m0 = 0; m0 < Cm; m0 += 8) {
                unsigned long long u[8][2];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (m0 + j < Cm) {
                        u[j][0] = ld_sc1_u64(base + (size_t)(m0 + j) * nhd);
                        u[j][1] = ld_sc1_u64(base + (size_t)(m0 + j) * nhd + 2);
                    }
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (m0 + j < Cm) {
                        sum[0] += __uint_as_float((unsigned)u[j][0]);
                        sum[1] += __uint_as_float((unsigned)(u[j][0] >> 32));
                        sum[2] += __uint_as_float((unsigned)u[j][1]);
                        sum[3] += __uint_as_float((unsigned)(u[j][1] >> 32));
                    }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int grow = row0 + 4 * (lane >> 4) + r;
                float v = 0.f;
                if (grow < a.B && col < a.Hd) {
                    v = sum[r] * elu_grad_from_out(svq[r]);
                    if (lead) a.d_q1_pre[(tb + grow) * a.Hd + col] = v;
                }
                dQ[acc_frag_off(nb, lane, r)] = v;
            }
        }
        lds_barrier();
        BD_CCSTAMP(21);
        BD_KARGS_FRESH(ap);
        // ---- 3: total d belief_{t+1}, GRU gate gradients (every member, full width) ----
        {
            const Seg segs3[1] = {{dQ, a.wt_q1h, Kb_hd}};
            tile_linear_pre<1, 1>(
                segs3, nullptr, a.Be,
                [&](int, int nb) {
                    PreGate p;
                    const int col = nb * 16 + (lane & 15);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int grow = row0 + 4 * (lane >> 4) + r;
                        const bool ok = grow < a.B && col < a.Be;
                        const float* gg = a.sv_gates + (tb + grow) * 4 * a.Be + col;
                        p.r[r] = ok ? gg[0] : 0.f;
                        p.z[r] = ok ? gg[a.Be] : 0.f;
                        p.n[r] = ok ? gg[2 * a.Be] : 0.f;
                        p.hn[r] = ok ? gg[3 * a.Be] : 0.f;
                        p.hprev[r] = !ok ? 0.f : (t > 0 ? a.feat[(tb - a.B + grow) * F + col]
                                                        : a.init_belief[(size_t)grow * a.Be + col]);
                        p.dfeat[r] = ok ? a.dfeat[(tb + grow) * F + col] : 0.f;
                    }
                    return p;
                },
                [&](int, int nb, floatx4 acc, const PreGate& p) {
                    const int col = nb * 16 + (lane & 15);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int grow = row0 + 4 * (lane >> 4) + r;
                        const int off = acc_frag_off(nb, lane, r);
                        float vr = 0.f, vz = 0.f, vni = 0.f, vnh = 0.f, carry = 0.f;
                        if (grow < a.B && col < a.Be) {
                            const float dh = acc[r] + dhc[off] + p.dfeat[r];
                            const float rr = p.r[r], zz = p.z[r], nn = p.n[r], hn = p.hn[r];
                            const float dn = dh * (1.f - zz);
                            const float dz = dh * (p.hprev[r] - nn);
                            vni = dn * (1.f - nn * nn);
                            vnh = vni * rr;
                            vr = vni * hn * rr * (1.f - rr);
                            vz = dz * zz * (1.f - zz);
                            carry = dh * zz;
                            if (lead) {
                                float* gi = a.d_gi + (tb + grow) * 3 * a.Be + col;
                                float* gh = a.d_gh + (tb + grow) * 3 * a.Be + col;
                                gi[0] = vr; gi[a.Be] = vz; gi[2 * a.Be] = vni;
                                gh[0] = vr; gh[a.Be] = vz; gh[2 * a.Be] = vnh;
                            }
                        }
                        dR[off] = vr; dZ[off] = vz; dNI[off] = vni; dNH[off] = vnh;
                        dhc[off] = carry;
                    }
                });
        }
        lds_barrier();
        BD_CCSTAMP(22);
        BD_KARGS_FRESH(ap);
        // ---- 4: through W_ih / W_hh: this member's column blocks, K split over the waves ----
        ++epoch;
        const int my_nb = c + wave * Cm;
        const bool reducer = wave < kLocalBlocks && my_nb < Nb;
        float svx[4] = {1.f, 1.f, 1.f, 1.f};
        if (reducer) {
            const int col = my_nb * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int grow = row0 + 4 * (lane >> 4) + r;
                if (grow < a.B && col < a.Be) svx[r] = a.sv_x[(tb + grow) * a.Be + col];
            }
        }
        {
            const floatx4* __restrict__ R4 = reinterpret_cast<const floatx4*>(dR) + lane;
            const floatx4* __restrict__ Z4 = reinterpret_cast<const floatx4*>(dZ) + lane;
            const floatx4* __restrict__ I4 = reinterpret_cast<const floatx4*>(dNI) + lane;
            const floatx4* __restrict__ H4 = reinterpret_cast<const floatx4*>(dNH) + lane;
#pragma unroll
            for (int bi = 0; bi < kLocalBlocks; ++bi) {
                const int nb = c + bi * Cm;
                if (nb < Nb) {
                    floatx4 DX = floatx4{0.f, 0.f, 0.f, 0.f}, DH = DX;
                    const size_t off = (size_t)nb * Kb_h * 64 + lane;
                    const floatx4* __restrict__ Wir = reinterpret_cast<const floatx4*>(a.wt_ir) + off;
                    const floatx4* __restrict__ Wiz = reinterpret_cast<const floatx4*>(a.wt_iz) + off;
                    const floatx4* __restrict__ Win = reinterpret_cast<const floatx4*>(a.wt_in) + off;
                    const floatx4* __restrict__ Whr = reinterpret_cast<const floatx4*>(a.wt_hr) + off;
                    const floatx4* __restrict__ Whz = reinterpret_cast<const floatx4*>(a.wt_hz) + off;
                    const floatx4* __restrict__ Whn = reinterpret_cast<const floatx4*>(a.wt_hn) + off;
                    for (int kb = wave; kb < Kb_h; kb += kWaves) {
                        const floatx4 ar = R4[kb * 64], az = Z4[kb * 64], ai = I4[kb * 64], ah = H4[kb * 64];
                        const floatx4 bir = Wir[kb * 64], biz = Wiz[kb * 64], bin = Win[kb * 64];
                        const floatx4 bhr = Whr[kb * 64], bhz = Whz[kb * 64], bhn = Whn[kb * 64];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            DX = mfma16(ar[j], bir[j], DX);
                            DH = mfma16(ar[j], bhr[j], DH);
                            DX = mfma16(az[j], biz[j], DX);
                            DH = mfma16(az[j], bhz[j], DH);
                            DX = mfma16(ai[j], bin[j], DX);
                            DH = mfma16(ah[j], bhn[j], DH);
                        }
                    }
                    floatx4* gp = G4 + ((wave * kLocalBlocks + bi) * 2) * 64 + lane;
                    gp[0] = DX; gp[64] = DH;
                }
            }
        }
        lds_barrier();
        float de_keep[4] = {0.f, 0.f, 0.f, 0.f};
        if (reducer) {
            floatx4 DX = floatx4{0.f, 0.f, 0.f, 0.f}, DH = DX;
            for (int w = 0; w < kWaves; ++w) {
                const floatx4* gp = G4 + ((w * kLocalBlocks + wave) * 2) * 64 + lane;
                DX += gp[0]; DH += gp[64];
            }
            const int col = my_nb * 16 + (lane & 15);
            float* xb = xbuf_c + (size_t)par * (2 * nh);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int grow = row0 + 4 * (lane >> 4) + r;
                const int off = acc_frag_off(my_nb, lane, r);
                float de = 0.f, carry = 0.f;
                if (grow < a.B && col < a.Be) {
                    de = DX[r] * elu_grad_from_out(svx[r]);
                    carry = dhc[off] + DH[r];
                }
                st_sc1(xb + off, carry);
                st_sc1(xb + nh + off, de);
                de_keep[r] = de;
            }
        }
        publish(flags + c, epoch);
        if (reducer) {
            const int col = my_nb * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int grow = row0 + 4 * (lane >> 4) + r;
                if (grow < a.B && col < a.Be) a.d_embed_pre[(tb + grow) * a.Be + col] = de_keep[r];
            }
        }
        BD_CCSTAMP(23);
        wait_all(flags, Cm, epoch, err, spin_limit, kErrBwd);
        BD_CCSTAMP(24);
        gather_payload(xbuf_c + (size_t)par * (2 * nh), dhc, 2 * nh);
        lds_barrier();
        BD_CCSTAMP(25);
    }
#undef a
}

// Cluster size: the largest Cm in {16, 8, 4} whose tiles * Cm members fit `max_wgs` workgroups and that divides the head
// into whole, 16-column-aligned groups of factors; 0 = not applicable (caller uses scan_cat.hip).
static int pick_cat_cluster(int B, int Be, int D, int C, int max_wgs) {
    const int Nb = cdiv(Be, 16), tiles = cdiv(B, 16);
    if (max_wgs > 256) max_wgs = 256;
    for (int Cm = kMaxCluster; Cm >= 4; Cm >>= 1) {
        if (D % Cm != 0) continue;
        const int ncols = (D / Cm) * C;
        if (ncols % 16 != 0 || ncols / 16 > kCatOwnBlocks || ncols / 16 > kWaves) continue;
        if (Nb > kLocalBlocks * Cm) continue;
        if (tiles * Cm > max_wgs) continue;
        return Cm;
    }
    return 0;
}

// LDS scratch: split-K partials of the GRU blocks (forward: 4 gate accumulators per block, backward: 2), of the member's own
// logit column blocks, and the narrow-head split-K scratch of the tile primitives
static size_t cat_scratch_floats(int nbl, bool fwd) {
    size_t n = (size_t)kWaves * kLocalBlocks * (fwd ? 4 : 2) * 64 * 4;
    const size_t head = (size_t)kWaves * nbl * 64 * 4;
    if (head > n) n = head;
    return n > (size_t)kSplitScratchFloats ? n : (size_t)kSplitScratchFloats;
}

}  // namespace bd

extern "C" {
using namespace bd;

#ifdef BD_STAMPS
int bd_debug_ccstamps(unsigned long long* out64) {
    return hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_ccstamps), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : -1;
}
#endif

int bd_observe_cat_cluster_size(int B, int Be, int D, int C, int max_wgs) { return pick_cat_cluster(B, Be, D, C, max_wgs); }

size_t bd_observe_cat_cluster_ws_floats(int B, int Be, int Hd, int D, int Cm) {
    const int tiles = cdiv(B, 16);
    const size_t nh = (size_t)cdiv(Be, 16) * kFragFloats, nhd = (size_t)cdiv(Hd, 16) * kFragFloats;
    const size_t fwd = 2 * nh + (size_t)2 * 16 * D;
    const size_t bwd = 2 * (2 * nh) + (size_t)2 * Cm * nhd;
    return cluster_ws_header_floats(tiles) + (size_t)tiles * (fwd > bwd ? fwd : bwd);
}

#define BD_CATC_GEO(who)                                                                                                     \
    const CatGeo g(a->D, a->C);                                                                                              \
    BD_REQUIRE(a->D > 0 && a->C > 0 && g.ok(), who ": %d x %d latents unsupported", a->D, a->C);                             \
    BD_REQUIRE(a->Hd <= 16 * 2 * kWaves && a->Hd <= 256, who ": hidden width %d above %d", a->Hd, 16 * 2 * kWaves);          \
    BD_REQUIRE(Cm >= 2 && Cm <= kMaxCluster && a->D % Cm == 0 && ((a->D / Cm) * a->C) % 16 == 0 &&                           \
                   (a->D / Cm) * a->C / 16 <= kCatOwnBlocks && (a->D / Cm) * a->C / 16 <= kWaves &&                          \
                   cdiv(a->Be, 16) <= kLocalBlocks * Cm && cdiv(a->B, 16) * Cm <= 256,                                        \
               who ": cluster size %d does not fit B=%d Be=%d D=%d C=%d (bd_observe_cat_cluster_size)", Cm, a->B, a->Be, a->D, \
               a->C);                                                                                                        \
    BD_REQUIRE(ws && ws_floats >= bd_observe_cat_cluster_ws_floats(a->B, a->Be, a->Hd, a->D, Cm), who ": workspace too small")

int bd_observe_cat_forward_cluster(const bd_observe_cat_fwd_args* a, int Cm, float* ws, size_t ws_floats, void* stream) {
    BD_REQUIRE(a && a->T > 0 && a->B > 0 && a->Be > 0 && a->A > 0 && a->Hd > 0, "bd_observe_cat_forward_cluster: bad dims");
    BD_CATC_GEO("bd_observe_cat_forward_cluster");
    BD_REQUIRE(a->w_embed_sT && a->w_embed_a && a->b_embed && a->w_ir && a->w_iz && a->w_in && a->w_hr && a->w_hz && a->w_hn &&
                   a->b_ih && a->b_hh && a->w_q1h && a->b_q1 && a->w_q2 && a->b_q2, "bd_observe_cat_forward_cluster: missing weights");
    BD_REQUIRE(a->init_belief && a->init_state && a->actions && a->pre_emb && a->q_post, "bd_observe_cat_forward_cluster: missing inputs");
    BD_REQUIRE(a->feat && a->post_logits && a->sidx, "bd_observe_cat_forward_cluster: missing outputs");
    const int tiles = cdiv(a->B, 16);
    const int Kb_h = cdiv(a->Be, 16), Kb_a = cdiv(a->A, 16), Kb_hd = cdiv(a->Hd, 16);
    const CatFull gl(a->D / Cm, a->C);
    const size_t lds = ((size_t)(3 * Kb_h + Kb_hd + Kb_a) * kFragFloats + 16 * a->Be + gl.image_floats() + 2 * 16 * g.D + 16 +
                        cat_scratch_floats((a->D / Cm) * a->C / 16, true)) * sizeof(float);
    BD_REQUIRE(lds <= (size_t)kMaxLds, "bd_observe_cat_forward_cluster: needs %zu B of LDS", lds);
    if (allow_big_lds(observe_cat_cfwd_kernel)) return -1;
    const size_t dyn = launch_lds(observe_cat_cfwd_kernel, lds, "bd_observe_cat_forward_cluster");
    if (!dyn) return -1;
    if (hipMemsetAsync(ws, 0, cluster_ws_flag_floats(tiles) * sizeof(float), (hipStream_t)stream) != hipSuccess)
        return fail("bd_observe_cat_forward_cluster: memset failed");
    hipLaunchKernelGGL(observe_cat_cfwd_kernel, dim3(tiles * Cm), dim3(kThreads), dyn, (hipStream_t)stream, *a, ws, Cm, tiles,
                       cluster_spin_limit());
    BD_CHECK_LAUNCH("bd_observe_cat_forward_cluster");
    return 0;
}

int bd_observe_cat_backward_cluster(const bd_observe_cat_bwd_args* a, int Cm, float* ws, size_t ws_floats, void* stream) {
    BD_REQUIRE(a && a->T > 0 && a->B > 0 && a->Be > 0 && a->A > 0 && a->Hd > 0, "bd_observe_cat_backward_cluster: bad dims");
    BD_CATC_GEO("bd_observe_cat_backward_cluster");
    BD_REQUIRE(a->wt_embed_s && a->wt_ir && a->wt_iz && a->wt_in && a->wt_hr && a->wt_hz && a->wt_hn && a->wt_q1h && a->wt_q2,
               "bd_observe_cat_backward_cluster: missing weights");
    BD_REQUIRE(a->init_belief && a->feat && a->post_logits && a->sv_x && a->sv_gates && a->sv_q && a->dfeat,
               "bd_observe_cat_backward_cluster: missing forward tensors");
    BD_REQUIRE(a->d_embed_pre && a->d_gi && a->d_gh && a->d_q1_pre && a->d_q2_out, "bd_observe_cat_backward_cluster: missing outputs");
    const int tiles = cdiv(a->B, 16);
    const int Kb_h = cdiv(a->Be, 16), Kb_hd = cdiv(a->Hd, 16);
    const CatGeo go(a->D / Cm, a->C);
    const size_t lds = ((size_t)(6 * Kb_h + Kb_hd + (a->D / Cm) * a->C / 16) * kFragFloats + 2 * go.image_floats() + 16 +
                        cat_scratch_floats((a->D / Cm) * a->C / 16, false)) * sizeof(float);
    BD_REQUIRE(lds <= (size_t)kMaxLds, "bd_observe_cat_backward_cluster: needs %zu B of LDS", lds);
    if (allow_big_lds(observe_cat_cbwd_kernel)) return -1;
    const size_t dyn = launch_lds(observe_cat_cbwd_kernel, lds, "bd_observe_cat_backward_cluster");
    if (!dyn) return -1;
    if (hipMemsetAsync(ws, 0, cluster_ws_flag_floats(tiles) * sizeof(float), (hipStream_t)stream) != hipSuccess)
        return fail("bd_observe_cat_backward_cluster: memset failed");
    hipLaunchKernelGGL(observe_cat_cbwd_kernel, dim3(tiles * Cm), dim3(kThreads), dyn, (hipStream_t)stream, *a, ws, Cm, tiles,
                       cluster_spin_limit());
    BD_CHECK_LAUNCH("bd_observe_cat_backward_cluster");
    return 0;
}

}  // extern "C"
