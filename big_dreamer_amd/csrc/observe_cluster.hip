// observe_cluster.hip -- RSSM observe scan and its backward with a CLUSTER of workgroups per 16-row tile.
//
// Why: at batch=50 the scan has only 4 row tiles; one workgroup per tile (observe.hip) is MFMA-issue bound on 4 of
// the 256 CUs (the GRU's 2x(3*Be x Be) contraction is 85 % of the per-step work and is re-streamed from L2 by every
// step).  Here C workgroups (one per CU) share a tile:
//   * the GRU (forward) / its dgrad W_ih^T, W_hh^T (backward) is split by output column blocks over the members,
//     and inside a member by K blocks over the waves (partials meet in LDS);
//   * everything small (embed, posterior hidden/out, their backward) is computed redundantly by every member, so
//   * there is exactly ONE all-gather per time step: the new belief (forward), or d(embed pre-activation) and
//     the belief-gradient carry (backward), through an exchange buffer in HBM/L2.
//
// Hand-off protocol (cdna_hip_programming.md, Guideline 16 form R1 / MI355X_MICROARCH.md "Valid forms"):
//   producer: payload stored write-through (sc1: relaxed agent-scope atomic stores) -> every storing wave
//             s_waitcnt vmcnt(0) -> workgroup barrier -> ONE lane stores the member's flag (sc1) = epoch;
//   consumer: ONE wave polls the C flags with relaxed agent-scope loads (lane i polls member i) until all have
//             reached the epoch -> workgroup barrier -> EVERY load of the payload is an sc1 load to registers.
//   Epoch = step + 1 (monotonic within a launch, never 0); flags are zeroed by a memset node ahead of the launch;
//   the payload is double-buffered by step parity (a member can run at most one step ahead of the slowest).
//   Every spin is bounded: on timeout the member ORs its code into the STICKY error word and stops waiting (outputs are
//   then wrong, the launch still terminates).  The error word is NOT part of the per-launch header memset: it survives
//   later launches until bd_observe_cluster_status reads (and clears) it -- the engine reads it with every log fetch.
// Residency: tiles*C <= 256 workgroups of one per CU are co-resident on an otherwise idle MI355X.
#include "bd_cluster.h"

namespace bd {

#ifdef BD_STAMPS
__device__ unsigned long long g_cstamps[64];
#define BD_CSTAMP(slot)                                                                                  \
    do {                                                                                                 \
        if (blockIdx.x == 0 && threadIdx.x == 0 && t == 5) g_cstamps[slot] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define BD_CSTAMP(slot)
#endif
struct ObsDimsC {
    int Kb_h, Kb_s, Kb_a, Kb_hd;
    __host__ __device__ ObsDimsC(int Be, int S, int A, int Hd)
        : Kb_h(cdiv(Be, 16)), Kb_s(cdiv(S, 16)), Kb_a(cdiv(A, 16)), Kb_hd(cdiv(Hd, 16)) {}
};

// ---- forward ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void observe_cfwd_kernel(bd_observe_fwd_args a_, float* __restrict__ ws, int C,
                                                                int tiles, unsigned spin_limit) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    BD_KARGS(bd_observe_fwd_args, ap);      // first kernel argument: offset 0 of the kernarg segment
#define a (*ap)
    const ObsDimsC d(a.Be, a.S, a.A, a.Hd);
    const int tile = blockIdx.x / C, c = blockIdx.x - tile * C;
    const int row0 = tile * 16;
    const int F = a.Be + a.S;
    const int nh = d.Kb_h * kFragFloats;
    const int Nb = d.Kb_h;
    float* h_cur = smem;
    float* h_nxt = h_cur + nh;
    float* xf = h_nxt + nh;
    float* qf = xf + nh;
    float* sf = qf + d.Kb_hd * kFragFloats;
    float* af = sf + d.Kb_s * kFragFloats;
    float* s_plain = af + d.Kb_a * kFragFloats;          // [16][S]
    float* scratch = s_plain + 16 * a.S;                 // max(split-K scratch, GRU partials)
    floatx4* __restrict__ G4 = reinterpret_cast<floatx4*>(scratch);   // GRU partials [wave][blk][4][64]

    unsigned* flags = reinterpret_cast<unsigned*>(ws) + tile * kMaxCluster;
    unsigned* err = reinterpret_cast<unsigned*>(ws) + tiles * kMaxCluster;
    float* xbuf = ws + cluster_ws_header_floats(tiles) + (size_t)tile * 2 * nh;

    load_tile_concat<1>(h_cur, d.Kb_h, row0, a.B, a.init_belief, a.Be, a.Be, nullptr, 0, 0);
    for (int i = threadIdx.x; i < 16 * a.S; i += blockDim.x) {
        const int r = i / a.S, k = i - r * a.S;
        s_plain[i] = (row0 + r < a.B) ? a.init_state[(size_t)(row0 + r) * a.S + k] : 0.f;
    }
    lds_barrier();

    const bool lead = (c == 0);     // the member that writes the redundantly computed outputs

    for (int t = 0; t < a.T; ++t) {
        const size_t tb = (size_t)t * a.B;
        const int tid = bd_tid();                   // opaque: nothing thread-dependent leaves this step (bd_tid)
        const int lane = tid & 63, wave = bd_wave(tid);
        const floatx4* __restrict__ X4 = reinterpret_cast<const floatx4*>(xf) + lane;
        BD_CSTAMP(0);
        BD_KARGS_FRESH(ap);
        // ---- A: masked state / action fragments (every member) ----
        for (int i = tid; i < 16 * d.Kb_s * 16; i += blockDim.x) {
            const int r = i / (d.Kb_s * 16), k = i - r * (d.Kb_s * 16);
            const int grow = row0 + r;
            float v = 0.f;
            if (grow < a.B && k < a.S) {
                v = s_plain[r * a.S + k];
                if (a.nonterm) v *= a.nonterm[tb + grow];
                if (lead && a.sv_s) a.sv_s[(tb + grow) * a.S + k] = v;
            }
            sf[frag_idx(r, k)] = v;
        }
        for (int i = tid; i < 16 * d.Kb_a * 16; i += blockDim.x) {
            const int r = i / (d.Kb_a * 16), k = i - r * (d.Kb_a * 16);
            const int grow = row0 + r;
            af[frag_idx(r, k)] = (grow < a.B && k < a.A) ? a.actions[(tb + grow) * a.A + k] : 0.f;
        }
        lds_barrier();
        BD_CSTAMP(1);
        BD_KARGS_FRESH(ap);
        // ---- B: embed (every member, full width) ----
        {
            const Seg segs[2] = {{sf, a.w_embed_s, d.Kb_s}, {af, a.w_embed_a, d.Kb_a}};
            tile_linear_seg<2>(segs, a.b_embed, a.Be, [&](int nb, floatx4 acc) {
                const int col = nb * 16 + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int grow = row0 + 4 * (lane >> 4) + r;
                    const float v = elu(acc[r]);
                    xf[acc_frag_off(nb, lane, r)] = v;
                    if (lead && a.sv_x && grow < a.B && col < a.Be) a.sv_x[(tb + grow) * a.Be + col] = v;
                }
            });
        }
        lds_barrier();
        BD_CSTAMP(2);
        BD_KARGS_FRESH(ap);
        // ---- C: GRU, this member's column blocks, K split over the waves ----
        const int my_nb = c + wave * C;                      // wave bi reduces block bi
        const bool reducer = wave < kLocalBlocks && my_nb < Nb;
        float br = 0.f, bz = 0.f, bni = 0.f, bnh = 0.f;      // reducer's biases: in flight before the contraction
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
                const int nb = c + bi * C;
                if (nb < Nb) {
                    floatx4 R = floatx4{0.f, 0.f, 0.f, 0.f}, Z = R, NI = R, NH = R;
                    const size_t off = (size_t)nb * d.Kb_h * 64 + lane;
                    const floatx4* __restrict__ Wir = reinterpret_cast<const floatx4*>(a.w_ir) + off;
                    const floatx4* __restrict__ Wiz = reinterpret_cast<const floatx4*>(a.w_iz) + off;
                    const floatx4* __restrict__ Win = reinterpret_cast<const floatx4*>(a.w_in) + off;
                    const floatx4* __restrict__ Whr = reinterpret_cast<const floatx4*>(a.w_hr) + off;
                    const floatx4* __restrict__ Whz = reinterpret_cast<const floatx4*>(a.w_hz) + off;
                    const floatx4* __restrict__ Whn = reinterpret_cast<const floatx4*>(a.w_hn) + off;
                    for (int kb = wave; kb < d.Kb_h; kb += kWaves) {
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
                    floatx4* g = G4 + ((wave * kLocalBlocks + bi) * 4) * 64 + lane;
                    g[0] = R; g[64] = Z; g[128] = NI; g[192] = NH;
                }
            }
        }
        lds_barrier();
        BD_CSTAMP(3);
        float hn_keep[4] = {0.f, 0.f, 0.f, 0.f}, g_keep[4][4];
        if (reducer) {
            const int col = my_nb * 16 + (lane & 15);
            const bool okc = col < a.Be;
            floatx4 R = floatx4{br, br, br, br}, Z = floatx4{bz, bz, bz, bz};
            floatx4 NI = floatx4{bni, bni, bni, bni}, NH = floatx4{bnh, bnh, bnh, bnh};
            for (int w = 0; w < kWaves; ++w) {
                const floatx4* g = G4 + ((w * kLocalBlocks + wave) * 4) * 64 + lane;
                R += g[0]; Z += g[64]; NI += g[128]; NH += g[192];
            }
            float* xb = xbuf + (size_t)(t & 1) * nh;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int grow = row0 + 4 * (lane >> 4) + r;
                const int off = acc_frag_off(my_nb, lane, r);
                const float rr = sigmoidf(R[r]), zz = sigmoidf(Z[r]);
                const float nn = tanh_act(NI[r] + rr * NH[r]);
                const float hn = (grow < a.B && okc) ? (1.f - zz) * nn + zz * h_cur[off] : 0.f;
                st_sc1(xb + off, hn);                        // write-through payload, fragment order
                hn_keep[r] = hn;
                g_keep[r][0] = rr; g_keep[r][1] = zz; g_keep[r][2] = nn; g_keep[r][3] = NH[r];
            }
        }
        BD_CSTAMP(4);
        publish(flags + c, (unsigned)(t + 1));
        BD_CSTAMP(5);
        if (reducer) {                                        // plain stores after the flag: they do not delay it
            const int col = my_nb * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int grow = row0 + 4 * (lane >> 4) + r;
                if (grow < a.B && col < a.Be) {
                    a.feat[(tb + grow) * F + col] = hn_keep[r];
                    if (a.sv_gates) {
                        float* g = a.sv_gates + (tb + grow) * 4 * a.Be + col;
                        g[0] = g_keep[r][0]; g[a.Be] = g_keep[r][1]; g[2 * a.Be] = g_keep[r][2]; g[3 * a.Be] = g_keep[r][3];
                    }
                }
            }
        }
        BD_CSTAMP(6);
        wait_all(flags, C, (unsigned)(t + 1), err, spin_limit, kErrFwd);
        BD_CSTAMP(7);
        gather_payload(xbuf + (size_t)(t & 1) * nh, h_nxt, nh);
        lds_barrier();
        BD_CSTAMP(8);
        BD_KARGS_FRESH(ap);
        // ---- D: posterior hidden (every member, full width) ----
        {
            const Seg segs[1] = {{h_nxt, a.w_q1h, d.Kb_h}};
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
        BD_CSTAMP(9);
        BD_KARGS_FRESH(ap);
        // ---- E: posterior mean / std / sample (every member; split-K over waves) ----
        {
            const Seg2 segs[1] = {{qf, a.w_q2m, a.w_q2s, d.Kb_hd}};
            tile_dual_head_elem<1>(
                segs, a.b_q2, a.b_q2 + a.S, a.S, scratch,
                // posterior noise: fetched before the contraction
                [&](int row, int col) { return row0 + row < a.B ? a.eps_post[(tb + row0 + row) * a.S + col] : 0.f; },
                [&](int row, int col, float Mn, float Rw, float eps) {
                    const int grow = row0 + row;
                    float st = 0.f;
                    if (grow < a.B) {
                        const size_t i = (tb + grow) * a.S + col;
                        const float sd = softplusf(Rw) + a.min_std;
                        st = Mn + sd * eps;
                        if (lead) {
                            a.post_mean[i] = Mn;
                            a.post_std[i] = sd;
                            a.feat[(tb + grow) * F + a.Be + col] = st;
                        }
                    }
                    s_plain[row * a.S + col] = st;
                });
        }
        lds_barrier();
        BD_CSTAMP(10);
        float* tmp = h_cur; h_cur = h_nxt; h_nxt = tmp;
    }
#undef a
}

// ---- backward --------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void observe_cbwd_kernel(bd_observe_bwd_args a_, float* __restrict__ ws, int C,
                                                                int tiles, unsigned spin_limit) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    BD_KARGS(bd_observe_bwd_args, ap);      // first kernel argument: offset 0 of the kernarg segment
#define a (*ap)
    const ObsDimsC d(a.Be, a.S, a.A, a.Hd);
    const int tile = blockIdx.x / C, c = blockIdx.x - tile * C;
    const int row0 = tile * 16;
    const int F = a.Be + a.S;
    const int nh = d.Kb_h * kFragFloats, ns = d.Kb_s * kFragFloats;
    const int Nb = d.Kb_h;
    float* dhc = smem;                                  // [dE | dhc] are contiguous: one payload of 2*nh floats
    float* dE = dhc + nh;
    float* dR = dE + nh;
    float* dZ = dR + nh;
    float* dNI = dZ + nh;
    float* dNH = dNI + nh;
    float* dQ = dNH + nh;                               // Kb_hd
    float* dM = dQ + d.Kb_hd * kFragFloats;             // Kb_s
    float* dRaw = dM + ns;
    float* ds_plain = dRaw + ns;                        // [16][S]
    float* scratch = ds_plain + 16 * a.S;
    floatx4* __restrict__ G4 = reinterpret_cast<floatx4*>(scratch);   // phase-4 partials [wave][blk][2][64]

    unsigned* flags = reinterpret_cast<unsigned*>(ws) + tile * kMaxCluster;
    unsigned* err = reinterpret_cast<unsigned*>(ws) + tiles * kMaxCluster;
    float* xbuf = ws + cluster_ws_header_floats(tiles) + (size_t)tile * 2 * (2 * nh);

    for (int i = threadIdx.x; i < 2 * nh; i += blockDim.x) dhc[i] = 0.f;
    for (int i = threadIdx.x; i < 16 * a.S; i += blockDim.x) ds_plain[i] = 0.f;
    lds_barrier();

    const bool lead = (c == 0);
    unsigned epoch = 0;

    for (int t = a.T - 1; t >= 0; --t) {
        const size_t tb = (size_t)t * a.B;
        const int tid = bd_tid();                   // opaque: nothing thread-dependent leaves this step (bd_tid)
        const int lane = tid & 63, wave = bd_wave(tid);
        ++epoch;
        BD_KARGS_FRESH(ap);
        BD_CSTAMP(16);
        // ---- 1: through the sample / softplus into (mean, raw) (every member) ----
        for (int i = tid; i < 16 * d.Kb_s * 16; i += blockDim.x) {
            const int r = i / (d.Kb_s * 16), k = i - r * (d.Kb_s * 16);
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
        lds_barrier();
        BD_KARGS_FRESH(ap);
        BD_CSTAMP(17);
        // ---- 2: d q (every member) ----
        {
            const Seg segs[2] = {{dM, a.wt_q2m, d.Kb_s}, {dRaw, a.wt_q2s, d.Kb_s}};
            tile_linear_pre<1, 2>(
                segs, nullptr, a.Hd,
                [&](int, int nb) {
                    Pre4 p;
                    const int col = nb * 16 + (lane & 15);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int grow = row0 + 4 * (lane >> 4) + r;
                        p.v[r] = (grow < a.B && col < a.Hd) ? a.sv_q[(tb + grow) * a.Hd + col] : 1.f;
                    }
                    return p;
                },
                [&](int, int nb, floatx4 acc, const Pre4& p) {
                    const int col = nb * 16 + (lane & 15);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int grow = row0 + 4 * (lane >> 4) + r;
                        float v = 0.f;
                        if (grow < a.B && col < a.Hd) {
                            v = acc[r] * elu_grad_from_out(p.v[r]);
                            if (lead) a.d_q1_pre[(tb + grow) * a.Hd + col] = v;
                        }
                        dQ[acc_frag_off(nb, lane, r)] = v;
                    }
                });
        }
        lds_barrier();
        BD_KARGS_FRESH(ap);
        BD_CSTAMP(18);
        // ---- 3: total d belief_{t+1}, GRU gate gradients (every member, full width) ----
        {
            const Seg segs3[1] = {{dQ, a.wt_q1h, d.Kb_hd}};
            tile_linear_pre<1, 1>(
                segs3, nullptr, a.Be,
                [&](int, int nb) {           // saved gates, previous belief, head gradients: before the contraction
                    PreGate p;
                    const int col = nb * 16 + (lane & 15);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int grow = row0 + 4 * (lane >> 4) + r;
                        const bool ok = grow < a.B && col < a.Be;
                        const float* g = a.sv_gates + (tb + grow) * 4 * a.Be + col;
                        p.r[r] = ok ? g[0] : 0.f;
                        p.z[r] = ok ? g[a.Be] : 0.f;
                        p.n[r] = ok ? g[2 * a.Be] : 0.f;
                        p.hn[r] = ok ? g[3 * a.Be] : 0.f;
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
                        dhc[off] = carry;      // direct path dh*z; phase 4 adds W_hh^T terms for this member's blocks
                    }
                });
        }
        lds_barrier();
        BD_KARGS_FRESH(ap);
        BD_CSTAMP(19);
        // ---- 4: through W_ih / W_hh: this member's column blocks, K split over the waves ----
        const int my_nb = c + wave * C;
        const bool reducer = wave < kLocalBlocks && my_nb < Nb;
        float svx[4] = {1.f, 1.f, 1.f, 1.f};          // reducer's saved embed outputs: in flight before the contraction
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
                const int nb = c + bi * C;
                if (nb < Nb) {
                    floatx4 DX = floatx4{0.f, 0.f, 0.f, 0.f}, DH = DX;
                    const size_t off = (size_t)nb * d.Kb_h * 64 + lane;
                    const floatx4* __restrict__ Wir = reinterpret_cast<const floatx4*>(a.wt_ir) + off;
                    const floatx4* __restrict__ Wiz = reinterpret_cast<const floatx4*>(a.wt_iz) + off;
                    const floatx4* __restrict__ Win = reinterpret_cast<const floatx4*>(a.wt_in) + off;
                    const floatx4* __restrict__ Whr = reinterpret_cast<const floatx4*>(a.wt_hr) + off;
                    const floatx4* __restrict__ Whz = reinterpret_cast<const floatx4*>(a.wt_hz) + off;
                    const floatx4* __restrict__ Whn = reinterpret_cast<const floatx4*>(a.wt_hn) + off;
                    for (int kb = wave; kb < d.Kb_h; kb += kWaves) {
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
                    floatx4* g = G4 + ((wave * kLocalBlocks + bi) * 2) * 64 + lane;
                    g[0] = DX; g[64] = DH;
                }
            }
        }
        lds_barrier();
        float de_keep[4] = {0.f, 0.f, 0.f, 0.f};
        if (reducer) {
            floatx4 DX = floatx4{0.f, 0.f, 0.f, 0.f}, DH = DX;
            for (int w = 0; w < kWaves; ++w) {
                const floatx4* g = G4 + ((w * kLocalBlocks + wave) * 2) * 64 + lane;
                DX += g[0]; DH += g[64];
            }
            const int col = my_nb * 16 + (lane & 15);
            float* xb = xbuf + (size_t)(epoch & 1) * (2 * nh);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int grow = row0 + 4 * (lane >> 4) + r;
                const int off = acc_frag_off(my_nb, lane, r);
                float de = 0.f, carry = 0.f;
                if (grow < a.B && col < a.Be) {
                    de = DX[r] * elu_grad_from_out(svx[r]);
                    carry = dhc[off] + DH[r];
                }
                st_sc1(xb + off, carry);            // payload = [dhc | dE], fragment order, as the LDS layout
                st_sc1(xb + nh + off, de);
                de_keep[r] = de;
            }
        }
        BD_CSTAMP(20);
        publish(flags + c, epoch);
        if (reducer) {
            const int col = my_nb * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int grow = row0 + 4 * (lane >> 4) + r;
                if (grow < a.B && col < a.Be) a.d_embed_pre[(tb + grow) * a.Be + col] = de_keep[r];
            }
        }
        BD_CSTAMP(21);
        wait_all(flags, C, epoch, err, spin_limit, kErrBwd);
        BD_CSTAMP(22);
        gather_payload(xbuf + (size_t)(epoch & 1) * (2 * nh), dhc, 2 * nh);
        lds_barrier();
        BD_KARGS_FRESH(ap);
        BD_CSTAMP(23);
        // ---- 5: d posterior_state_t through the embed layer and the nonterminal mask (every member) ----
        tile_linear<1>(dE, d.Kb_h, a.wt_embed_s, nullptr, a.S, [&](int, int nb, floatx4 acc) {
            const int col = nb * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * (lane >> 4) + r, grow = row0 + row;
                if (col < a.S) {
                    float v = 0.f;
                    if (grow < a.B) v = acc[r] * (a.nonterm ? a.nonterm[tb + grow] : 1.f);
                    ds_plain[row * a.S + col] = v;
                }
            }
        }, scratch);
        lds_barrier();
        BD_CSTAMP(24);
    }
#undef a
}

// Cluster size: one GRU column block per member while all tiles*Nb workgroups stay co-resident (measured best:
// 13 CUs per tile at Be=200, B=50), else two blocks per member; 0 = not applicable (caller uses observe.hip).
static int pick_cluster(int B, int Be) {
    const int Nb = cdiv(Be, 16), tiles = cdiv(B, 16);
    static const char* force2 = getenv("BD_OBS_CLUSTER_BLOCKS");      // tuning: "2" forces two blocks per member
    int C = (tiles * Nb <= 256 && !(force2 && force2[0] == '2')) ? Nb : cdiv(Nb, kLocalBlocks);
    if (C < 1) C = 1;
    return (C <= kMaxCluster && tiles * C <= 256) ? C : 0;
}

unsigned& cluster_spin_limit() {
    static unsigned v = kSpinLimit;
    return v;
}

static size_t scratch_floats_fwd() {
    const size_t gru = (size_t)kWaves * kLocalBlocks * 4 * 64 * 4;
    return gru > (size_t)kSplitScratchFloats ? gru : (size_t)kSplitScratchFloats;
}
static size_t scratch_floats_bwd() {
    const size_t gru = (size_t)kWaves * kLocalBlocks * 2 * 64 * 4;
    return gru > (size_t)kSplitPartialFloats ? gru : (size_t)kSplitPartialFloats;
}

}  // namespace bd

extern "C" {
using namespace bd;

#ifdef BD_STAMPS
int bd_debug_cstamps(unsigned long long* out64) {
    return hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_cstamps), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : -1;
}
#endif

int bd_observe_cluster_size(int B, int Be) { return pick_cluster(B, Be); }

size_t bd_observe_cluster_ws_floats(int B, int Be) {
    const int tiles = cdiv(B, 16);
    size_t per_tile = (size_t)2 * 2 * cdiv(Be, 16) * kFragFloats;
    const int C = pick_cluster(B, Be);
    if (C > 0 && ksplit_ws_floats_per_tile(C) > per_tile) per_tile = ksplit_ws_floats_per_tile(C);   // K-split form (observe_ksplit.hip)
    return cluster_ws_header_floats(tiles) + (size_t)tiles * per_tile;
}

size_t bd_observe_cluster_err_offset(int B) { return cluster_ws_flag_floats(cdiv(B, 16)); }

// which cluster form bd_observe_forward_cluster / _backward_cluster run wherever the K-split form applies: 3 = K-split with the
// forward GRU split by output columns (observe_ksplit.hip, two hand-offs per forward step), 1 = K-split in both directions
// (three hand-offs), 2 = as 1 with granule hand-offs (R2), 0 = the round-1 form (GRU columns split, the rest redundant),
// -1 = default (environment BD_OBS_KSPLIT, else 3)
int bd_observe_cluster_set_ksplit(int mode) {
    ksplit_mode() = mode < 0 ? -1 : (mode > 3 ? 3 : mode);
    return 0;
}

int bd_observe_cluster_set_spin_limit(unsigned limit) {
    cluster_spin_limit() = limit ? limit : kSpinLimit;
    return 0;
}

// 0 = no member has timed out in ANY cluster launch that used `ws` since the last call (the error word is sticky:
// launches never clear it).  Reads and clears the word: synchronises `stream`.
int bd_observe_cluster_status(float* ws, int B, void* stream) {
    unsigned* w = reinterpret_cast<unsigned*>(ws) + bd_observe_cluster_err_offset(B);
    unsigned e = 0;
    if (hipMemcpyAsync(&e, w, sizeof(e), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess ||
        hipStreamSynchronize((hipStream_t)stream) != hipSuccess)
        return fail("bd_observe_cluster_status: copy failed");
    if (e == 0) return 0;
    if (hipMemsetAsync(w, 0, sizeof(e), (hipStream_t)stream) != hipSuccess) return fail("bd_observe_cluster_status: clear failed");
    return fail("bd_observe_cluster_status: a cluster member timed out waiting for its peers (%s%s scan): results of that "
                "launch are wrong", (e & kErrFwd) ? "forward" : "", (e & kErrBwd) ? ((e & kErrFwd) ? "+backward" : "backward") : "");
}

int bd_observe_forward_cluster(const bd_observe_fwd_args* a, float* ws, size_t ws_floats, void* stream) {
    BD_REQUIRE(a && ws && a->T > 0 && a->B > 0 && a->Be > 0 && a->S > 0 && a->A > 0 && a->Hd > 0,
               "bd_observe_forward_cluster: bad arguments");
    BD_REQUIRE(a->w_embed_s && a->w_embed_a && a->b_embed && a->w_ir && a->w_iz && a->w_in && a->w_hr && a->w_hz &&
                   a->w_hn && a->b_ih && a->b_hh && a->w_q1h && a->b_q1 && a->w_q2m && a->w_q2s && a->b_q2 &&
                   a->init_belief && a->init_state && a->actions && a->pre_emb && a->eps_post && a->feat &&
                   a->post_mean && a->post_std, "bd_observe_forward_cluster: missing pointers");
    BD_REQUIRE(a->S <= kHeadMaxN, "bd_observe_forward_cluster: state width above %d", kHeadMaxN);
    const int C = pick_cluster(a->B, a->Be);
    BD_REQUIRE(C > 0, "bd_observe_forward_cluster: B=%d, Be=%d do not fit the cluster variant", a->B, a->Be);
    const int tiles = cdiv(a->B, 16);
    BD_REQUIRE(ws_floats >= bd_observe_cluster_ws_floats(a->B, a->Be), "bd_observe_forward_cluster: workspace too small");
    const ObsDimsC d(a->Be, a->S, a->A, a->Hd);
    const size_t lds = ((size_t)(3 * d.Kb_h + d.Kb_hd + d.Kb_s + d.Kb_a) * kFragFloats + 16 * a->S + scratch_floats_fwd()) *
                       sizeof(float);
    BD_REQUIRE(lds <= (size_t)kMaxLds, "bd_observe_forward_cluster: needs %zu B of LDS", lds);
    if (ksplit_ok(a->Be, a->S, a->A, a->Hd, C))          // every layer split along K over the members, weights in registers
        return launch_observe_kfwd(a, ws, C, tiles, (hipStream_t)stream);      // (zeroes what its hand-off form polls)
    if (allow_big_lds(observe_cfwd_kernel)) return -1;
    const size_t dyn = launch_lds(observe_cfwd_kernel, lds, "bd_observe_forward_cluster");
    if (!dyn) return -1;
    if (hipMemsetAsync(ws, 0, cluster_ws_flag_floats(tiles) * sizeof(float), (hipStream_t)stream) != hipSuccess)
        return fail("bd_observe_forward_cluster: memset failed");
    hipLaunchKernelGGL(observe_cfwd_kernel, dim3(tiles * C), dim3(kThreads), dyn, (hipStream_t)stream, *a, ws, C, tiles,
                       cluster_spin_limit());
    BD_CHECK_LAUNCH("bd_observe_forward_cluster");
    return 0;
}

int bd_observe_backward_cluster(const bd_observe_bwd_args* a, float* ws, size_t ws_floats, void* stream) {
    BD_REQUIRE(a && ws && a->T > 0 && a->B > 0 && a->Be > 0 && a->S > 0 && a->A > 0 && a->Hd > 0,
               "bd_observe_backward_cluster: bad arguments");
    BD_REQUIRE(a->wt_embed_s && a->wt_ir && a->wt_iz && a->wt_in && a->wt_hr && a->wt_hz && a->wt_hn && a->wt_q1h &&
                   a->wt_q2m && a->wt_q2s && a->init_belief && a->eps_post && a->feat && a->post_std && a->sv_x &&
                   a->sv_gates && a->sv_q && a->dfeat && a->d_embed_pre && a->d_gi && a->d_gh && a->d_q1_pre && a->d_q2_out,
               "bd_observe_backward_cluster: missing pointers");
    const int C = pick_cluster(a->B, a->Be);
    BD_REQUIRE(C > 0, "bd_observe_backward_cluster: B=%d, Be=%d do not fit the cluster variant", a->B, a->Be);
    const int tiles = cdiv(a->B, 16);
    BD_REQUIRE(ws_floats >= bd_observe_cluster_ws_floats(a->B, a->Be), "bd_observe_backward_cluster: workspace too small");
    const ObsDimsC d(a->Be, a->S, a->A, a->Hd);
    const size_t lds = ((size_t)(6 * d.Kb_h + d.Kb_hd + 2 * d.Kb_s) * kFragFloats + 16 * a->S + scratch_floats_bwd()) *
                       sizeof(float);
    BD_REQUIRE(lds <= (size_t)kMaxLds, "bd_observe_backward_cluster: needs %zu B of LDS", lds);
    if (ksplit_ok(a->Be, a->S, a->A, a->Hd, C)) return launch_observe_kbwd(a, ws, C, tiles, (hipStream_t)stream);
    if (allow_big_lds(observe_cbwd_kernel)) return -1;
    const size_t dyn = launch_lds(observe_cbwd_kernel, lds, "bd_observe_backward_cluster");
    if (!dyn) return -1;
    if (hipMemsetAsync(ws, 0, cluster_ws_flag_floats(tiles) * sizeof(float), (hipStream_t)stream) != hipSuccess)
        return fail("bd_observe_backward_cluster: memset failed");
    hipLaunchKernelGGL(observe_cbwd_kernel, dim3(tiles * C), dim3(kThreads), dyn, (hipStream_t)stream, *a, ws, C, tiles,
                       cluster_spin_limit());
    BD_CHECK_LAUNCH("bd_observe_backward_cluster");
    return 0;
}

}  // extern "C"
