// scan_cat.hip -- the RSSM observe scan and the imagination rollout for latent_distribution="Categorical"
// (BASELINE configs[4]: algorithm=dreamerV2, 32 x 32 latents): TransitionModel.forward (src/models.py:191-299,
// Categorical branches :226-228,258-260,269-271) and Dreamer.imagine_ahead (src/dreamer.py:179-237) with
// CategoricalBeliefModel heads (src/models.py:76-117), forward and backward.
//
// Same design as observe.hip / imagine.hip: one persistent launch per direction, a workgroup owns 16 rows for all time
// steps, carries live in LDS in MFMA fragment order, weights stream from L2.  What the one-hot state changes:
//   * it is carried as D class indices (+ one weight per factor: 1 after a sample, the stored value for a caller's
//     initial state, 0 for the all-zero initial state); W_es s~ of the embed layer and W_a0s s of the actor's first layer
//     are GATHERS of D rows of the transposed weights instead of K = D*C contractions;
//   * heads produce D*C logits in 256-column chunks through an LDS image (bd_categorical.h);
//   * backward: per chunk, g = W_es^T d(embed pre-activation of step t+1) [* nonterminal] + heads' gradient, then the
//     straight-through softmax Jacobian per factor, then the chunk's K blocks of d hidden = d logits W2 accumulate in
//     registers -- the carry between time steps is the embed gradient (Be wide), never an S-wide vector.
#include "bd_device.h"
#include "bd_host.h"
#include "bd_scan.h"
#include "bd_categorical.h"

namespace bd {

// Diagnostic (-DBD_STAMPS builds only): s_memtime of thread 0 of workgroup 0 at the phase boundaries of step t == 5.
#ifdef BD_STAMPS
__device__ unsigned long long g_catstamps[64];
#define CAT_STAMP(slot)                                                                                    \
    do {                                                                                                   \
        if (blockIdx.x == 0 && threadIdx.x == 0 && t == 5) g_catstamps[slot] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define CAT_STAMP(slot)
#endif

// =====================================================================================================================
// observe, forward
// =====================================================================================================================
__global__ __launch_bounds__(kThreads) void observe_cat_fwd_kernel(bd_observe_cat_fwd_args a_) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    BD_KARGS(bd_observe_cat_fwd_args, ap);
#define a (*ap)
    const CatGeo g(a.D, a.C);
    const int Kb_h = cdiv(a.Be, 16), Kb_a = cdiv(a.A, 16), Kb_hd = cdiv(a.Hd, 16);
    const int lane = threadIdx.x & 63;
    const int row0 = blockIdx.x * 16;
    const int S = g.S, F = a.Be + S;
    const int nh = Kb_h * kFragFloats, nhd = Kb_hd * kFragFloats;
    const int rows_valid = a.B - row0 < 16 ? a.B - row0 : 16;
    float* h_cur = smem;
    float* h_nxt = h_cur + nh;
    float* xf = h_nxt + nh;
    float* qf = xf + nh;
    float* af = qf + nhd;
    const CatFull gf(a.D, a.C);
    float* xs = af + Kb_a * kFragFloats;          // [16][Be] gathered W_es s~
    float* lg = xs + 16 * a.Be;                   // all S logits of the tile (swizzled image)
    float* sw_l = lg + gf.image_floats();         // [16][D]
    float* mrow = sw_l + 16 * g.D;                // [16] nonterminal mask of the step
    int* sidx_l = reinterpret_cast<int*>(mrow + 16);   // [16][D]

    load_tile_concat<1>(h_cur, Kb_h, row0, a.B, a.init_belief, a.Be, a.Be, nullptr, 0, 0);
    state_to_indices(g, a.init_state, (size_t)S, row0, a.B, sidx_l, sw_l);
    lds_barrier();

    const GruW gw{a.w_ir, a.w_iz, a.w_in, a.w_hr, a.w_hz, a.w_hn, a.b_ih, a.b_hh};

    for (int t = 0; t < a.T; ++t) {
        const size_t tb = (size_t)t * a.B;
        CAT_STAMP(0);
        BD_KARGS_FRESH(ap);
        // ---- A: mask, action fragments ----
        if (threadIdx.x < 16)
            mrow[threadIdx.x] = (a.nonterm && row0 + (int)threadIdx.x < a.B) ? a.nonterm[tb + row0 + threadIdx.x] : 1.f;
        for (int i = threadIdx.x; i < 16 * Kb_a * 16; i += blockDim.x) {
            const int r = i / (Kb_a * 16), k = i - r * (Kb_a * 16);
            const int grow = row0 + r;
            af[frag_idx(r, k)] = (grow < a.B && k < a.A) ? a.actions[(tb + grow) * a.A + k] : 0.f;
        }
        lds_barrier();
        CAT_STAMP(1);
        BD_KARGS_FRESH(ap);
        // ---- A2: W_es s~ as a gather; the masked state for the embed weight gradient ----
        state_gather(g, a.w_embed_sT, a.Be, sidx_l, sw_l, mrow, xs);
        CAT_STAMP(2);
        if (a.sv_s) write_onehot(g, sidx_l, sw_l, mrow, a.sv_s + (tb + row0) * S, (size_t)S, rows_valid);
        lds_barrier();
        CAT_STAMP(3);
        BD_KARGS_FRESH(ap);
        // ---- B: x = ELU(W_es s~ + W_ea a + b_e) ----
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
                    if (a.sv_x && ok) a.sv_x[(tb + grow) * a.Be + col] = v;
                }
            });
        }
        lds_barrier();
        CAT_STAMP(4);
        BD_KARGS_FRESH(ap);
        // ---- C: GRU ----
        gru_tile(xf, h_cur, Kb_h, a.Be, gw, [&](int nb, floatx4 R, floatx4 Z, floatx4 NI, floatx4 NH) {
            const int col = nb * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int grow = row0 + 4 * (lane >> 4) + r;
                const int off = acc_frag_off(nb, lane, r);
                const float rr = sigmoidf(R[r]), zz = sigmoidf(Z[r]);
                const float nn = tanh_act(NI[r] + rr * NH[r]);
                const float hn = (1.f - zz) * nn + zz * h_cur[off];
                const bool ok = grow < a.B && col < a.Be;
                h_nxt[off] = ok ? hn : 0.f;
                if (ok) {
                    a.feat[(tb + grow) * F + col] = hn;
                    if (a.sv_gates) {
                        float* gg = a.sv_gates + (tb + grow) * 4 * a.Be + col;
                        gg[0] = rr; gg[a.Be] = zz; gg[2 * a.Be] = nn; gg[3 * a.Be] = NH[r];
                    }
                }
            }
        }, lg);
        lds_barrier();
        CAT_STAMP(5);
        BD_KARGS_FRESH(ap);
        // ---- D: posterior hidden ----
        tile_linear<1>(h_nxt, Kb_h, a.w_q1h, a.b_q1, a.Hd, [&](int, int nb, floatx4 acc) {
            const int col = nb * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int grow = row0 + 4 * (lane >> 4) + r;
                const bool ok = grow < a.B && col < a.Hd;
                const float v = ok ? elu(acc[r] + a.pre_emb[(tb + grow) * a.Hd + col]) : 0.f;
                qf[acc_frag_off(nb, lane, r)] = v;
                if (ok && a.sv_q) a.sv_q[(tb + grow) * a.Hd + col] = v;
            }
        });
        lds_barrier();
        CAT_STAMP(6);
        BD_KARGS_FRESH(ap);
        // ---- E: posterior logits, sample ----
        cat_head_forward_full(gf, qf, Kb_hd, a.w_q2, a.b_q2, a.q_post + (tb + row0) * S, a.post_logits + (tb + row0) * S,
                              rows_valid, lg, sidx_l);
        CAT_STAMP(7);
        for (int i = threadIdx.x; i < 16 * g.D; i += blockDim.x) {
            const int row = i / g.D;
            sw_l[i] = row < rows_valid ? 1.f : 0.f;
            if (row < rows_valid) a.sidx[(tb + row0) * g.D + i] = (unsigned char)sidx_l[i];
        }
        lds_barrier();
        CAT_STAMP(8);
        write_onehot(g, sidx_l, sw_l, nullptr, a.feat + (tb + row0) * F + a.Be, (size_t)F, rows_valid);
        CAT_STAMP(9);
        float* tmp = h_cur; h_cur = h_nxt; h_nxt = tmp;
    }
#undef a
}

// The head's backward shared by both scans.  On entry dE holds d(embed pre-activation) of the step AFTER this one
// (fragment tile, Kb_h blocks; ignored when !have_carry).  Accumulates d hidden = d logits * W2 into acc[] (this wave's
// column blocks nb = wave, wave + kWaves of an Hd-wide output) and writes the logit gradients to dlogit_row0 (global).
//   g = [have_carry] (dE W_es^T-packed chunk) * scale_l[row]  +  dstate_row0 (global, row stride ld_ds)
//   d logits = jacobian(g | logits)  +  dextra_row0 (global, row stride S; may be null)
// Per 256-column chunk: (a) the carry's contraction, its epilogue adding the heads' gradient, beside the staging of the
// chunk's logits -> barrier -> (b) one thread per (row, factor): softmax Jacobian, + direct logit gradient, out to HBM and
// into the fragment tile of the chunk's K blocks -> barrier -> (c) MFMA accumulate (overlaps the next chunk's (a)).
template <int NACC>
__device__ __forceinline__ void cat_head_backward(const CatGeo& g, bool have_carry, const float* __restrict__ dE, int Kb_h,
                                                  const float* __restrict__ wt_embed_s, const float* __restrict__ scale_l,
                                                  const float* __restrict__ dstate_row0, size_t ld_ds,
                                                  const float* __restrict__ logits_row0, const float* __restrict__ dextra_row0,
                                                  float* __restrict__ dlogit_row0, const float* __restrict__ wt2, int Hd,
                                                  int rows_valid, float* __restrict__ pl, float* __restrict__ lgs,
                                                  float* __restrict__ dLf, floatx4 (&acc)[NACC]) {
    const int tid = bd_tid();
    const int lane = tid & 63, wave = bd_wave(tid);
    const int Nb_hd = cdiv(Hd, 16), Kb_S = cdiv(g.S, 16);
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = floatx4{0.f, 0.f, 0.f, 0.f};
    for (int ch = 0; ch < g.NCH; ++ch) {
        const int n = g.cols(ch);
        // (a) g = carry through the embed layer's state columns + heads' gradient; logits staged beside it
        cat_stage(g, ch, logits_row0, (size_t)g.S, rows_valid, 0.f, lgs);
        if (have_carry) {
            const Seg seg[1] = {{dE, wt_embed_s + (size_t)ch * (g.CW / 16) * Kb_h * kFragFloats, Kb_h}};
            tile_linear_pre<1, 1>(
                seg, nullptr, n,
                [&](int, int nb) {          // the heads' gradient for this accumulator: requested before the contraction
                    Pre4 p;
                    const int colc = nb * 16 + (lane & 15);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 4 * (lane >> 4) + r;
                        p.v[r] = (colc < n && row < rows_valid) ? dstate_row0[(size_t)row * ld_ds + ch * g.CW + colc] : 0.f;
                    }
                    return p;
                },
                [&](int, int nb, floatx4 a4, const Pre4& p) {
                    const int colc = nb * 16 + (lane & 15);
                    if (colc >= n) return;
                    const int fl = colc / g.C, c = colc - fl * g.C;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 4 * (lane >> 4) + r;
                        pl[g.addr(row, fl, c)] = row < rows_valid ? a4[r] * (scale_l ? scale_l[row] : 1.f) + p.v[r] : 0.f;
                    }
                });
        } else {
            cat_stage(g, ch, dstate_row0, ld_ds, rows_valid, 0.f, pl);
        }
        lds_barrier();
        // (b) straight-through Jacobian per (row, factor), + direct logit gradient (KL); out to HBM and into the fragment
        //     tile of this chunk's K blocks (columns beyond the chunk's factors stay zero from the kernel's start)
        if (g.C == 32) {          // four lanes per (row, factor): every thread works, one exponential per class
            const int nf = n / 32;
            for (int i = tid; i < 16 * nf * 4; i += blockDim.x) {
                const int grp = i >> 2, quad = i & 3;
                const int row = grp / nf, fl = grp - row * nf;
                const size_t gi = (size_t)row * g.S + ch * g.CW + fl * 32;
                cat_jacobian_quad32(g, lgs, pl, row, fl, quad, row < rows_valid, dextra_row0 ? dextra_row0 + gi : nullptr,
                                    dlogit_row0 ? dlogit_row0 + gi : nullptr, dLf);
            }
            const int npad = cdiv(n, 16) * 16 - n;
            for (int i = tid; i < 16 * npad; i += blockDim.x) dLf[frag_idx(i / npad, n + i % npad)] = 0.f;
        } else {
            const int nf = n / g.C;
            for (int i = tid; i < 16 * nf; i += blockDim.x) {
                const int row = i / nf, fl = i - row * nf;
                const size_t gi = (size_t)row * g.S + ch * g.CW + fl * g.C;
                cat_jacobian(g, lgs, pl, row, fl);
                for (int c = 0; c < g.C; ++c) {
                    float v = 0.f;
                    if (row < rows_valid) {
                        v = pl[g.addr(row, fl, c)] + (dextra_row0 ? dextra_row0[gi + c] : 0.f);
                        if (dlogit_row0) dlogit_row0[gi + c] = v;
                    }
                    dLf[frag_idx(row, fl * g.C + c)] = v;
                }
            }
            const int npad = cdiv(n, 16) * 16 - n;       // columns of the last K block beyond the chunk's factors
            for (int i = tid; i < 16 * npad; i += blockDim.x) dLf[frag_idx(i / npad, n + i % npad)] = 0.f;
        }
        lds_barrier();
        // (c) d hidden += d logits(chunk) * W2[chunk rows, :]   (wt2: packed transpose, out = Hd, in = S)
        {
            const int kbn = cdiv(n, 16);
            const floatx4* __restrict__ A4 = reinterpret_cast<const floatx4*>(dLf) + lane;
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                const int nb = wave + i * kWaves;
                if (nb < Nb_hd) {
                    const floatx4* __restrict__ W4 =
                        reinterpret_cast<const floatx4*>(wt2) + ((size_t)nb * Kb_S + (size_t)ch * (g.CW / 16)) * 64 + lane;
                    floatx4 a0 = acc[i], a1 = floatx4{0.f, 0.f, 0.f, 0.f};
                    pipelined_k<2>(
                        kbn, [&](int kb) { return LinFrag<1, 1>{{A4[kb * 64]}, {W4[(size_t)kb * 64]}}; },
                        [&](const LinFrag<1, 1>& f) {
                            a0 = mfma16(f.a[0][0], f.b[0][0], a0);
                            a1 = mfma16(f.a[0][1], f.b[0][1], a1);
                            a0 = mfma16(f.a[0][2], f.b[0][2], a0);
                            a1 = mfma16(f.a[0][3], f.b[0][3], a1);
                        });
                    acc[i] = a0 + a1;
                }
            }
        }
        // the next chunk's (a) writes pl / lgs (last read in (b), behind the barrier above); its (b) writes dLf, which
        // (c) reads: that hazard is covered by the barrier after the next (a)
    }
    lds_barrier();
}

// =====================================================================================================================
// observe, backward
// =====================================================================================================================
__global__ __launch_bounds__(kThreads) void observe_cat_bwd_kernel(bd_observe_cat_bwd_args a_) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    BD_KARGS(bd_observe_cat_bwd_args, ap);
#define a (*ap)
    const CatGeo g(a.D, a.C);
    const int Kb_h = cdiv(a.Be, 16), Kb_hd = cdiv(a.Hd, 16);
    const int row0 = blockIdx.x * 16;
    const int S = g.S, F = a.Be + S;
    const int nh = Kb_h * kFragFloats, nhd = Kb_hd * kFragFloats;
    const int rows_valid = a.B - row0 < 16 ? a.B - row0 : 16;
    float* dhc = smem;
    float* dR = dhc + nh;
    float* dZ = dR + nh;
    float* dNI = dZ + nh;
    float* dNH = dNI + nh;
    float* dE = dNH + nh;
    float* dQ = dE + nh;                               // Kb_hd
    float* pl = dQ + nhd;
    float* lgs = pl + g.image_floats();
    float* dLf = lgs + g.image_floats();               // CW/16 fragment blocks
    float* mrow = dLf + (g.CW / 16) * kFragFloats;     // [16]

    for (int i = threadIdx.x; i < nh; i += blockDim.x) { dhc[i] = 0.f; dE[i] = 0.f; }
    lds_barrier();

    const GruWT gw{a.wt_ir, a.wt_iz, a.wt_in, a.wt_hr, a.wt_hz, a.wt_hn};
    constexpr int NACC = (16 + kWaves - 1) / kWaves > 2 ? (16 + kWaves - 1) / kWaves : 2;   // Hd <= 256

    for (int t = a.T - 1; t >= 0; --t) {
        const size_t tb = (size_t)t * a.B;
        const int tid = bd_tid();
        const int lane = tid & 63, wave = bd_wave(tid);
        // mask of step t+1 (its input state is posterior_state_t * nonterminal_{t+1})
        CAT_STAMP(16);
        if (tid < 16)
            mrow[tid] = (a.nonterm && t + 1 < a.T && row0 + tid < a.B) ? a.nonterm[tb + a.B + row0 + tid] : 1.f;
        lds_barrier();
        CAT_STAMP(17);
        BD_KARGS_FRESH(ap);
        // ---- 1: d posterior logits_t, d posterior hidden ----
        floatx4 accQ[NACC];
        cat_head_backward<NACC>(g, t + 1 < a.T, dE, Kb_h, a.wt_embed_s, mrow, a.dfeat + (tb + row0) * F + a.Be, (size_t)F,
                                a.post_logits + (tb + row0) * S, a.dpost_logits ? a.dpost_logits + (tb + row0) * S : nullptr,
                                a.d_q2_out + (tb + row0) * S, a.wt_q2, a.Hd, rows_valid, pl, lgs, dLf, accQ);
        CAT_STAMP(18);
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            const int nb = wave + i * kWaves;
            if (nb < Kb_hd) {
                const int col = nb * 16 + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int grow = row0 + 4 * (lane >> 4) + r;
                    float v = 0.f;
                    if (grow < a.B && col < a.Hd) {
                        v = accQ[i][r] * elu_grad_from_out(a.sv_q[(tb + grow) * a.Hd + col]);
                        a.d_q1_pre[(tb + grow) * a.Hd + col] = v;
                    }
                    dQ[acc_frag_off(nb, lane, r)] = v;
                }
            }
        }
        lds_barrier();
        CAT_STAMP(19);
        BD_KARGS_FRESH(ap);
        // ---- 3: total d belief_{t+1}, GRU gate gradients ----
        tile_linear<1>(dQ, Kb_hd, a.wt_q1h, nullptr, a.Be, [&](int, int nb, floatx4 acc) {
            const int col = nb * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int grow = row0 + 4 * (lane >> 4) + r;
                const int off = acc_frag_off(nb, lane, r);
                float vr = 0.f, vz = 0.f, vni = 0.f, vnh = 0.f, carry = 0.f;
                if (grow < a.B && col < a.Be) {
                    const float dh = acc[r] + dhc[off] + a.dfeat[(tb + grow) * F + col];
                    const float* gg = a.sv_gates + (tb + grow) * 4 * a.Be + col;
                    const float rr = gg[0], zz = gg[a.Be], nn = gg[2 * a.Be], hn = gg[3 * a.Be];
                    const float hprev = t > 0 ? a.feat[(tb - a.B + grow) * F + col]
                                              : a.init_belief[(size_t)grow * a.Be + col];
                    const float dn = dh * (1.f - zz);
                    const float dz = dh * (hprev - nn);
                    vni = dn * (1.f - nn * nn);
                    vnh = vni * rr;
                    vr = vni * hn * rr * (1.f - rr);
                    vz = dz * zz * (1.f - zz);
                    carry = dh * zz;
                    float* gi = a.d_gi + (tb + grow) * 3 * a.Be + col;
                    float* gh = a.d_gh + (tb + grow) * 3 * a.Be + col;
                    gi[0] = vr; gi[a.Be] = vz; gi[2 * a.Be] = vni;
                    gh[0] = vr; gh[a.Be] = vz; gh[2 * a.Be] = vnh;
                }
                dR[off] = vr; dZ[off] = vz; dNI[off] = vni; dNH[off] = vnh;
                dhc[off] = carry;
            }
        });
        lds_barrier();
        CAT_STAMP(20);
        BD_KARGS_FRESH(ap);
        // ---- 4: through W_ih / W_hh: d embed pre-activation (the carry to step t-1's head), d belief_t ----
        gru_tile_bwd(
            dR, dZ, dNI, dNH, Kb_h, a.Be, gw,
            [&](int nb) {
                Pre4 p;
                const int col = nb * 16 + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int grow = row0 + 4 * (lane >> 4) + r;
                    p.v[r] = (grow < a.B && col < a.Be) ? a.sv_x[(tb + grow) * a.Be + col] : 1.f;
                }
                return p;
            },
            [&](int nb, floatx4 DX, floatx4 DH, const Pre4& p) {
                const int col = nb * 16 + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int grow = row0 + 4 * (lane >> 4) + r;
                    const int off = acc_frag_off(nb, lane, r);
                    float de = 0.f;
                    if (grow < a.B && col < a.Be) {
                        de = DX[r] * elu_grad_from_out(p.v[r]);
                        a.d_embed_pre[(tb + grow) * a.Be + col] = de;
                        dhc[off] += DH[r];
                    }
                    dE[off] = de;
                }
            }, pl);
        lds_barrier();
        CAT_STAMP(21);
    }
#undef a
}

// =====================================================================================================================
// imagination, forward
// =====================================================================================================================
__global__ __launch_bounds__(kThreads) void imagine_cat_fwd_kernel(bd_imagine_cat_fwd_args a_) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    BD_KARGS(bd_imagine_cat_fwd_args, ap);
#define a (*ap)
    const CatGeo g(a.D, a.C);
    const int Kb_h = cdiv(a.Be, 16), Kb_a = cdiv(a.A, 16), Kb_hd = cdiv(a.Hd, 16);
    // a launch of fewer workgroups than tiles (host: whole rounds, see cat_grid) walks its tiles one after the other
    for (int tile_ = blockIdx.x; tile_ * 16 < a.N; tile_ += gridDim.x) {
    const int row0 = tile_ * 16;
    const int S = g.S, F = a.Be + S, A = a.A;
    const int nh = Kb_h * kFragFloats, nhd = Kb_hd * kFragFloats;
    const int rows_valid = a.N - row0 < 16 ? a.N - row0 : 16;
    const int wmax = a.Be > a.Hd ? a.Be : a.Hd;
    float* h_cur = smem;
    float* h_nxt = h_cur + nh;
    float* xf = h_nxt + nh;
    float* bufA = xf + nh;
    float* bufB = bufA + nhd;
    float* af = bufB + nhd;
    float* xs = af + Kb_a * kFragFloats;          // [16][max(Be, Hd)] gathered state columns of the layer at hand
    float* mean_s = xs + 16 * wmax;               // [16][A]
    float* std_s = mean_s + 16 * A;
    float* lp_rj = std_s + 16 * A;
    float* sw_l = lp_rj + 16 * A;                 // [16][D]
    int* sidx_l = reinterpret_cast<int*>(sw_l + 16 * g.D);
    // one region, three tenants in disjoint phases: the actor head's split-K scratch, the entropy partial sums, the
    // prior head's logits / draws images (16-byte aligned: the host rounds the offsets)
    float* uni = reinterpret_cast<float*>(sidx_l + 16 * g.D);
    const CatFull gf(a.D, a.C);
    float* scratch = uni;
    float* part = uni;                            // [kWaves][16][A][3]
    float* lg = uni;                              // all S prior logits of the tile (swizzled image)

    load_tile_concat<1>(h_cur, Kb_h, row0, a.N, a.start_feat, F, a.Be, nullptr, 0, 0);
    if (a.start_sidx) {
        for (int i = threadIdx.x; i < 16 * g.D; i += blockDim.x) {
            const int row = i / g.D;
            sidx_l[i] = row < rows_valid ? (int)a.start_sidx[(size_t)row0 * g.D + i] : 0;
            sw_l[i] = row < rows_valid ? 1.f : 0.f;
        }
    } else {
        // no indices given (API callers: Dreamer.get_action / imagine_ahead on a dense state): every factor of the
        // dense start state is all-zero (the collect loop's initial state, src/main.py:91-95: the actor and the embed
        // layer then see zeros, as in the reference) or one-hot -- same rule as the observe scan's init_state
        state_to_indices(g, a.start_feat + a.Be, (size_t)F, row0, a.N, sidx_l, sw_l);
    }
    for (int i = threadIdx.x; i < Kb_a * kFragFloats; i += blockDim.x) af[i] = 0.f;
    lds_barrier();

    const GruW gw{a.w_ir, a.w_iz, a.w_in, a.w_hr, a.w_hz, a.w_hn, a.b_ih, a.b_hh};
    const size_t act_stride = (size_t)a.Hm * a.N * a.Hd;
    const float inv_ns = 1.f / (float)a.n_samples;

    for (int t = 0; t < a.Hm; ++t) {
        const size_t tn = (size_t)t * a.N;
        const int tid = bd_tid();
        const int lane = tid & 63, wave = bd_wave(tid);
        auto hidden_epi = [&](float* dst, float* save, size_t tn_, int width) {
            return HiddenEpi{dst, save, tn_, width, a.N, row0, lane};
        };
        BD_KARGS_FRESH(ap);
        // ---- actor layer 0: W_a0h h + gather(W_a0s, s) ----
        state_gather(g, a.w_a0sT, a.Hd, sidx_l, sw_l, nullptr, xs);
        lds_barrier();
        {
            const Seg segs[1] = {{h_cur, a.w_a0h, Kb_h}};
            tile_linear_seg<1>(segs, a.b_a[0], a.Hd, [&](int nb, floatx4 acc) {
                const int col = nb * 16 + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 4 * (lane >> 4) + r, grow = row0 + row;
                    const bool ok = grow < a.N && col < a.Hd;
                    const float v = ok ? elu(acc[r] + xs[row * a.Hd + col]) : 0.f;
                    bufA[acc_frag_off(nb, lane, r)] = v;
                    if (ok && a.sv_actor) a.sv_actor[(tn + grow) * a.Hd + col] = v;
                }
            });
        }
        lds_barrier();
        {
            float* src = bufA;
            float* dst = bufB;
            for (int l = 1; l < 4; ++l) {
                const Seg segs[1] = {{src, a.w_a[l - 1], Kb_hd}};
                tile_linear_seg<1>(segs, a.b_a[l], a.Hd,
                                      hidden_epi(dst, a.sv_actor ? a.sv_actor + l * act_stride : nullptr, tn, a.Hd));
                lds_barrier();
                float* tmp = src; src = dst; dst = tmp;
            }
        }
        BD_KARGS_FRESH(ap);
        // ---- actor output, action sample (layer-3 activations are in bufB) ----
        {
            const Seg2 segs[1] = {{bufB, a.w_a4m, a.w_a4s, Kb_hd}};
            tile_dual_head_elem<1>(
                segs, a.b_a4, a.b_a4 + A, A, scratch,
                [&](int row, int col) { return row0 + row < a.N ? a.eps_action[(tn + row0 + row) * A + col] : 0.f; },
                [&](int row, int col, float Mn, float Rw, float eps) {
                    const int grow = row0 + row;
                    float act = 0.f;
                    if (grow < a.N) {
                        const float th = tanh_act(Mn / a.act_mean_scale);
                        const float mean = a.act_mean_scale * th;
                        const float pre = Rw + a.act_raw_init_std;
                        const float sd = softplusf(pre) + a.act_min_std;
                        act = tanh_act(mean + sd * eps);
                        a.action[(tn + grow) * A + col] = act;
                        mean_s[row * A + col] = mean;
                        std_s[row * A + col] = sd;
                        if (a.sv_act_stats) {      // slots 2, 3: mean and std for bd_actor_entropy, which replaces them
                            float* st = a.sv_act_stats + (tn + grow) * 4 * A + col;
                            st[0] = th;
                            st[A] = sigmoidf(pre);
                            st[2 * A] = mean;
                            st[3 * A] = sd;
                        }
                    }
                    af[frag_idx(row, col)] = act;
                });
        }
        lds_barrier();
        BD_KARGS_FRESH(ap);
        // ---- entropy: n_samples draws per (row, action dim); thread = (row, sample lane) ----
        // (off the recurrence when the actor statistics are saved: bd_actor_entropy after the scan, see imagine.hip)
        const bool ent_inline = a.sv_act_stats == nullptr;
        if (ent_inline) {
            const int row = tid & 15, sl = tid >> 4;
            const int grow = row0 + row;
            for (int j = 0; j < A; ++j) {
                float lp = 0.f, dm = 0.f, ds = 0.f;
                if (grow < a.N) {
                    const EntConst ec = entropy_const(mean_s[row * A + j], std_s[row * A + j]);
                    for (int k = sl; k < a.n_samples; k += kThreads / 16) {
                        const float e = a.eps_entropy[(((size_t)t * a.n_samples + k) * a.N + grow) * A + j];
                        float l1, d1, d2;
                        entropy_sample(ec, e, l1, d1, d2);
                        lp += l1; dm += d1; ds += d2;
                    }
                }
                lp += __shfl_xor(lp, 16, 64); lp += __shfl_xor(lp, 32, 64);
                dm += __shfl_xor(dm, 16, 64); dm += __shfl_xor(dm, 32, 64);
                ds += __shfl_xor(ds, 16, 64); ds += __shfl_xor(ds, 32, 64);
                if (lane < 16) {
                    float* p = part + ((wave * 16 + row) * A + j) * 3;
                    p[0] = lp; p[1] = dm; p[2] = ds;
                }
            }
            lds_barrier();
            for (int i = tid; i < 16 * A; i += blockDim.x) {
                const int row = i / A, j = i - row * A;
                float lp = 0.f;
                for (int w = 0; w < kWaves; ++w) lp += part[((w * 16 + row) * A + j) * 3];
                lp_rj[i] = lp;
            }
        }
        // W_es s as a gather (xs is free since actor layer 0)
        state_gather(g, a.w_embed_sT, a.Be, sidx_l, sw_l, nullptr, xs);
        lds_barrier();
        if (ent_inline && tid < 16 && row0 + tid < a.N) {
            float s = 0.f;
            for (int j = 0; j < A; ++j) s += lp_rj[tid * A + j];
            a.entropy[tn + row0 + tid] = -s * inv_ns;
        }
        BD_KARGS_FRESH(ap);
        // ---- embed ----
        {
            const Seg segs[1] = {{af, a.w_embed_a, Kb_a}};
            tile_linear_seg<1>(segs, a.b_embed, a.Be, [&](int nb, floatx4 acc) {
                const int col = nb * 16 + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 4 * (lane >> 4) + r, grow = row0 + row;
                    const bool ok = grow < a.N && col < a.Be;
                    const float v = ok ? elu(acc[r] + xs[row * a.Be + col]) : 0.f;
                    xf[acc_frag_off(nb, lane, r)] = v;
                    if (ok && a.sv_x) a.sv_x[(tn + grow) * a.Be + col] = v;
                }
            });
        }
        lds_barrier();
        BD_KARGS_FRESH(ap);
        // ---- GRU ----
        gru_tile(xf, h_cur, Kb_h, a.Be, gw, [&](int nb, floatx4 R, floatx4 Z, floatx4 NI, floatx4 NH) {
            const int col = nb * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int grow = row0 + 4 * (lane >> 4) + r;
                const int off = acc_frag_off(nb, lane, r);
                const float rr = sigmoidf(R[r]), zz = sigmoidf(Z[r]);
                const float nn = tanh_act(NI[r] + rr * NH[r]);
                const float hn = (1.f - zz) * nn + zz * h_cur[off];
                const bool ok = grow < a.N && col < a.Be;
                h_nxt[off] = ok ? hn : 0.f;
                if (ok) {
                    a.feat[(tn + grow) * F + col] = hn;
                    if (a.sv_gates) {
                        float* gg = a.sv_gates + (tn + grow) * 4 * a.Be + col;
                        gg[0] = rr; gg[a.Be] = zz; gg[2 * a.Be] = nn; gg[3 * a.Be] = NH[r];
                    }
                }
            }
        }, uni);
        lds_barrier();
        BD_KARGS_FRESH(ap);
        // ---- prior hidden, logits, sample ----
        {
            const Seg segs[1] = {{h_nxt, a.w_p1, Kb_h}};
            tile_linear_seg<1>(segs, a.b_p1, a.Hd, hidden_epi(bufA, a.sv_p, tn, a.Hd));
        }
        lds_barrier();
        cat_head_forward_full(gf, bufA, Kb_hd, a.w_p2, a.b_p2, a.q_prior + (tn + row0) * S, a.prior_logits + (tn + row0) * S,
                              rows_valid, lg, sidx_l);
        for (int i = tid; i < 16 * g.D; i += blockDim.x) {
            const bool ok = i / g.D < rows_valid;
            if (ok) a.sidx[(tn + row0) * g.D + i] = (unsigned char)sidx_l[i];
            sw_l[i] = ok ? 1.f : 0.f;      // a sampled state is one-hot whatever the start state's weights were (same i as below)
        }
        write_onehot(g, sidx_l, sw_l, nullptr, a.feat + (tn + row0) * F + a.Be, (size_t)F, rows_valid);
        lds_barrier();
        float* tmp = h_cur; h_cur = h_nxt; h_nxt = tmp;
    }
    lds_barrier();        // the next tile re-uses every LDS region
    }
#undef a
}

// =====================================================================================================================
// imagination, backward
// =====================================================================================================================
__global__ __launch_bounds__(kThreads) void imagine_cat_bwd_kernel(bd_imagine_cat_bwd_args a_) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    BD_KARGS(bd_imagine_cat_bwd_args, ap);
#define a (*ap)
    const CatGeo g(a.D, a.C);
    const int Kb_h = cdiv(a.Be, 16), Kb_a = cdiv(a.A, 16), Kb_hd = cdiv(a.Hd, 16);
    // a launch of fewer workgroups than tiles (host: whole rounds, see cat_grid) walks its tiles one after the other
    for (int tile_ = blockIdx.x; tile_ * 16 < a.N; tile_ += gridDim.x) {
    const int row0 = tile_ * 16;
    const int S = g.S, F = a.Be + S, A = a.A;
    const int nh = Kb_h * kFragFloats, nhd = Kb_hd * kFragFloats, na = Kb_a * kFragFloats;
    const int rows_valid = a.N - row0 < 16 ? a.N - row0 : 16;
    float* dhc = smem;
    float* dR = dhc + nh;
    float* dZ = dR + nh;
    float* dNI = dZ + nh;
    float* dNH = dNI + nh;
    float* dE = dNH + nh;
    float* dP = dE + nh;          // Kb_hd
    float* dAm = dP + nhd;        // Kb_a
    float* dAr = dAm + na;
    // one region, tenants in disjoint phases: head backward images | split-K scratch | actor backward buffers
    float* uni = dAr + na;
    float* pl = uni;
    float* lgs = pl + g.image_floats();
    float* dLf = lgs + g.image_floats();
    float* scratch = uni;
    float* bufA = uni;
    float* bufB = bufA + nhd;

    for (int i = threadIdx.x; i < nh; i += blockDim.x) { dhc[i] = 0.f; dE[i] = 0.f; }
    lds_barrier();

    const GruWT gw{a.wt_ir, a.wt_iz, a.wt_in, a.wt_hr, a.wt_hz, a.wt_hn};
    const size_t act_stride = (size_t)a.Hm * a.N * a.Hd;
    constexpr int NACC = (16 + kWaves - 1) / kWaves > 2 ? (16 + kWaves - 1) / kWaves : 2;

    for (int t = a.Hm - 1; t >= 0; --t) {
        const size_t tn = (size_t)t * a.N;
        const int tid = bd_tid();
        const int lane = tid & 63, wave = bd_wave(tid);
        auto dpre_epi = [&](float* dst, float* out, size_t tn_, int width) {
            return DpreEpi{dst, out, tn_, width, a.N, row0, lane};
        };
        auto dpre_pre = [&](const float* saved, size_t tn_, int width) {
            return DprePre{saved, tn_, width, a.N, row0, lane};
        };
        BD_KARGS_FRESH(ap);
        // ---- 1: d prior logits_t (straight-through), d prior hidden ----
        floatx4 accP[NACC];
        cat_head_backward<NACC>(g, t + 1 < a.Hm, dE, Kb_h, a.wt_embed_s, nullptr, a.dfeat + (tn + row0) * F + a.Be, (size_t)F,
                                a.prior_logits + (tn + row0) * S, nullptr, nullptr, a.wt_p2, a.Hd, rows_valid, pl, lgs, dLf,
                                accP);
        // d action of the NEXT step's embed is consumed below from dE before dE is overwritten: order matters --
        // the action gradient of step t comes from d(embed pre-activation of step t), produced in phase 4 of THIS
        // iteration; dE of step t+1 (read above) is dead from here on.
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            const int nb = wave + i * kWaves;
            if (nb < Kb_hd) {
                const int col = nb * 16 + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int grow = row0 + 4 * (lane >> 4) + r;
                    float v = 0.f;
                    if (grow < a.N && col < a.Hd) v = accP[i][r] * elu_grad_from_out(a.sv_p[(tn + grow) * a.Hd + col]);
                    dP[acc_frag_off(nb, lane, r)] = v;
                }
            }
        }
        lds_barrier();
        BD_KARGS_FRESH(ap);
        // ---- 3: total d belief_{t+1}; GRU gates ----
        {
            const Seg segs3[1] = {{dP, a.wt_p1, Kb_hd}};
            tile_linear_pre<1, 1>(
                segs3, nullptr, a.Be,
                [&](int, int nb) {
                    PreGate p;
                    const int col = nb * 16 + (lane & 15);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int grow = row0 + 4 * (lane >> 4) + r;
                        const bool ok = grow < a.N && col < a.Be;
                        const float* gg = a.sv_gates + (tn + grow) * 4 * a.Be + col;
                        p.r[r] = ok ? gg[0] : 0.f;
                        p.z[r] = ok ? gg[a.Be] : 0.f;
                        p.n[r] = ok ? gg[2 * a.Be] : 0.f;
                        p.hn[r] = ok ? gg[3 * a.Be] : 0.f;
                        p.hprev[r] = !ok ? 0.f : (t > 0 ? a.feat[(tn - a.N + grow) * F + col]
                                                        : a.start_feat[(size_t)grow * F + col]);
                        p.dfeat[r] = ok ? a.dfeat[(tn + grow) * F + col] : 0.f;
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
                        if (grow < a.N && col < a.Be) {
                            const float dh = acc[r] + dhc[off] + p.dfeat[r];
                            const float rr = p.r[r], zz = p.z[r], nn = p.n[r], hn = p.hn[r];
                            const float dn = dh * (1.f - zz);
                            const float dz = dh * (p.hprev[r] - nn);
                            vni = dn * (1.f - nn * nn);
                            vnh = vni * rr;
                            vr = vni * hn * rr * (1.f - rr);
                            vz = dz * zz * (1.f - zz);
                            carry = dh * zz;
                        }
                        dR[off] = vr; dZ[off] = vz; dNI[off] = vni; dNH[off] = vnh;
                        dhc[off] = carry;
                    }
                });
        }
        lds_barrier();
        BD_KARGS_FRESH(ap);
        // ---- 4: through W_ih / W_hh ----
        gru_tile_bwd(
            dR, dZ, dNI, dNH, Kb_h, a.Be, gw,
            [&](int nb) {
                Pre4 p;
                const int col = nb * 16 + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int grow = row0 + 4 * (lane >> 4) + r;
                    p.v[r] = (grow < a.N && col < a.Be) ? a.sv_x[(tn + grow) * a.Be + col] : 1.f;
                }
                return p;
            },
            [&](int nb, floatx4 DX, floatx4 DH, const Pre4& p) {
                const int col = nb * 16 + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int grow = row0 + 4 * (lane >> 4) + r;
                    const int off = acc_frag_off(nb, lane, r);
                    float de = 0.f;
                    if (grow < a.N && col < a.Be) {
                        de = DX[r] * elu_grad_from_out(p.v[r]);
                        dhc[off] += DH[r];
                    }
                    dE[off] = de;
                }
            }, uni);
        lds_barrier();
        BD_KARGS_FRESH(ap);
        // ---- 5: embed layer -> d action_t -> actor output gradients (d state_t is taken by the next iteration's head) ----
        {
            const Seg segs5[1] = {{dE, a.wt_embed_a, Kb_h}};
            tile_linear_pre<1, 1>(
                segs5, nullptr, A,
                [&](int, int nb) {
                    PreAct p;
                    const int col = nb * 16 + (lane & 15);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int grow = row0 + 4 * (lane >> 4) + r;
                        const bool ok = grow < a.N && col < A;
                        const size_t i = (tn + grow) * A + col;
                        const float* st = a.sv_act_stats + (tn + grow) * 4 * A + col;
                        p.act[r] = ok ? a.action[i] : 0.f;
                        p.eps[r] = ok ? a.eps_action[i] : 0.f;
                        p.th[r] = ok ? st[0] : 0.f;
                        p.sg[r] = ok ? st[A] : 0.f;
                        p.dm[r] = ok ? st[2 * A] : 0.f;
                        p.ds[r] = ok ? st[3 * A] : 0.f;
                    }
                    return p;
                },
                [&](int, int nb, floatx4 acc, const PreAct& p) {
                    const int col = nb * 16 + (lane & 15);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int grow = row0 + 4 * (lane >> 4) + r;
                        float gm = 0.f, gr = 0.f;
                        if (grow < a.N && col < A) {
                            const float dxa = acc[r] * (1.f - p.act[r] * p.act[r]);
                            const float dent = a.ent_weight ? a.dentropy * a.ent_weight[tn + grow] : a.dentropy;
                            const float dmean = dxa + dent * p.dm[r];
                            const float dstd = dxa * p.eps[r] + dent * p.ds[r];
                            gm = dmean * (1.f - p.th[r] * p.th[r]);
                            gr = dstd * p.sg[r];
                            a.d_actor_out[(tn + grow) * 2 * A + col] = gm;
                            a.d_actor_out[(tn + grow) * 2 * A + A + col] = gr;
                        }
                        dAm[acc_frag_off(nb, lane, r)] = gm;
                        dAr[acc_frag_off(nb, lane, r)] = gr;
                    }
                },
                scratch);
        }
        lds_barrier();
        BD_KARGS_FRESH(ap);
        // ---- 6: actor MLP backward (input detached: no gradient below layer 0) ----
        // (off the recurrence: with d_actor_pre == NULL the caller runs it as one dense chain over all rows, imagine.hip)
        if (a.d_actor_pre != nullptr) {
        {
            const Seg segs[2] = {{dAm, a.wt_a4m, Kb_a}, {dAr, a.wt_a4s, Kb_a}};
            tile_linear_pre<1, 2>(segs, nullptr, a.Hd, dpre_pre(a.sv_actor + 3 * act_stride, tn, a.Hd),
                                  dpre_epi(bufA, a.d_actor_pre + 3 * act_stride, tn, a.Hd));
        }
        lds_barrier();
        {
            float* src = bufA;
            float* dst = bufB;
            for (int l = 2; l >= 0; --l) {
                const Seg segs[1] = {{src, a.wt_a[l], Kb_hd}};
                tile_linear_pre<1, 1>(segs, nullptr, a.Hd, dpre_pre(a.sv_actor + l * act_stride, tn, a.Hd),
                                      dpre_epi(l > 0 ? dst : nullptr, a.d_actor_pre + l * act_stride, tn, a.Hd));
                lds_barrier();
                float* tmp = src; src = dst; dst = tmp;
            }
        }
        }
    }
    lds_barrier();        // the next tile re-uses every LDS region
    }
#undef a
}

// Launch geometry of the imagination scans.  A tile (16 rows) takes the same time whatever else runs, and tiles beyond the
// chip's 256 CUs wait for a second round: 307 tiles (configs[4]: 4900 rows) are two rounds, the second one 20 % full, with
// every CU held by a 120 KB workgroup for the whole launch.  The same two rounds on ceil(307 / 2) = 154 workgroups that walk
// two tiles each take the behaviour-learning chain exactly as long and leave 102 CUs to the other streams (conv stacks,
// weight gradients, critic) for the whole launch.  Such a launch asks for the CU's whole LDS so that its workgroups cannot
// share a CU.  BD_CAT_TILE_LOOP=0: one workgroup per tile.  Returns the dynamic LDS size to request (0: error).
template <class K>
static size_t cat_grid(K kernel, int tiles, size_t lds, int* grid) {
    static const char* e = getenv("BD_CAT_TILE_LOOP");
    const int kCUs = 256;
    *grid = tiles;
    if (lds > 64 * 1024 && allow_big_lds(kernel)) return 0;
    if (tiles <= kCUs || (e && e[0] == '0')) return lds;
    const int rounds = cdiv(tiles, kCUs);
    *grid = cdiv(tiles, rounds);
    hipFuncAttributes at;
    if (hipFuncGetAttributes(&at, reinterpret_cast<const void*>(kernel)) != hipSuccess) {
        fail("imagination scan: hipFuncGetAttributes failed");
        return 0;
    }
    const size_t room = (size_t)kMaxLds > at.sharedSizeBytes ? (size_t)kMaxLds - at.sharedSizeBytes : 0;
    return room > lds ? room : lds;
}

}  // namespace bd

extern "C" {
using namespace bd;

#ifdef BD_STAMPS
int bd_debug_catstamps(unsigned long long* out64) {
    return hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_catstamps), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : -1;
}
#endif

#define BD_CAT_GEO(who)                                                                                                     \
    const CatGeo g(a->D, a->C);                                                                                             \
    BD_REQUIRE(a->D > 0 && a->C > 0 && g.ok(), who ": %d x %d latents unsupported (C <= 256; S <= 256 or 256 %% C == 0)",   \
               a->D, a->C);                                                                                                 \
    BD_REQUIRE(a->Hd <= 16 * 2 * kWaves && a->Hd <= 256, who ": hidden width %d above %d", a->Hd, 16 * 2 * kWaves)

int bd_observe_cat_forward(const bd_observe_cat_fwd_args* a, void* stream) {
    BD_REQUIRE(a && a->T > 0 && a->B > 0 && a->Be > 0 && a->A > 0 && a->Hd > 0, "bd_observe_cat_forward: bad dims");
    BD_CAT_GEO("bd_observe_cat_forward");
    BD_REQUIRE(a->w_embed_sT && a->w_embed_a && a->b_embed && a->w_ir && a->w_iz && a->w_in && a->w_hr && a->w_hz && a->w_hn &&
                   a->b_ih && a->b_hh && a->w_q1h && a->b_q1 && a->w_q2 && a->b_q2, "bd_observe_cat_forward: missing weights");
    BD_REQUIRE(a->init_belief && a->init_state && a->actions && a->pre_emb && a->q_post, "bd_observe_cat_forward: missing inputs");
    BD_REQUIRE(a->feat && a->post_logits && a->sidx, "bd_observe_cat_forward: missing outputs");
    const int Kb_h = cdiv(a->Be, 16), Kb_a = cdiv(a->A, 16), Kb_hd = cdiv(a->Hd, 16);
    const CatFull gf(a->D, a->C);
    const size_t lds = ((size_t)(3 * Kb_h + Kb_hd + Kb_a) * kFragFloats + 16 * a->Be + gf.image_floats() + 2 * 16 * g.D + 16) *
                       sizeof(float);
    BD_REQUIRE(lds <= (size_t)kMaxLds, "bd_observe_cat_forward: needs %zu B of LDS", lds);
    if (lds > 64 * 1024 && allow_big_lds(observe_cat_fwd_kernel)) return -1;
    hipLaunchKernelGGL(observe_cat_fwd_kernel, dim3(cdiv(a->B, 16)), dim3(kThreads), lds, (hipStream_t)stream, *a);
    BD_CHECK_LAUNCH("bd_observe_cat_forward");
    return 0;
}

int bd_observe_cat_backward(const bd_observe_cat_bwd_args* a, void* stream) {
    BD_REQUIRE(a && a->T > 0 && a->B > 0 && a->Be > 0 && a->A > 0 && a->Hd > 0, "bd_observe_cat_backward: bad dims");
    BD_CAT_GEO("bd_observe_cat_backward");
    BD_REQUIRE(a->wt_embed_s && a->wt_ir && a->wt_iz && a->wt_in && a->wt_hr && a->wt_hz && a->wt_hn && a->wt_q1h && a->wt_q2,
               "bd_observe_cat_backward: missing weights");
    BD_REQUIRE(a->init_belief && a->feat && a->post_logits && a->sv_x && a->sv_gates && a->sv_q && a->dfeat,
               "bd_observe_cat_backward: missing forward tensors");
    BD_REQUIRE(a->d_embed_pre && a->d_gi && a->d_gh && a->d_q1_pre && a->d_q2_out, "bd_observe_cat_backward: missing outputs");
    const int Kb_h = cdiv(a->Be, 16), Kb_hd = cdiv(a->Hd, 16);
    const size_t lds = ((size_t)(6 * Kb_h + Kb_hd + g.CW / 16) * kFragFloats + 2 * g.image_floats() + 16) * sizeof(float);
    BD_REQUIRE(lds <= (size_t)kMaxLds, "bd_observe_cat_backward: needs %zu B of LDS", lds);
    if (lds > 64 * 1024 && allow_big_lds(observe_cat_bwd_kernel)) return -1;
    hipLaunchKernelGGL(observe_cat_bwd_kernel, dim3(cdiv(a->B, 16)), dim3(kThreads), lds, (hipStream_t)stream, *a);
    BD_CHECK_LAUNCH("bd_observe_cat_backward");
    return 0;
}

int bd_imagine_cat_forward(const bd_imagine_cat_fwd_args* a, void* stream) {
    BD_REQUIRE(a && a->N > 0 && a->Hm > 0 && a->Be > 0 && a->A > 0 && a->Hd > 0 && a->n_samples > 0, "bd_imagine_cat_forward: bad dims");
    BD_CAT_GEO("bd_imagine_cat_forward");
    BD_REQUIRE(a->A <= kHeadMaxN, "bd_imagine_cat_forward: action width %d above %d", a->A, kHeadMaxN);
    BD_REQUIRE(a->w_embed_sT && a->w_embed_a && a->b_embed && a->w_ir && a->w_iz && a->w_in && a->w_hr && a->w_hz && a->w_hn &&
                   a->b_ih && a->b_hh && a->w_p1 && a->b_p1 && a->w_p2 && a->b_p2 && a->w_a0h && a->w_a0sT && a->w_a[0] &&
                   a->w_a[1] && a->w_a[2] && a->b_a[0] && a->b_a[1] && a->b_a[2] && a->b_a[3] && a->w_a4m && a->w_a4s && a->b_a4,
               "bd_imagine_cat_forward: missing weights");
    BD_REQUIRE(a->start_feat && a->eps_action && a->q_prior && (a->eps_entropy || a->sv_act_stats),
               "bd_imagine_cat_forward: missing inputs");
    BD_REQUIRE(a->feat && a->sidx && a->prior_logits && a->entropy && a->action, "bd_imagine_cat_forward: missing outputs");
    const int Kb_h = cdiv(a->Be, 16), Kb_a = cdiv(a->A, 16), Kb_hd = cdiv(a->Hd, 16);
    const int wmax = a->Be > a->Hd ? a->Be : a->Hd;
    size_t uni = (size_t)kSplitScratchFloats;
    if ((size_t)kWaves * 16 * a->A * 3 > uni) uni = (size_t)kWaves * 16 * a->A * 3;
    const CatFull gf(a->D, a->C);
    if ((size_t)gf.image_floats() > uni) uni = (size_t)gf.image_floats();
    const size_t fixed = (size_t)(3 * Kb_h + 2 * Kb_hd + Kb_a) * kFragFloats + 16 * wmax + 3 * 16 * a->A + 2 * 16 * g.D;
    const size_t lds = (fixed + uni) * sizeof(float);      // every term of `fixed` is a multiple of 16 floats
    BD_REQUIRE(lds <= (size_t)kMaxLds, "bd_imagine_cat_forward: needs %zu B of LDS", lds);
    int grid;
    const size_t dyn = cat_grid(imagine_cat_fwd_kernel, cdiv(a->N, 16), lds, &grid);
    if (!dyn) return -1;
    hipLaunchKernelGGL(imagine_cat_fwd_kernel, dim3(grid), dim3(kThreads), dyn, (hipStream_t)stream, *a);
    BD_CHECK_LAUNCH("bd_imagine_cat_forward");
    if (a->sv_act_stats != nullptr && a->eps_entropy != nullptr)     // the entropy estimate is off the recurrence (imagine.hip)
        return bd_actor_entropy(a->eps_entropy, a->sv_act_stats, a->entropy, a->Hm, a->N, a->A, a->n_samples, stream);
    return 0;
}

int bd_imagine_cat_backward(const bd_imagine_cat_bwd_args* a, void* stream) {
    BD_REQUIRE(a && a->N > 0 && a->Hm > 0 && a->Be > 0 && a->A > 0 && a->Hd > 0, "bd_imagine_cat_backward: bad dims");
    BD_CAT_GEO("bd_imagine_cat_backward");
    BD_REQUIRE(a->wt_embed_s && a->wt_embed_a && a->wt_ir && a->wt_iz && a->wt_in && a->wt_hr && a->wt_hz && a->wt_hn &&
                   a->wt_p1 && a->wt_p2 && a->wt_a[0] && a->wt_a[1] && a->wt_a[2] && a->wt_a4m && a->wt_a4s,
               "bd_imagine_cat_backward: missing weights");
    BD_REQUIRE(a->start_feat && a->feat && a->prior_logits && a->action && a->eps_action && a->sv_actor && a->sv_act_stats &&
                   a->sv_x && a->sv_gates && a->sv_p && a->dfeat, "bd_imagine_cat_backward: missing forward tensors");
    BD_REQUIRE(a->d_actor_out, "bd_imagine_cat_backward: missing outputs");
    const int Kb_h = cdiv(a->Be, 16), Kb_a = cdiv(a->A, 16), Kb_hd = cdiv(a->Hd, 16);
    size_t uni = (size_t)2 * g.image_floats() + (size_t)(g.CW / 16) * kFragFloats;
    if ((size_t)kSplitScratchFloats > uni) uni = (size_t)kSplitScratchFloats;
    if ((size_t)2 * Kb_hd * kFragFloats > uni) uni = (size_t)2 * Kb_hd * kFragFloats;
    const size_t lds = ((size_t)(6 * Kb_h + Kb_hd + 2 * Kb_a) * kFragFloats + uni) * sizeof(float);
    BD_REQUIRE(lds <= (size_t)kMaxLds, "bd_imagine_cat_backward: needs %zu B of LDS", lds);
    int grid;
    const size_t dyn = cat_grid(imagine_cat_bwd_kernel, cdiv(a->N, 16), lds, &grid);
    if (!dyn) return -1;
    hipLaunchKernelGGL(imagine_cat_bwd_kernel, dim3(grid), dim3(kThreads), dyn, (hipStream_t)stream, *a);
    BD_CHECK_LAUNCH("bd_imagine_cat_backward");
    return 0;
}

}  // extern "C"
