// observe.hip -- RSSM observe scan (TransitionModel.forward with embeddings, src/models.py:191-299) and its
// backward (BPTT).  One persistent launch per direction: a workgroup owns 16 batch rows and walks all T
// steps with the belief / state / gradient carries resident in LDS (MFMA fragment order); batch rows are
// independent, so there is no inter-workgroup synchronisation.  Weights (1.2 MB on the critical path) are
// re-streamed from L2 every step in the packed layout.
//
// Per step (src/models.py:241-271):
//   s~ = s * nonterminal                                   (:247)
//   x  = ELU(W_e [s~; a] + b_e)                            (:251)
//   h' = GRUCell(x, h)                                     (:252)
//   q  = ELU(W_q1[:, :Be] h' + pre_emb_t + b_q1)           (:266-267, first layer of belief_posterior)
//   mean, raw = W_q2 q + b_q2; std = softplus(raw)+min_std; s' = mean + std*eps   (models.py:70-73)
#include "bd_device.h"
#include "bd_host.h"

namespace bd {

struct ObsDims {
    int Kb_h, Kb_s, Kb_a, Kb_hd;
    __host__ __device__ ObsDims(int Be, int S, int A, int Hd)
        : Kb_h(cdiv(Be, 16)), Kb_s(cdiv(S, 16)), Kb_a(cdiv(A, 16)), Kb_hd(cdiv(Hd, 16)) {}
};

__global__ __launch_bounds__(kThreads) void observe_fwd_kernel(bd_observe_fwd_args a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const ObsDims d(a.Be, a.S, a.A, a.Hd);
    const int lane = threadIdx.x & 63;
    const int row0 = blockIdx.x * 16;
    const int F = a.Be + a.S;
    float* h_cur = smem;
    float* h_nxt = h_cur + d.Kb_h * kFragFloats;
    float* xf = h_nxt + d.Kb_h * kFragFloats;
    float* qf = xf + d.Kb_h * kFragFloats;
    float* sf = qf + d.Kb_hd * kFragFloats;
    float* af = sf + d.Kb_s * kFragFloats;
    float* s_plain = af + d.Kb_a * kFragFloats;   // [16][S] unmasked posterior state of the previous step
    float* scratch = s_plain + 16 * a.S;          // split-K partials (kSplitScratchFloats)

    load_tile_concat<1>(h_cur, d.Kb_h, row0, a.B, a.init_belief, a.Be, a.Be, nullptr, 0, 0);
    for (int i = threadIdx.x; i < 16 * a.S; i += blockDim.x) {
        const int r = i / a.S, k = i - r * a.S;
        s_plain[i] = (row0 + r < a.B) ? a.init_state[(size_t)(row0 + r) * a.S + k] : 0.f;
    }
    lds_barrier();

    const GruW gw{a.w_ir, a.w_iz, a.w_in, a.w_hr, a.w_hz, a.w_hn, a.b_ih, a.b_hh};

    for (int t = 0; t < a.T; ++t) {
        const size_t tb = (size_t)t * a.B;
        // ---- A: masked state and action fragments ----
        for (int i = threadIdx.x; i < 16 * d.Kb_s * 16; i += blockDim.x) {
            const int r = i / (d.Kb_s * 16), k = i - r * (d.Kb_s * 16);
            const int grow = row0 + r;
            float v = 0.f;
            if (grow < a.B && k < a.S) {
                v = s_plain[r * a.S + k];
                if (a.nonterm) v *= a.nonterm[tb + grow];
                if (a.sv_s) a.sv_s[(tb + grow) * a.S + k] = v;
            }
            sf[frag_idx(r, k)] = v;
        }
        for (int i = threadIdx.x; i < 16 * d.Kb_a * 16; i += blockDim.x) {
            const int r = i / (d.Kb_a * 16), k = i - r * (d.Kb_a * 16);
            const int grow = row0 + r;
            af[frag_idx(r, k)] = (grow < a.B && k < a.A) ? a.actions[(tb + grow) * a.A + k] : 0.f;
        }
        lds_barrier();
        // ---- B: x = ELU(W_e [s~; a] + b_e) ----
        {
            const Seg segs[2] = {{sf, a.w_embed_s, d.Kb_s}, {af, a.w_embed_a, d.Kb_a}};
            tile_linear_seg<2>(segs, a.b_embed, a.Be, [&](int nb, floatx4 acc) {
                const int col = nb * 16 + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int grow = row0 + 4 * (lane >> 4) + r;
                    const float v = elu(acc[r]);
                    xf[acc_frag_off(nb, lane, r)] = v;
                    if (a.sv_x && grow < a.B && col < a.Be) a.sv_x[(tb + grow) * a.Be + col] = v;
                }
            });
        }
        lds_barrier();
        // ---- C: GRU ----
        gru_tile(xf, h_cur, d.Kb_h, a.Be, gw, [&](int nb, floatx4 R, floatx4 Z, floatx4 NI, floatx4 NH) {
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
                        float* g = a.sv_gates + (tb + grow) * 4 * a.Be + col;
                        g[0] = rr; g[a.Be] = zz; g[2 * a.Be] = nn; g[3 * a.Be] = NH[r];
                    }
                }
            }
        }, scratch);
        lds_barrier();
        // ---- D: posterior hidden ----
        tile_linear<1>(h_nxt, d.Kb_h, a.w_q1h, a.b_q1, a.Hd, [&](int, int nb, floatx4 acc) {
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
        // ---- E: posterior mean / std / sample ----
        {
            const Seg2 segs[1] = {{qf, a.w_q2m, a.w_q2s, d.Kb_hd}};
            tile_linear_dual<1>(segs, a.b_q2, a.b_q2 + a.S, a.S, [&](int nb, floatx4 Mn, floatx4 Rw) {
                const int col = nb * 16 + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 4 * (lane >> 4) + r, grow = row0 + row;
                    if (col >= a.S) continue;
                    float st = 0.f;
                    if (grow < a.B) {
                        const size_t i = (tb + grow) * a.S + col;
                        const float sd = softplusf(Rw[r]) + a.min_std;
                        st = Mn[r] + sd * a.eps_post[i];
                        a.post_mean[i] = Mn[r];
                        a.post_std[i] = sd;
                        a.feat[(tb + grow) * F + a.Be + col] = st;
                    }
                    s_plain[row * a.S + col] = st;
                }
            }, scratch);
        }
        lds_barrier();
        float* tmp = h_cur; h_cur = h_nxt; h_nxt = tmp;
    }
}

// ---- backward ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void observe_bwd_kernel(bd_observe_bwd_args a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const ObsDims d(a.Be, a.S, a.A, a.Hd);
    const int lane = threadIdx.x & 63;
    const int row0 = blockIdx.x * 16;
    const int F = a.Be + a.S;
    const int nh = d.Kb_h * kFragFloats, ns = d.Kb_s * kFragFloats;
    float* dhc = smem;            // carry: d loss / d belief_t (through the recurrence)
    float* dR = dhc + nh;
    float* dZ = dR + nh;
    float* dNI = dZ + nh;
    float* dNH = dNI + nh;
    float* dE = dNH + nh;
    float* dQ = dE + nh;                               // Kb_hd
    float* dM = dQ + d.Kb_hd * kFragFloats;            // Kb_s
    float* dRaw = dM + ns;
    float* ds_plain = dRaw + ns;                       // [16][S] carry: d loss / d posterior_state_t
    float* scratch = ds_plain + 16 * a.S;              // split-K partials (kSplitScratchFloats)

    for (int i = threadIdx.x; i < nh; i += blockDim.x) dhc[i] = 0.f;
    for (int i = threadIdx.x; i < 16 * a.S; i += blockDim.x) ds_plain[i] = 0.f;
    lds_barrier();

    const GruWT gw{a.wt_ir, a.wt_iz, a.wt_in, a.wt_hr, a.wt_hz, a.wt_hn};

    for (int t = a.T - 1; t >= 0; --t) {
        const size_t tb = (size_t)t * a.B;
        // ---- 1: through the sample / softplus into (mean, raw) ----
        for (int i = threadIdx.x; i < 16 * d.Kb_s * 16; i += blockDim.x) {
            const int r = i / (d.Kb_s * 16), k = i - r * (d.Kb_s * 16);
            const int grow = row0 + r;
            float dm = 0.f, dr = 0.f;
            if (grow < a.B && k < a.S) {
                const size_t idx = (tb + grow) * a.S + k;
                const float dst = ds_plain[r * a.S + k] + a.dfeat[(tb + grow) * F + a.Be + k];
                dm = dst + (a.dpost_mean ? a.dpost_mean[idx] : 0.f);
                const float dsd = dst * a.eps_post[idx] + (a.dpost_std ? a.dpost_std[idx] : 0.f);
                // sigmoid(raw) from std = softplus(raw) + min_std:  1 - exp(-softplus(raw))
                dr = dsd * one_minus_exp_neg(a.post_std[idx] - a.min_std);
                a.d_q2_out[(tb + grow) * 2 * a.S + k] = dm;
                a.d_q2_out[(tb + grow) * 2 * a.S + a.S + k] = dr;
            }
            dM[frag_idx(r, k)] = dm;
            dRaw[frag_idx(r, k)] = dr;
        }
        lds_barrier();
        // ---- 2: d q (posterior hidden) ----
        {
            const Seg segs[2] = {{dM, a.wt_q2m, d.Kb_s}, {dRaw, a.wt_q2s, d.Kb_s}};
            tile_linear_seg<2>(segs, nullptr, a.Hd, [&](int nb, floatx4 acc) {
                const int col = nb * 16 + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int grow = row0 + 4 * (lane >> 4) + r;
                    float v = 0.f;
                    if (grow < a.B && col < a.Hd) {
                        v = acc[r] * elu_grad_from_out(a.sv_q[(tb + grow) * a.Hd + col]);
                        a.d_q1_pre[(tb + grow) * a.Hd + col] = v;
                    }
                    dQ[acc_frag_off(nb, lane, r)] = v;
                }
            });
        }
        lds_barrier();
        // ---- 3: total d belief_{t+1}, GRU gate gradients ----
        tile_linear<1>(dQ, d.Kb_hd, a.wt_q1h, nullptr, a.Be, [&](int, int nb, floatx4 acc) {
            const int col = nb * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int grow = row0 + 4 * (lane >> 4) + r;
                const int off = acc_frag_off(nb, lane, r);
                float vr = 0.f, vz = 0.f, vni = 0.f, vnh = 0.f, carry = 0.f;
                if (grow < a.B && col < a.Be) {
                    const float dh = acc[r] + dhc[off] + a.dfeat[(tb + grow) * F + col];
                    const float* g = a.sv_gates + (tb + grow) * 4 * a.Be + col;
                    const float rr = g[0], zz = g[a.Be], nn = g[2 * a.Be], hn = g[3 * a.Be];
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
        // ---- 4: through W_ih / W_hh ----
        gru_tile_bwd(
            dR, dZ, dNI, dNH, d.Kb_h, a.Be, gw,
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
            }, scratch);
        lds_barrier();
        // ---- 5: d posterior_state_t through the embed layer and the nonterminal mask ----
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
    }
}

}  // namespace bd

extern "C" {
using namespace bd;

int bd_observe_forward(const bd_observe_fwd_args* a, void* stream) {
    BD_REQUIRE(a && a->T > 0 && a->B > 0 && a->Be > 0 && a->S > 0 && a->A > 0 && a->Hd > 0, "bd_observe_forward: bad dims");
    BD_REQUIRE(a->w_embed_s && a->w_embed_a && a->b_embed && a->w_ir && a->w_iz && a->w_in && a->w_hr && a->w_hz &&
                   a->w_hn && a->b_ih && a->b_hh && a->w_q1h && a->b_q1 && a->w_q2m && a->w_q2s && a->b_q2,
               "bd_observe_forward: missing weights");
    BD_REQUIRE(a->init_belief && a->init_state && a->actions && a->pre_emb && a->eps_post,
               "bd_observe_forward: missing inputs");
    BD_REQUIRE(a->feat && a->post_mean && a->post_std, "bd_observe_forward: missing outputs");
    const ObsDims d(a->Be, a->S, a->A, a->Hd);
    const size_t lds = ((size_t)(3 * d.Kb_h + d.Kb_hd + d.Kb_s + d.Kb_a) * kFragFloats + 16 * a->S + kSplitScratchFloats) *
                       sizeof(float);
    BD_REQUIRE(lds <= (size_t)kMaxLds, "bd_observe_forward: needs %zu B of LDS", lds);
    if (lds > 64 * 1024 && allow_big_lds(observe_fwd_kernel)) return -1;
    hipLaunchKernelGGL(observe_fwd_kernel, dim3(cdiv(a->B, 16)), dim3(kThreads), lds, (hipStream_t)stream, *a);
    BD_CHECK_LAUNCH("bd_observe_forward");
    return 0;
}

int bd_observe_backward(const bd_observe_bwd_args* a, void* stream) {
    BD_REQUIRE(a && a->T > 0 && a->B > 0 && a->Be > 0 && a->S > 0 && a->A > 0 && a->Hd > 0, "bd_observe_backward: bad dims");
    BD_REQUIRE(a->wt_embed_s && a->wt_ir && a->wt_iz && a->wt_in && a->wt_hr && a->wt_hz && a->wt_hn && a->wt_q1h &&
                   a->wt_q2m && a->wt_q2s, "bd_observe_backward: missing weights");
    BD_REQUIRE(a->init_belief && a->eps_post && a->feat && a->post_std && a->sv_x && a->sv_gates && a->sv_q && a->dfeat,
               "bd_observe_backward: missing forward tensors");
    BD_REQUIRE(a->d_embed_pre && a->d_gi && a->d_gh && a->d_q1_pre && a->d_q2_out, "bd_observe_backward: missing outputs");
    const ObsDims d(a->Be, a->S, a->A, a->Hd);
    const size_t lds = ((size_t)(6 * d.Kb_h + d.Kb_hd + 2 * d.Kb_s) * kFragFloats + 16 * a->S + kSplitScratchFloats) *
                       sizeof(float);
    BD_REQUIRE(lds <= (size_t)kMaxLds, "bd_observe_backward: needs %zu B of LDS", lds);
    if (lds > 64 * 1024 && allow_big_lds(observe_bwd_kernel)) return -1;
    hipLaunchKernelGGL(observe_bwd_kernel, dim3(cdiv(a->B, 16)), dim3(kThreads), lds, (hipStream_t)stream, *a);
    BD_CHECK_LAUNCH("bd_observe_backward");
    return 0;
}

}  // extern "C"
