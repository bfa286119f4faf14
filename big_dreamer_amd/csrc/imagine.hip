// imagine.hip -- imagination rollout (Dreamer.imagine_ahead, src/dreamer.py:179-237) with the actor
// (ActorModel, src/models.py:506-517), the tanh-Normal action sample and its 100-sample entropy estimate
// (Dreamer.get_action src/dreamer.py:429-444; SampleDist.entropy / TanhBijector src/models.py:630-733), and
// the backward pass through the whole rollout.  One persistent launch per direction; a workgroup owns 16
// trajectories for all Hm steps (trajectories are independent: no inter-workgroup synchronisation), all
// per-step vectors stay in LDS in MFMA fragment order, weights stream from L2.
//
// Per step t (src/dreamer.py:213-227), from (h, s):
//   actor: 4 x (Linear+ELU) on [h; s] (detached), out -> mean = 5 tanh(m/5), std = softplus(r + c0) + 1e-4
//   a = tanh(mean + std*eps_a);   entropy = -mean_k log p(tanh(mean + std*eps_k))
//   x = ELU(W_e [s; a] + b_e);  h' = GRUCell(x, h);  p = ELU(W_p1 h' + b);  s' = mean_p + std_p * eps_p
#include "bd_device.h"
#include "bd_host.h"
#include "bd_scan.h"
#include "bd_rng.h"

namespace bd {

constexpr int kMaxA = 64;

// Diagnostic build only (-DBD_STAMPS): workgroup 0 / thread 0 records s_memtime at the phase boundaries of one
// step into a buffer that nothing else reads (bd_debug_stamps copies it out).  Never defined in the shipped .so.
#ifdef BD_STAMPS
__device__ unsigned long long g_stamps[64];
#define BD_STAMP(slot)                                                                      \
    do {                                                                                    \
        if (blockIdx.x == BD_STAMP_WG && threadIdx.x == 0 && t == 3) g_stamps[slot] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#ifndef BD_STAMP_WG
#define BD_STAMP_WG 0
#endif
#else
#define BD_STAMP(slot)
#endif

struct ImgDims {
    int Kb_h, Kb_s, Kb_a, Kb_hd;
    __host__ __device__ ImgDims(int Be, int S, int A, int Hd)
        : Kb_h(cdiv(Be, 16)), Kb_s(cdiv(S, 16)), Kb_a(cdiv(A, 16)), Kb_hd(cdiv(Hd, 16)) {}
};


__global__ __launch_bounds__(kThreads) void imagine_fwd_kernel(bd_imagine_fwd_args a_) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    BD_KARGS(bd_imagine_fwd_args, ap);
#define a (*ap)
    const ImgDims d(a.Be, a.S, a.A, a.Hd);
    const int row0 = blockIdx.x * 16;
    const int F = a.Be + a.S, A = a.A;
    const int nh = d.Kb_h * kFragFloats, nhd = d.Kb_hd * kFragFloats;
    float* h_cur = smem;
    float* h_nxt = h_cur + nh;
    float* xf = h_nxt + nh;
    float* bufA = xf + nh;
    float* bufB = bufA + nhd;
    float* sf = bufB + nhd;
    float* af = sf + d.Kb_s * kFragFloats;
    float* mean_s = af + d.Kb_a * kFragFloats;    // [16][A]
    float* std_s = mean_s + 16 * A;               // [16][A]
    float* lp_rj = std_s + 16 * A;                // [16][A]
    float* part = lp_rj + 16 * A;                 // [kWaves][16][A][3]
    float* scratch = part + kWaves * 16 * A * 3;  // split-K partials (kSplitScratchFloats); 16-byte aligned: see host

    load_tile_concat<1>(h_cur, d.Kb_h, row0, a.N, a.start_feat, F, a.Be, nullptr, 0, 0);
    load_tile_concat<1>(sf, d.Kb_s, row0, a.N, a.start_feat + a.Be, F, a.S, nullptr, 0, 0);
    for (int i = threadIdx.x; i < d.Kb_a * kFragFloats; i += blockDim.x) af[i] = 0.f;   // columns >= A stay zero
    lds_barrier();

    const size_t act_stride = a.sv_actor_stride ? a.sv_actor_stride : (size_t)a.Hm * a.N * a.Hd;
    const float inv_ns = 1.f / (float)a.n_samples;

    for (int t = 0; t < a.Hm; ++t) {
        const size_t tn = (size_t)t * a.N;
        const int tid = bd_tid();                   // opaque: nothing thread-dependent leaves this step (bd_tid)
        const int lane = tid & 63, wave = bd_wave(tid);
        // hidden layer epilogue: ELU -> LDS fragment (+ optional save)
        auto hidden_epi = [&](float* dst, float* save, size_t tn_, int width) {
            return HiddenEpiTR{dst, save, tn_, width, a.N, row0, lane};
        };
        BD_STAMP(0);
        BD_KARGS_FRESH(ap);
        // ---- actor hidden layers ----
        {
            const Seg segs[2] = {{h_cur, a.w_a0h, d.Kb_h}, {sf, a.w_a0s, d.Kb_s}};
            tile_linear_seg_tr<2>(segs, a.b_a[0], a.Hd, hidden_epi(bufA, a.sv_actor, tn, a.Hd));
        }
        BD_STAMP(1);
        lds_barrier();
        BD_STAMP(2);
        BD_KARGS_FRESH(ap);
        {
            float* src = bufA;
            float* dst = bufB;
            for (int l = 1; l < 4; ++l) {
                const Seg segs[1] = {{src, a.w_a[l - 1], d.Kb_hd}};
                tile_linear_seg_tr<1>(segs, a.b_a[l], a.Hd,
                                      hidden_epi(dst, a.sv_actor ? a.sv_actor + l * act_stride : nullptr, tn, a.Hd),
                                      (t == 3 && l == 2) ? 32 : -1);
                BD_DSTAMP((t == 3 && l == 2) ? 32 : -1, 4);
                lds_barrier();
                BD_DSTAMP((t == 3 && l == 2) ? 32 : -1, 5);
                float* tmp = src; src = dst; dst = tmp;
            }
            // after 3 layers the activations of layer 3 are in bufB (A->B, B->A, A->B)
        }
        BD_STAMP(3);
        BD_KARGS_FRESH(ap);
        // ---- actor output, action sample ----
        {
            const Seg2 segs[1] = {{bufB, a.w_a4m, a.w_a4s, d.Kb_hd}};
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
                        if (a.sv_act_stats) {      // slots 2, 3: mean and std for actor_entropy_kernel, which replaces them
                            float* st = a.sv_act_stats + (tn + grow) * 4 * A + col;
                            st[0] = th;
                            st[A] = sigmoidf(pre);
                            st[2 * A] = mean;
                            st[3 * A] = sd;
                        }
                    }
                    af[frag_idx(row, col)] = act;
                },
                t == 3 ? 0 : -1);
        }
        BD_STAMP(4);
        lds_barrier();
        BD_STAMP(5);
        BD_KARGS_FRESH(ap);
        // ---- entropy: n_samples draws per (row, action dim) ----
        // With saved actor statistics (training) the estimate is NOT on the recurrence: it needs only (mean, std) of this
        // step, so it runs after the scan as one elementwise launch over all Hm x N rows (actor_entropy_kernel) instead of
        // 10k cycles of every step of every tile (s_memtime stamps: 7 % of the step).  Without them (acting: one step, no
        // backward) the estimate stays here; thread = (row, sample lane).
        if (a.sv_act_stats == nullptr) {
            const int row = tid & 15, sl = tid >> 4;   // 16 sample lanes
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
                // the 4 sample lanes of this wave that share `row` sit 16 lanes apart
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
            lds_barrier();
            if (tid < 16 && row0 + tid < a.N) {
                float s = 0.f;
                for (int j = 0; j < A; ++j) s += lp_rj[tid * A + j];
                a.entropy[tn + row0 + tid] = -s * inv_ns;
            }
        }
        BD_STAMP(6);
        BD_KARGS_FRESH(ap);
        // ---- embed ----
        {
            const Seg segs[2] = {{sf, a.w_embed_s, d.Kb_s}, {af, a.w_embed_a, d.Kb_a}};
            tile_linear_seg_tr<2>(segs, a.b_embed, a.Be, hidden_epi(xf, a.sv_x, tn, a.Be));
        }
        BD_STAMP(7);
        lds_barrier();
        BD_STAMP(8);
        BD_KARGS_FRESH(ap);
        // ---- GRU ----
        const GruW gw{a.w_ir, a.w_iz, a.w_in, a.w_hr, a.w_hz, a.w_hn, a.b_ih, a.b_hh};
        gru_tile(xf, h_cur, d.Kb_h, a.Be, gw, [&](int nb, floatx4 R, floatx4 Z, floatx4 NI, floatx4 NH) {
            const int c = lane & 15, q = lane >> 4;
            const int off0 = acc_frag_off(nb, lane, 0);
            if (row0 + 16 <= a.N && nb * 16 + 16 <= a.Be) {      // uniform: block wholly inside the matrix
                float rr[4], zz[4], nn[4], hn[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    rr[r] = sigmoidf(R[r]);
                    zz[r] = sigmoidf(Z[r]);
                    nn[r] = tanh_act(NI[r] + rr[r] * NH[r]);
                    hn[r] = (1.f - zz[r]) * nn[r] + zz[r] * h_cur[off0 + 4 * r];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) h_nxt[off0 + 4 * r] = hn[r];
                const unsigned lo = (unsigned)(row0 + 4 * q), cc = (unsigned)(nb * 16 + c);
                float* __restrict__ f = a.feat + tn * F + (lo * (unsigned)F + cc);
#pragma unroll
                for (int r = 0; r < 4; ++r) st_save(f + (unsigned)r * (unsigned)F, hn[r]);
                if (a.sv_gates) {
                    const unsigned Be = (unsigned)a.Be;
                    float* __restrict__ g = a.sv_gates + tn * 4 * a.Be + (lo * 4u * Be + cc);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float* gr = g + (unsigned)r * 4u * Be;
                        st_save(gr, rr[r]); st_save(gr + Be, zz[r]); st_save(gr + 2u * Be, nn[r]); st_save(gr + 3u * Be, NH[r]);
                    }
                }
                return;
            }
            const int col = nb * 16 + c;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int grow = row0 + 4 * q + r;
                const int off = off0 + 4 * r;
                const float rr = sigmoidf(R[r]), zz = sigmoidf(Z[r]);
                const float nn = tanh_act(NI[r] + rr * NH[r]);
                const float hn = (1.f - zz) * nn + zz * h_cur[off];
                const bool ok = grow < a.N && col < a.Be;
                h_nxt[off] = ok ? hn : 0.f;
                if (ok) {
                    a.feat[(tn + grow) * F + col] = hn;
                    if (a.sv_gates) {
                        float* g = a.sv_gates + (tn + grow) * 4 * a.Be + col;
                        g[0] = rr; g[a.Be] = zz; g[2 * a.Be] = nn; g[3 * a.Be] = NH[r];
                    }
                }
            }
        }, scratch);
        BD_STAMP(9);
        lds_barrier();
        BD_STAMP(10);
        BD_KARGS_FRESH(ap);
        // ---- prior ----
        {
            const Seg segs[1] = {{h_nxt, a.w_p1, d.Kb_h}};
            tile_linear_seg_tr<1>(segs, a.b_p1, a.Hd, hidden_epi(bufA, a.sv_p, tn, a.Hd));
        }
        BD_STAMP(11);
        lds_barrier();
        BD_STAMP(12);
        BD_KARGS_FRESH(ap);
        {
            const Seg2 segs[1] = {{bufA, a.w_p2m, a.w_p2s, d.Kb_hd}};
            tile_dual_head_elem<1>(
                segs, a.b_p2, a.b_p2 + a.S, a.S, scratch,
                [&](int row, int col) { return row0 + row < a.N ? a.eps_prior[(tn + row0 + row) * a.S + col] : 0.f; },
                [&](int row, int col, float Mn, float Rw, float eps) {
                    const int grow = row0 + row;
                    float st = 0.f;
                    if (grow < a.N) {
                        const size_t i = (tn + grow) * a.S + col;
                        const float sd = softplusf(Rw) + a.min_std;
                        st = Mn + sd * eps;
                        if (a.prior_mean) a.prior_mean[i] = Mn;
                        a.prior_std[i] = sd;
                        a.feat[(tn + grow) * F + a.Be + col] = st;
                    }
                    sf[frag_idx(row, col)] = st;
                },
                t == 3 ? 8 : -1);
        }
        BD_STAMP(13);
        lds_barrier();
        BD_STAMP(14);
        float* tmp = h_cur; h_cur = h_nxt; h_nxt = tmp;
    }
#undef a
}

// ---- entropy estimate of the imagined actions, off the recurrence -----------------------------------------------------
// entropy[t][n] = -mean_k log p(tanh(mean + std * eps_k)) summed over the action dimensions, and its derivatives w.r.t.
// mean and std (SampleDist.entropy / TanhBijector, src/models.py:630-733) from the (mean, std) the scan left in slots 2, 3
// of the actor statistics; it overwrites them with d entropy / d mean, d entropy / d std for the backward scan.
// One workgroup = 64 / A rows; lane = (row, action dim), the sixteen waves take the draws k = wave, wave + 16, ...;
// partials are summed in fixed order (deterministic).
constexpr int kEntParts = 16;
// RNG = true (perf mode): the draws come from Philox (bd_rng.h) instead of `eps`: sample lane `part` of (row, action dim)
// generates its ceil(ns / kEntParts) draws four at a time -- the entropy noise tensor never exists in HBM.
template <bool RNG>
__global__ __launch_bounds__(64 * kEntParts) void actor_entropy_kernel(const float* __restrict__ eps, float* __restrict__ stats,
                                                            float* __restrict__ entropy, int Hm, int N, int A, int ns, Rng rng) {
    __shared__ float red[kEntParts][64][3];
    __shared__ float lpj[64];
    const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int rows_pb = 64 / A;
    const int rl = lane / A, j = lane - rl * A;
    const long row = (long)blockIdx.x * rows_pb + rl;            // index over Hm x N
    const bool valid = rl < rows_pb && row < (long)Hm * N;
    float lp = 0.f, dm = 0.f, ds = 0.f;
    if (valid) {
        const int t = (int)(row / N), n = (int)(row - (long)t * N);
        const float* st = stats + (size_t)row * 4 * A + j;
        const EntConst ec = entropy_const(st[2 * A], st[3 * A]);
        if constexpr (RNG) {
            const int mine = (ns - part + kEntParts - 1) / kEntParts;      // samples k = part, part + kEntParts, ... < ns
            const uint64_t base = (((uint64_t)row * A + j) * kEntParts + part) * (uint64_t)((ns / kEntParts + 4) / 4);
            for (int k0 = 0; k0 < mine; k0 += 4) {
                float e[4];
                rng_normal4(rng, base + (k0 >> 2), e);
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (k0 + q < mine) {
                        float l1, d1, d2;
                        entropy_sample(ec, e[q], l1, d1, d2);
                        lp += l1; dm += d1; ds += d2;
                    }
            }
        } else {
            const float* e0 = eps + ((size_t)t * ns * N + n) * A + j;
            for (int k = part; k < ns; k += kEntParts) {
                float l1, d1, d2;
                entropy_sample(ec, e0[(size_t)k * N * A], l1, d1, d2);
                lp += l1; dm += d1; ds += d2;
            }
        }
    }
    red[part][lane][0] = lp; red[part][lane][1] = dm; red[part][lane][2] = ds;
    __syncthreads();
    const float inv_ns = 1.f / (float)ns;
    if (part == 0) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        for (int w = 0; w < kEntParts; ++w) { a0 += red[w][lane][0]; a1 += red[w][lane][1]; a2 += red[w][lane][2]; }
        lpj[lane] = a0;
        if (valid) {
            float* st = stats + (size_t)row * 4 * A + j;
            st[2 * A] = -a1 * inv_ns;     // d entropy / d mean
            st[3 * A] = -a2 * inv_ns;     // d entropy / d std
        }
    }
    __syncthreads();
    if (part == 0 && valid && j == 0) {
        float s = 0.f;
        for (int jj = 0; jj < A; ++jj) s += lpj[lane + jj];
        entropy[row] = -s * inv_ns;
    }
}

// ---- backward --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void imagine_bwd_kernel(bd_imagine_bwd_args a_) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    BD_KARGS(bd_imagine_bwd_args, ap);
#define a (*ap)
    const ImgDims d(a.Be, a.S, a.A, a.Hd);
    const int row0 = blockIdx.x * 16;
    const int F = a.Be + a.S, A = a.A;
    const int nh = d.Kb_h * kFragFloats, nhd = d.Kb_hd * kFragFloats, ns = d.Kb_s * kFragFloats,
              na = d.Kb_a * kFragFloats;
    float* dhc = smem;
    float* dR = dhc + nh;
    float* dZ = dR + nh;
    float* dNI = dZ + nh;
    float* dNH = dNI + nh;
    float* dE = dNH + nh;
    float* dP = dE + nh;          // Kb_hd
    float* bufA = dP + nhd;
    float* bufB = bufA + nhd;
    float* dM = bufB + nhd;       // Kb_s
    float* dRaw = dM + ns;
    float* dAm = dRaw + ns;       // Kb_a
    float* dAr = dAm + na;
    float* ds_plain = dAr + na;   // [16][S]
    float* scratch = ds_plain + 16 * a.S;   // split-K partials (kSplitScratchFloats)

    for (int i = threadIdx.x; i < nh; i += blockDim.x) dhc[i] = 0.f;
    for (int i = threadIdx.x; i < 16 * a.S; i += blockDim.x) ds_plain[i] = 0.f;
    lds_barrier();

    const size_t act_stride = (size_t)a.Hm * a.N * a.Hd;

    for (int t = a.Hm - 1; t >= 0; --t) {
        const size_t tn = (size_t)t * a.N;
        const int tid = bd_tid();                   // opaque: nothing thread-dependent leaves this step (bd_tid)
        const int lane = tid & 63;
        auto dpre_epi = [&](float* dst, float* out, size_t tn_, int width) {
            return DpreEpi{dst, out, tn_, width, a.N, row0, lane};
        };
        auto dpre_pre = [&](const float* saved, size_t tn_, int width) {
            return DprePre{saved, tn_, width, a.N, row0, lane};
        };
        BD_KARGS_FRESH(ap);
        BD_STAMP(20);
        // ---- 1: prior sample -> (mean, raw) ----
        for (int i = tid; i < 16 * d.Kb_s * 16; i += blockDim.x) {
            const int r = i / (d.Kb_s * 16), k = i - r * (d.Kb_s * 16);
            const int grow = row0 + r;
            float dm = 0.f, dr = 0.f;
            if (grow < a.N && k < a.S) {
                const size_t idx = (tn + grow) * a.S + k;
                dm = ds_plain[r * a.S + k] + a.dfeat[(tn + grow) * F + a.Be + k];
                dr = dm * a.eps_prior[idx] * one_minus_exp_neg(a.prior_std[idx] - a.min_std);
            }
            dM[frag_idx(r, k)] = dm;
            dRaw[frag_idx(r, k)] = dr;
        }
        lds_barrier();
        BD_KARGS_FRESH(ap);
        BD_STAMP(21);
        // ---- 2: prior hidden ----
        {
            const Seg segs[2] = {{dM, a.wt_p2m, d.Kb_s}, {dRaw, a.wt_p2s, d.Kb_s}};
            tile_linear_pre<1, 2>(segs, nullptr, a.Hd, dpre_pre(a.sv_p, tn, a.Hd), dpre_epi(dP, nullptr, tn, a.Hd));
        }
        lds_barrier();
        BD_KARGS_FRESH(ap);
        BD_STAMP(22);
        // ---- 3: total d belief_{t+1}; GRU gates ----
        {
            const Seg segs3[1] = {{dP, a.wt_p1, d.Kb_hd}};
            tile_linear_pre<1, 1>(
                segs3, nullptr, a.Be,
                [&](int, int nb) {
                    PreGate p;
                    const int col = nb * 16 + (lane & 15);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int grow = row0 + 4 * (lane >> 4) + r;
                        const bool ok = grow < a.N && col < a.Be;
                        const float* g = a.sv_gates + (tn + grow) * 4 * a.Be + col;
                        p.r[r] = ok ? g[0] : 0.f;
                        p.z[r] = ok ? g[a.Be] : 0.f;
                        p.n[r] = ok ? g[2 * a.Be] : 0.f;
                        p.hn[r] = ok ? g[3 * a.Be] : 0.f;
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
        BD_STAMP(23);
        // ---- 4: through W_ih / W_hh ----
        const GruWT gw{a.wt_ir, a.wt_iz, a.wt_in, a.wt_hr, a.wt_hz, a.wt_hn};
        gru_tile_bwd(
            dR, dZ, dNI, dNH, d.Kb_h, a.Be, gw,
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
            }, scratch);
        lds_barrier();
        BD_KARGS_FRESH(ap);
        BD_STAMP(24);
        // ---- 5: embed layer -> d state_t (carry) and d action_t -> actor output gradients ----
        tile_linear<1>(dE, d.Kb_h, a.wt_embed_s, nullptr, a.S, [&](int, int nb, floatx4 acc) {
            const int col = nb * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * (lane >> 4) + r;
                if (col < a.S) ds_plain[row * a.S + col] = (row0 + row < a.N) ? acc[r] : 0.f;
            }
        }, scratch);
        lds_barrier();   // the split-K scratch is reused by the next contraction
        {
            const Seg segs5[1] = {{dE, a.wt_embed_a, d.Kb_h}};
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
                            const float dxa = acc[r] * (1.f - p.act[r] * p.act[r]);          // through a = tanh(x)
                            const float dent = a.ent_weight ? a.dentropy * a.ent_weight[tn + grow] : a.dentropy;
                            const float dmean = dxa + dent * p.dm[r];
                            const float dstd = dxa * p.eps[r] + dent * p.ds[r];
                            gm = dmean * (1.f - p.th[r] * p.th[r]);      // mean = scale * tanh(m / scale)
                            gr = dstd * p.sg[r];                          // std = softplus(r + c0) + min
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
        BD_STAMP(25);
        // ---- 6: actor MLP backward (input detached: no gradient below layer 0) ----
        // Its result feeds nothing on the recurrence (the actor's input is detached: it only produces the pre-activation
        // gradients for the weight-gradient GEMMs), so with d_actor_pre == NULL the caller runs it as ONE dense chain over
        // all Hm x N rows from d_actor_out (bd_mlp_backward, tall form: 0.14 ms on the whole chip) instead of 45k of the
        // 131k cycles of every step of every tile (s_memtime stamps).
        if (a.d_actor_pre != nullptr) {
        {
            const Seg segs[2] = {{dAm, a.wt_a4m, d.Kb_a}, {dAr, a.wt_a4s, d.Kb_a}};
            tile_linear_pre<1, 2>(segs, nullptr, a.Hd, dpre_pre(a.sv_actor + 3 * act_stride, tn, a.Hd),
                                  dpre_epi(bufA, a.d_actor_pre + 3 * act_stride, tn, a.Hd));
        }
        lds_barrier();
        {
            float* src = bufA;
            float* dst = bufB;
            for (int l = 2; l >= 0; --l) {
                const Seg segs[1] = {{src, a.wt_a[l], d.Kb_hd}};
                tile_linear_pre<1, 1>(segs, nullptr, a.Hd, dpre_pre(a.sv_actor + l * act_stride, tn, a.Hd),
                                      dpre_epi(l > 0 ? dst : nullptr, a.d_actor_pre + l * act_stride, tn, a.Hd));
                lds_barrier();
                float* tmp = src; src = dst; dst = tmp;
            }
        }
        }
        BD_STAMP(26);
    }
#undef a
}

}  // namespace bd

namespace bd {
// BD_IMG_EXCLUSIVE=1 (experiment): a scan whose tiles fit the chip in one round asks for the CU's whole LDS, so that no
// LDS-using workgroup of another stream shares a CU (and its issue slots) with a tile for the length of the launch.
static size_t img_lds(size_t need, int tiles) {
    static const char* e = getenv("BD_IMG_EXCLUSIVE");
    return (e && e[0] == '1' && tiles <= 256) ? (size_t)kMaxLds : need;
}
}  // namespace bd

extern "C" {
using namespace bd;

#ifdef BD_STAMPS
// diagnostic: copy the s_memtime stamps out (synchronises the device)
int bd_debug_stamps(unsigned long long* out64) {
    return hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : -1;
}
int bd_debug_dstamps(unsigned long long* out64) {
    return hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_dstamps), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : -1;
}
#endif

static int imagine_forward_scan(const bd_imagine_fwd_args* a, void* stream) {
    BD_REQUIRE(a && a->N > 0 && a->Hm > 0 && a->Be > 0 && a->S > 0 && a->A > 0 && a->A <= kMaxA && a->Hd > 0 &&
                   a->n_samples > 0, "bd_imagine_forward: bad dims");
    BD_REQUIRE(a->S <= kHeadMaxN && a->A <= kHeadMaxN, "bd_imagine_forward: state / action width above %d", kHeadMaxN);
    BD_REQUIRE(a->w_embed_s && a->w_embed_a && a->b_embed && a->w_ir && a->w_iz && a->w_in && a->w_hr && a->w_hz &&
                   a->w_hn && a->b_ih && a->b_hh && a->w_p1 && a->b_p1 && a->w_p2m && a->w_p2s && a->b_p2,
               "bd_imagine_forward: missing world-model weights");
    BD_REQUIRE(a->w_a0h && a->w_a0s && a->w_a[0] && a->w_a[1] && a->w_a[2] && a->b_a[0] && a->b_a[1] && a->b_a[2] &&
                   a->b_a[3] && a->w_a4m && a->w_a4s && a->b_a4, "bd_imagine_forward: missing actor weights");
    BD_REQUIRE(a->start_feat && a->eps_action && a->eps_prior && (a->eps_entropy || a->sv_act_stats),
               "bd_imagine_forward: missing inputs (eps_entropy may be NULL only with sv_act_stats: the entropy estimate is then "
               "the caller's bd_actor_entropy / bd_actor_entropy_rng launch)");
    BD_REQUIRE(a->feat && a->prior_std && a->entropy && a->action, "bd_imagine_forward: missing outputs");
    const ImgDims d(a->Be, a->S, a->A, a->Hd);
    // (3*16 + kWaves*16*3)*A floats of small arrays: a multiple of 16 floats, so the scratch stays 16-byte aligned
    const size_t lds = ((size_t)(3 * d.Kb_h + 2 * d.Kb_hd + d.Kb_s + d.Kb_a) * kFragFloats +
                        (size_t)(3 * 16 + kWaves * 16 * 3) * a->A + kSplitScratchFloats) * sizeof(float);
    BD_REQUIRE(lds <= (size_t)kMaxLds, "bd_imagine_forward: needs %zu B of LDS", lds);
    const size_t dyn1 = img_lds(lds, cdiv(a->N, 16));
    if (dyn1 > 64 * 1024 && allow_big_lds(imagine_fwd_kernel)) return -1;
    hipLaunchKernelGGL(imagine_fwd_kernel, dim3(cdiv(a->N, 16)), dim3(kThreads), dyn1, (hipStream_t)stream, *a);
    BD_CHECK_LAUNCH("bd_imagine_forward");
    return 0;
}

int bd_actor_entropy(const float* eps_entropy, float* act_stats, float* entropy, int Hm, int N, int A, int n_samples,
                     void* stream) {
    BD_REQUIRE(eps_entropy && act_stats && entropy && Hm > 0 && N > 0 && A > 0 && A <= kMaxA && n_samples > 0,
               "bd_actor_entropy: bad arguments");
    const int rows_pb = 64 / A;
    hipLaunchKernelGGL(actor_entropy_kernel<false>, dim3(cdiv(Hm * N, rows_pb)), dim3(64 * kEntParts), 0, (hipStream_t)stream,
                       eps_entropy, act_stats, entropy, Hm, N, A, n_samples, Rng{0, 0, 0, 0});
    BD_CHECK_LAUNCH("bd_actor_entropy");
    return 0;
}

int bd_actor_entropy_rng(unsigned long long seed, unsigned long long step, unsigned stream_id, float* act_stats, float* entropy,
                         int Hm, int N, int A, int n_samples, void* stream) {
    BD_REQUIRE(act_stats && entropy && Hm > 0 && N > 0 && A > 0 && A <= kMaxA && n_samples > 0, "bd_actor_entropy_rng: bad arguments");
    const int rows_pb = 64 / A;
    hipLaunchKernelGGL(actor_entropy_kernel<true>, dim3(cdiv(Hm * N, rows_pb)), dim3(64 * kEntParts), 0, (hipStream_t)stream,
                       nullptr, act_stats, entropy, Hm, N, A, n_samples,
                       Rng{(uint32_t)seed, (uint32_t)(seed >> 32), stream_id, (uint32_t)step});
    BD_CHECK_LAUNCH("bd_actor_entropy_rng");
    return 0;
}

int bd_imagine_forward_scan(const bd_imagine_fwd_args* a, void* stream) { return imagine_forward_scan(a, stream); }

int bd_imagine_forward(const bd_imagine_fwd_args* a, void* stream) {
    if (int rc = imagine_forward_scan(a, stream)) return rc;
    // with saved actor statistics the scan leaves (mean, std) in their slots 2, 3 and the estimate is one more launch
    if (a->sv_act_stats != nullptr && a->eps_entropy != nullptr)
        return bd_actor_entropy(a->eps_entropy, a->sv_act_stats, a->entropy, a->Hm, a->N, a->A, a->n_samples, stream);
    return 0;
}

int bd_imagine_backward(const bd_imagine_bwd_args* a, void* stream) {
    BD_REQUIRE(a && a->N > 0 && a->Hm > 0 && a->Be > 0 && a->S > 0 && a->A > 0 && a->A <= kMaxA && a->Hd > 0,
               "bd_imagine_backward: bad dims");
    BD_REQUIRE(a->wt_embed_s && a->wt_embed_a && a->wt_ir && a->wt_iz && a->wt_in && a->wt_hr && a->wt_hz && a->wt_hn &&
                   a->wt_p1 && a->wt_p2m && a->wt_p2s && a->wt_a[0] && a->wt_a[1] && a->wt_a[2] && a->wt_a4m && a->wt_a4s,
               "bd_imagine_backward: missing weights");
    BD_REQUIRE(a->start_feat && a->feat && a->prior_std && a->action && a->eps_action && a->eps_prior && a->sv_actor &&
                   a->sv_act_stats && a->sv_x && a->sv_gates && a->sv_p && a->dfeat,
               "bd_imagine_backward: missing forward tensors");
    BD_REQUIRE(a->d_actor_out, "bd_imagine_backward: missing outputs");
    const ImgDims d(a->Be, a->S, a->A, a->Hd);
    // (the backward kernels have no element-wise Gaussian head: the split-K partials alone)
    const size_t lds = ((size_t)(6 * d.Kb_h + 3 * d.Kb_hd + 2 * d.Kb_s + 2 * d.Kb_a) * kFragFloats + 16 * a->S +
                        kSplitPartialFloats) * sizeof(float);
    BD_REQUIRE(lds <= (size_t)kMaxLds, "bd_imagine_backward: needs %zu B of LDS", lds);
    const size_t dyn2 = img_lds(lds, cdiv(a->N, 16));
    if (dyn2 > 64 * 1024 && allow_big_lds(imagine_bwd_kernel)) return -1;
    hipLaunchKernelGGL(imagine_bwd_kernel, dim3(cdiv(a->N, 16)), dim3(kThreads), dyn2, (hipStream_t)stream, *a);
    BD_CHECK_LAUNCH("bd_imagine_backward");
    return 0;
}

}  // extern "C"
