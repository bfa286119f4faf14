// planner.hip -- the hot loop of the CEM planner (MPCPlanner.forward, src/planner.py:28-90):
//
//   bd_plan_rollout: one persistent launch per CEM iteration.  A workgroup owns 16 candidate action sequences and
//     walks the H planning steps with belief / state resident in LDS (MFMA fragment order): the candidate's action
//     a_t = mean_t + std_t * eps (src/planner.py:60-62) is formed on load, the prior-only RSSM step follows
//     (TransitionModel.forward with embeddings=None, src/models.py:241-256: x = ELU(W_e [s; a]), h' = GRUCell(x, h),
//     s' ~ belief_prior(h')), then the reward model (DenseModel 4 x (Linear+ELU) + Linear, src/models.py:365-408) runs
//     on [h'; s'] and the prediction is added to the candidate's return (src/planner.py:68-72).  Beliefs and states
//     never leave the CU; the only outputs are the H x rows x A actions and one return per candidate.
//     With fewer candidate tiles than CUs (one environment: 63 tiles) the step is latency bound and the reward
//     model doubles its length, so the host may ask for the features instead (returns == null, feat != null) and run
//     the reward model as ONE dense chain over all H x rows rows, which fills the chip (bd_mlp_forward).
//   bd_cem_refit: per environment, pick the `top` candidates by return (src/planner.py:74-76) and refit the action
//     belief to them: mean and biased std over the selected sequences (src/planner.py:81-87).
#include "bd_device.h"
#include "bd_host.h"

namespace bd {

struct PlanDims {
    int Kb_h, Kb_s, Kb_a, Kb_hd, Kb_f;
    __host__ __device__ PlanDims(int Be, int S, int A, int Hd)
        : Kb_h(cdiv(Be, 16)), Kb_s(cdiv(S, 16)), Kb_a(cdiv(A, 16)), Kb_hd(cdiv(Hd, 16)), Kb_f(cdiv(Be + S, 16)) {}
};

__global__ __launch_bounds__(kThreads) void plan_rollout_kernel(bd_plan_args a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const PlanDims d(a.Be, a.S, a.A, a.Hd);
    const int row0 = blockIdx.x * 16;
    const int nh = d.Kb_h * kFragFloats, nhd = d.Kb_hd * kFragFloats;
    float* h_cur = smem;
    float* h_nxt = h_cur + nh;
    float* xf = h_nxt + nh;
    float* bufA = xf + nh;
    float* bufB = bufA + nhd;
    float* sf = bufB + nhd;
    float* af = sf + d.Kb_s * kFragFloats;
    float* ff = af + d.Kb_a * kFragFloats;        // [h'; s'] as ONE K range: the reward model's first layer is packed over Be+S
    float* ret_s = ff + d.Kb_f * kFragFloats;     // [16] returns
    float* scratch = ret_s + 16;                  // split-K partials (kSplitScratchFloats), 16-byte aligned

    // every candidate of environment b starts from the same belief / state (src/planner.py:37-38)
    for (int i = threadIdx.x; i < 16 * d.Kb_h * 16; i += blockDim.x) {
        const int r = i / (d.Kb_h * 16), k = i - r * (d.Kb_h * 16), grow = row0 + r;
        h_cur[frag_idx(r, k)] = (grow < a.rows && k < a.Be) ? a.init_belief[(size_t)(grow / a.cand) * a.Be + k] : 0.f;
    }
    for (int i = threadIdx.x; i < 16 * d.Kb_s * 16; i += blockDim.x) {
        const int r = i / (d.Kb_s * 16), k = i - r * (d.Kb_s * 16), grow = row0 + r;
        sf[frag_idx(r, k)] = (grow < a.rows && k < a.S) ? a.init_state[(size_t)(grow / a.cand) * a.S + k] : 0.f;
    }
    for (int i = threadIdx.x; i < d.Kb_f * kFragFloats; i += blockDim.x) ff[i] = 0.f;   // k >= Be+S stays zero
    if (threadIdx.x < 16) ret_s[threadIdx.x] = 0.f;
    lds_barrier();

    const GruW gw{a.w_ir, a.w_iz, a.w_in, a.w_hr, a.w_hz, a.w_hn, a.b_ih, a.b_hh};
    const int B = a.rows / a.cand;
    const int F = a.Be + a.S;

    for (int t = 0; t < a.H; ++t) {
        const size_t tn = (size_t)t * a.rows;
        const int tid = bd_tid();                 // opaque: nothing thread-dependent leaves this step (bd_tid)
        const int lane = tid & 63;
        // ---- candidate actions ----
        for (int i = tid; i < 16 * d.Kb_a * 16; i += blockDim.x) {
            const int r = i / (d.Kb_a * 16), k = i - r * (d.Kb_a * 16), grow = row0 + r;
            float v = 0.f;
            if (grow < a.rows && k < a.A) {
                const size_t mi = ((size_t)t * B + grow / a.cand) * a.A + k;
                v = a.act_mean[mi] + a.act_std[mi] * a.eps_action[(tn + grow) * a.A + k];
                a.actions[(tn + grow) * a.A + k] = v;
            }
            af[frag_idx(r, k)] = v;
        }
        lds_barrier();
        // ---- x = ELU(W_e [s; a] + b_e) ----
        {
            const Seg segs[2] = {{sf, a.w_embed_s, d.Kb_s}, {af, a.w_embed_a, d.Kb_a}};
            tile_linear_seg<2>(segs, a.b_embed, a.Be, [&](int nb, floatx4 acc) {
                const int col = nb * 16 + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool ok = row0 + 4 * (lane >> 4) + r < a.rows && col < a.Be;
                    xf[acc_frag_off(nb, lane, r)] = ok ? elu(acc[r]) : 0.f;
                }
            });
        }
        lds_barrier();
        // ---- GRU ----
        gru_tile(xf, h_cur, d.Kb_h, a.Be, gw, [&](int nb, floatx4 R, floatx4 Z, floatx4 NI, floatx4 NH) {
            const int col = nb * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * (lane >> 4) + r;
                const int off = acc_frag_off(nb, lane, r);
                const float rr = sigmoidf(R[r]), zz = sigmoidf(Z[r]);
                const float nn = tanh_act(NI[r] + rr * NH[r]);
                const bool ok = row0 + row < a.rows && col < a.Be;
                const float hn = ok ? (1.f - zz) * nn + zz * h_cur[off] : 0.f;
                h_nxt[off] = hn;
                if (col < a.Be) ff[frag_idx(row, col)] = hn;
                if (a.feat && ok) a.feat[(tn + row0 + row) * F + col] = hn;
            }
        }, scratch);
        lds_barrier();
        // ---- prior: s' = mean + std * eps ----
        {
            const Seg segs[1] = {{h_nxt, a.w_p1, d.Kb_h}};
            tile_linear_seg<1>(segs, a.b_p1, a.Hd, [&](int nb, floatx4 acc) {
                const int col = nb * 16 + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool ok = row0 + 4 * (lane >> 4) + r < a.rows && col < a.Hd;
                    bufA[acc_frag_off(nb, lane, r)] = ok ? elu(acc[r]) : 0.f;
                }
            });
        }
        lds_barrier();
        {
            const Seg2 segs[1] = {{bufA, a.w_p2m, a.w_p2s, d.Kb_hd}};
            tile_dual_head_elem<1>(
                segs, a.b_p2, a.b_p2 + a.S, a.S, scratch,
                [&](int row, int col) { return row0 + row < a.rows ? a.eps_state[(tn + row0 + row) * a.S + col] : 0.f; },
                [&](int row, int col, float Mn, float Rw, float eps) {
                    const float st = row0 + row < a.rows ? Mn + (softplusf(Rw) + a.min_std) * eps : 0.f;
                    sf[frag_idx(row, col)] = st;
                    ff[frag_idx(row, a.Be + col)] = st;
                    if (a.feat && row0 + row < a.rows) a.feat[(tn + row0 + row) * F + a.Be + col] = st;
                });
        }
        lds_barrier();
        // ---- reward model on [h'; s'] (skipped when the host runs it batched over all H steps: a.returns == null) ----
        if (a.returns) {
            const Seg s0[1] = {{ff, a.w_r[0], d.Kb_f}};
            tile_linear_seg<1>(s0, a.b_r[0], a.Hd, [&](int nb, floatx4 acc) {
                const int col = nb * 16 + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) bufA[acc_frag_off(nb, lane, r)] = col < a.Hd ? elu(acc[r]) : 0.f;
            });
            lds_barrier();
            float* src = bufA;
            float* dst = bufB;
#pragma unroll
            for (int l = 1; l < 4; ++l) {
                const Seg sl[1] = {{src, a.w_r[l], d.Kb_hd}};
                tile_linear_seg<1>(sl, a.b_r[l], a.Hd, [&](int nb, floatx4 acc) {
                    const int col = nb * 16 + (lane & 15);
#pragma unroll
                    for (int r = 0; r < 4; ++r) dst[acc_frag_off(nb, lane, r)] = col < a.Hd ? elu(acc[r]) : 0.f;
                });
                lds_barrier();
                float* tmp = src; src = dst; dst = tmp;
            }
            const Seg so[1] = {{src, a.w_r[4], d.Kb_hd}};
            tile_linear_seg<1>(so, a.b_r[4], 1, [&](int nb, floatx4 acc) {
                if (nb == 0 && (lane & 15) == 0) {       // column 0: one lane per group of four rows
#pragma unroll
                    for (int r = 0; r < 4; ++r) ret_s[4 * (lane >> 4) + r] += acc[r];      // sum over the horizon (:72)
                }
            }, scratch);
            lds_barrier();
        }
        float* tmp = h_cur; h_cur = h_nxt; h_nxt = tmp;
    }
    if (a.returns && threadIdx.x < 16 && row0 + threadIdx.x < a.rows) a.returns[row0 + threadIdx.x] = ret_s[threadIdx.x];
}

// ---- CEM refit ---------------------------------------------------------------------------------------------
// One workgroup per environment.  Candidates are ordered by (return, lower index first; a NaN return ranks first, as
// torch.topk orders it) with a bitonic sort of 64-bit keys in LDS -- (order-preserving image of the float) << 32 |
// ~index -- padded to a power of two with keys below every real one; the first `top` entries are the selection, and
// every (t, a) pair is reduced over them by one wave.
__device__ __forceinline__ unsigned long long refit_key(float x, int i) {
    unsigned u = __float_as_uint(x);
    if (x != x) u = 0xFFFFFFFFu;
    else u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((unsigned long long)u << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
}

__global__ __launch_bounds__(1024) void cem_refit_kernel(const float* __restrict__ returns, int ret_steps,
                                                         const float* __restrict__ actions, int H, int B, int cand, int top,
                                                         int A, int n2, float* __restrict__ mean, float* __restrict__ stdev) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];   // [n2]
    __shared__ int idx[1024];                                                    // selected candidates (top <= 1024)
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < n2; i += blockDim.x) {
        unsigned long long key = 0ull;
        if (i < cand) {                                 // return = sum of the per-step rewards (src/planner.py:72)
            float r = 0.f;
            for (int t = 0; t < ret_steps; ++t) r += returns[((size_t)t * B + b) * cand + i];
            key = refit_key(r, i);
        }
        keys[i] = key;
    }
    __syncthreads();
    for (int k = 2; k <= n2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < n2; i += blockDim.x) {
                const int q = i ^ j;
                if (q > i) {
                    const unsigned long long x = keys[i], y = keys[q];
                    const bool desc = (i & k) == 0;
                    if (desc ? (x < y) : (x > y)) {
                        keys[i] = y;
                        keys[q] = x;
                    }
                }
            }
            __syncthreads();
        }
    }
    for (int j = threadIdx.x; j < top; j += blockDim.x) idx[j] = (int)(0xFFFFFFFFu - (unsigned)(keys[j] & 0xFFFFFFFFull));
    __syncthreads();
    // one wave per (t, a) pair, lanes over the selected candidates: two load rounds per pair instead of 2 * top
    // dependent ones
    const size_t rows = (size_t)B * cand;
    const float inv = 1.f / (float)top;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    for (int p = wave; p < H * A; p += nwaves) {
        const int t = p / A, k = p - t * A;
        const float* base = actions + ((size_t)t * rows + (size_t)b * cand) * A + k;
        float s = 0.f;
        for (int j = lane; j < top; j += 64) s += base[(size_t)idx[j] * A];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        const float m = s * inv;
        float v = 0.f;
        for (int j = lane; j < top; j += 64) {
            const float dlt = base[(size_t)idx[j] * A] - m;
            v += dlt * dlt;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) {
            const size_t o = ((size_t)t * B + b) * A + k;
            mean[o] = m;
            stdev[o] = sqrtf(v * inv);                    // std(unbiased=False), src/planner.py:87
        }
    }
}

}  // namespace bd

extern "C" {
using namespace bd;

int bd_plan_rollout(const bd_plan_args* a, void* stream) {
    BD_REQUIRE(a && a->rows > 0 && a->H > 0 && a->cand > 0 && a->rows % a->cand == 0 && a->Be > 0 && a->S > 0 && a->A > 0 &&
                   a->Hd > 0, "bd_plan_rollout: bad dims");
    BD_REQUIRE(a->S <= kHeadMaxN, "bd_plan_rollout: state_size %d > %d", a->S, kHeadMaxN);
    BD_REQUIRE(a->w_embed_s && a->w_embed_a && a->b_embed && a->w_ir && a->w_iz && a->w_in && a->w_hr && a->w_hz && a->w_hn &&
                   a->b_ih && a->b_hh && a->w_p1 && a->b_p1 && a->w_p2m && a->w_p2s && a->b_p2,
               "bd_plan_rollout: missing transition weights");
    for (int l = 0; l < 5; ++l) BD_REQUIRE(a->w_r[l] && a->b_r[l], "bd_plan_rollout: missing reward weights (layer %d)", l);
    BD_REQUIRE(a->init_belief && a->init_state && a->act_mean && a->act_std && a->eps_action && a->eps_state,
               "bd_plan_rollout: missing inputs");
    BD_REQUIRE(a->actions && (a->returns || a->feat), "bd_plan_rollout: missing outputs");
    const PlanDims d(a->Be, a->S, a->A, a->Hd);
    const size_t lds = ((size_t)(3 * d.Kb_h + 2 * d.Kb_hd + d.Kb_s + d.Kb_a + d.Kb_f) * kFragFloats + 16 + kSplitScratchFloats) *
                       sizeof(float);
    BD_REQUIRE(lds <= (size_t)kMaxLds, "bd_plan_rollout: needs %zu B of LDS", lds);
    if (lds > 64 * 1024 && allow_big_lds(plan_rollout_kernel)) return -1;
    hipLaunchKernelGGL(plan_rollout_kernel, dim3(cdiv(a->rows, 16)), dim3(kThreads), lds, (hipStream_t)stream, *a);
    BD_CHECK_LAUNCH("bd_plan_rollout");
    return 0;
}

int bd_cem_refit(const float* returns, int ret_steps, const float* actions, int H, int B, int cand, int top, int A,
                 float* mean, float* stdev, void* stream) {
    BD_REQUIRE(returns && ret_steps > 0 && actions && mean && stdev && H > 0 && B > 0 && cand > 0 && A > 0,
               "bd_cem_refit: bad arguments");
    BD_REQUIRE(top > 0 && top <= cand, "bd_cem_refit: top_candidates %d must be in 1..%d", top, cand);
    BD_REQUIRE(top <= 1024 && cand <= 4096, "bd_cem_refit: at most 4096 candidates / 1024 top candidates (got %d / %d)", cand,
               top);
    int n2 = 2;
    while (n2 < cand) n2 <<= 1;
    const size_t lds = (size_t)n2 * sizeof(unsigned long long);
    hipLaunchKernelGGL(cem_refit_kernel, dim3(B), dim3(1024), lds, (hipStream_t)stream, returns, ret_steps, actions, H, B, cand,
                       top, A, n2, mean, stdev);
    BD_CHECK_LAUNCH("bd_cem_refit");
    return 0;
}

}  // extern "C"
