// reduce.hip -- losses, reductions, Gaussian belief tail, lambda-return scan, clip+Adam, polyak.
// All HBM-bound elementwise / reduction work: coalesced grid-stride loops, wave-shuffle reductions,
// fp64 partials summed in fixed order (deterministic).
#include "bd_device.h"
#include "bd_host.h"

namespace bd {

constexpr int kRedBlocks = 1024;
constexpr float kHalfLog2Pi = 0.91893853320467274178f;

static inline int red_blocks(size_t n) {
    size_t b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > kRedBlocks ? kRedBlocks : b));
}

__global__ __launch_bounds__(256) void final_sum_kernel(const double* __restrict__ partials, int n,
                                                        float* __restrict__ scalars, int slot) {
    __shared__ double red[kWaves];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += partials[i];
    s = block_sum_d(s, red);
    if (threadIdx.x == 0) scalars[slot] = (float)s;
}

template <int MODE>   // 0: sum, 1: sum of squares
__global__ __launch_bounds__(256) void sum_kernel(const float* __restrict__ x, size_t n, double* __restrict__ partials) {
    __shared__ double red[kWaves];
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = x[i];
        s += MODE ? (double)v * v : (double)v;
    }
    s = block_sum_d(s, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void normal_nll_kernel(const float* __restrict__ pred, int ldp,
                                                         const float* __restrict__ target, int ldt, int rows, int D,
                                                         float grad_scale, float* __restrict__ dpred, int ldd,
                                                         double* __restrict__ partials) {
    __shared__ double red[kWaves];
    double s = 0.0;
    const size_t n = (size_t)rows * D;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / D, c = i - r * D;
        const float d = pred[r * ldp + c] - target[r * ldt + c];
        s += (double)(0.5f * d * d + kHalfLog2Pi);
        if (dpred) dpred[r * ldd + c] = d * grad_scale;
    }
    s = block_sum_d(s, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

// -log Bernoulli(logits=x).prob(t) = binary_cross_entropy_with_logits(x, t) = max(x, 0) - x t + log1p(exp(-|x|))
// (Dreamer._discount_loss, src/dreamer.py:239-251); d/dx = sigmoid(x) - t
__global__ __launch_bounds__(256) void bernoulli_nll_kernel(const float* __restrict__ logits, const float* __restrict__ target,
                                                            size_t n, float grad_scale, float* __restrict__ dlogits,
                                                            double* __restrict__ partials) {
    __shared__ double red[kWaves];
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float x = logits[i], t = target[i];
        s += (double)(fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x))));
        if (dlogits) dlogits[i] = (1.f / (1.f + expf(-x)) - t) * grad_scale;
    }
    s = block_sum_d(s, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

// torch.distributions.kl._kl_normal_normal
__device__ __forceinline__ float kl_elem(float qm, float qs, float pm, float ps) {
    const float ratio = qs / ps;
    const float vr = ratio * ratio;
    const float t = (qm - pm) / ps;
    return 0.5f * (vr + t * t - 1.f - logf(vr));
}

__global__ __launch_bounds__(256) void kl_fwd_kernel(const float* __restrict__ qm, const float* __restrict__ qs,
                                                     const float* __restrict__ pm, const float* __restrict__ ps,
                                                     int rows, int S, float free_nats, int sum_form,
                                                     double* __restrict__ partials) {
    __shared__ double red[kWaves];
    double s = 0.0;
    if (!sum_form) {
        const size_t n = (size_t)rows * S;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
            s += (double)kl_elem(qm[i], qs[i], pm[i], ps[i]);
    } else {
        for (size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x; r < (size_t)rows;
             r += (size_t)gridDim.x * blockDim.x) {
            float rs = 0.f;
            for (int c = 0; c < S; ++c) {
                const size_t i = r * S + c;
                rs += kl_elem(qm[i], qs[i], pm[i], ps[i]);
            }
            s += (double)fmaxf(rs, free_nats);
        }
    }
    s = block_sum_d(s, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

// gradient factor of torch.max(x, free_nats) w.r.t. x (ties split evenly, as torch.maximum's backward)
__device__ __forceinline__ float max_grad(float x, float fn) { return x > fn ? 1.f : (x == fn ? 0.5f : 0.f); }

__global__ __launch_bounds__(256) void kl_bwd_kernel(const float* __restrict__ qm, const float* __restrict__ qs,
                                                     const float* __restrict__ pm, const float* __restrict__ ps,
                                                     int rows, int S, float free_nats, float kl_balance, float weight,
                                                     float inv_count, const float* __restrict__ scalars, int slot,
                                                     float* __restrict__ dqm, float* __restrict__ dqs,
                                                     float* __restrict__ dpm, float* __restrict__ dps) {
    const size_t n = (size_t)rows * S;
    const bool sum_form = kl_balance == -1.f;
    float fq, fp;
    if (!sum_form) {
        const float f = max_grad(scalars[slot] * inv_count, free_nats) * weight * inv_count;
        fp = kl_balance * f;            // lhs: KL(sg(post) || prior) -> prior parameters
        fq = (1.f - kl_balance) * f;    // rhs: KL(post || sg(prior)) -> posterior parameters
    } else {
        fq = fp = weight / (float)rows;
    }
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float gq = fq, gp = fp;
        if (sum_form) {
            const size_t r = i / S;
            float rs = 0.f;
            for (int c = 0; c < S; ++c) {
                const size_t j = r * S + c;
                rs += kl_elem(qm[j], qs[j], pm[j], ps[j]);
            }
            const float f = max_grad(rs, free_nats);
            gq *= f;
            gp *= f;
        }
        const float a = qm[i], b = qs[i], c = pm[i], d = ps[i];
        const float inv_d = 1.f / d, inv_d2 = inv_d * inv_d, diff = a - c;
        dqm[i] = gq * diff * inv_d2;
        dqs[i] = gq * (b * inv_d2 - 1.f / b);
        dpm[i] = -gp * diff * inv_d2;
        dps[i] = gp * (inv_d - (b * b + diff * diff) * inv_d2 * inv_d);
    }
}

// ---- Gaussian belief tail ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gauss_head_fwd_kernel(const float* __restrict__ out, const float* __restrict__ eps,
                                                             int M, int S, float min_std, float* __restrict__ mean,
                                                             float* __restrict__ std, float* __restrict__ state) {
    const size_t n = (size_t)M * S;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / S, c = i - r * S;
        const float m = out[r * 2 * S + c];
        const float sd = softplusf(out[r * 2 * S + S + c]) + min_std;
        if (mean) mean[i] = m;
        if (std) std[i] = sd;
        if (state) state[i] = m + sd * eps[i];
    }
}

__global__ __launch_bounds__(256) void gauss_head_bwd_kernel(const float* __restrict__ out, const float* __restrict__ eps,
                                                             const float* __restrict__ dstate,
                                                             const float* __restrict__ dmean,
                                                             const float* __restrict__ dstd, int M, int S,
                                                             float* __restrict__ dout) {
    const size_t n = (size_t)M * S;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / S, c = i - r * S;
        const float ds = dstate ? dstate[i] : 0.f;
        const float gm = ds + (dmean ? dmean[i] : 0.f);
        const float gs = (dstate ? ds * eps[i] : 0.f) + (dstd ? dstd[i] : 0.f);
        dout[r * 2 * S + c] = gm;
        dout[r * 2 * S + S + c] = gs * sigmoidf(out[r * 2 * S + S + c]);   // d softplus = sigmoid
    }
}

// ---- lambda-return scan (one lane per trajectory, coalesced over trajectories) ---------------------------
__global__ __launch_bounds__(256) void lambda_fwd_kernel(const float* __restrict__ reward, const float* __restrict__ value,
                                                         int Hm, int N, float disc, float lam, float* __restrict__ ret) {
    const float one_minus = 1.f - lam, dl = disc * lam;
    for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) {
        const float boot = value[(size_t)(Hm - 1) * N + n];
        float last = boot;
        for (int t = Hm - 1; t >= 0; --t) {
            const float next = (t == Hm - 1) ? boot : value[(size_t)(t + 1) * N + n];
            const float inp = reward[(size_t)t * N + n] + disc * next * one_minus;
            last = inp + dl * last;
            ret[(size_t)t * N + n] = last;
        }
    }
}

__global__ __launch_bounds__(256) void lambda_bwd_kernel(const float* __restrict__ dret, float dconst, int Hm, int N,
                                                         float disc, float lam, float* __restrict__ dreward,
                                                         float* __restrict__ dvalue) {
    const float c = disc * lam, w = disc * (1.f - lam);
    for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) {
        float G = 0.f;
        for (int t = 0; t < Hm; ++t) {
            G = (dret ? dret[(size_t)t * N + n] : dconst) + c * G;   // total gradient reaching returns[t]
            dreward[(size_t)t * N + n] = G;
            // inputs[t] uses next_values[t] = value[t+1] (t < Hm-1) or the bootstrap value[Hm-1]
            if (t == 0 && Hm > 1) dvalue[n] = 0.f;                   // value[0] is never read
            if (t < Hm - 1) {
                dvalue[(size_t)(t + 1) * N + n] = G * w;
            } else {
                // bootstrap enters twice: as next_values[Hm-1] and as the scan's initial `last`
                const float prev = (Hm > 1) ? dvalue[(size_t)t * N + n] : 0.f;
                dvalue[(size_t)t * N + n] = prev + G * w + c * G;
            }
        }
    }
}

// ---- clip_grad_norm_ + Adam ------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, size_t n, float beta1, float beta2, float eps,
                                                   float wd, float step_size, float inv_sqrt_bc2, float max_norm,
                                                   const float* __restrict__ scalars, int sqnorm_slot) {
    const float total_norm = sqrtf(scalars[sqnorm_slot]);
    const float coef = fminf(max_norm / (total_norm + 1e-6f), 1.f);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float gc = g[i] * coef;
        g[i] = gc;
        const float pi = p[i];
        const float gw = gc + wd * pi;
        const float mi = m[i] + (1.f - beta1) * (gw - m[i]);           // lerp_
        const float vi = v[i] * beta2 + (1.f - beta2) * gw * gw;       // mul_().addcmul_()
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
        p[i] = pi - step_size * (mi / denom);
    }
}

__global__ __launch_bounds__(256) void polyak_kernel(float* __restrict__ t, const float* __restrict__ s, size_t n, float w) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        t[i] = s[i] * w + t[i] * (1.f - w);
}

// ---- Categorical latents (CategoricalBeliefModel, src/models.py:101-117; KL branch src/dreamer.py:102-106,131-144) ----
// rows x D groups of C classes (reference: 32 x 32).  A 32-lane half wave owns one group; a lane holds classes
// c = l, l+32, ... (C <= 128).  Reductions are xor-shuffles with offsets < 32, which stay inside the half wave.
constexpr int kCatPer = 4;     // classes per lane

__device__ __forceinline__ float half_max(float v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// normalised log-probabilities of one group, as Categorical(logits=...) keeps them: x - logsumexp(x)
// (torch/distributions/categorical.py); lp[i] is class l + 32 i (-inf beyond C)
__device__ __forceinline__ void group_log_softmax(const float* __restrict__ x, int C, int l, float (&lp)[kCatPer]) {
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < kCatPer; ++i) {
        const int c = l + 32 * i;
        lp[i] = c < C ? x[c] : -INFINITY;
        m = fmaxf(m, lp[i]);
    }
    m = half_max(m);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < kCatPer; ++i) s += (l + 32 * i < C) ? expf(lp[i] - m) : 0.f;
    const float lse = m + logf(half_sum(s));
#pragma unroll
    for (int i = 0; i < kCatPer; ++i) lp[i] -= lse;
}

// state = one_hot(argmax(probs / q)) with q ~ Exp(1): torch.multinomial's single-draw path (ATen
// native/Distributions.cpp: `q = empty_like(probs).exponential_(1); argmax(probs / q)`), reached from
// OneHotCategoricalStraightThrough.rsample(); the straight-through term probs - probs.detach() is exactly zero.
__global__ __launch_bounds__(256) void cat_head_fwd_kernel(const float* __restrict__ logits, const float* __restrict__ q,
                                                           size_t groups, int C, float* __restrict__ state,
                                                           float* __restrict__ probs) {
    const int l = threadIdx.x & 31;
    for (size_t g = (size_t)blockIdx.x * 8 + (threadIdx.x >> 5); g < groups; g += (size_t)gridDim.x * 8) {
        const size_t base = g * C;
        float lp[kCatPer];
        group_log_softmax(logits + base, C, l, lp);
        // probs = softmax(normalised logits) (logits_to_probs)
        float pr[kCatPer], m = -INFINITY, s = 0.f;
#pragma unroll
        for (int i = 0; i < kCatPer; ++i) m = fmaxf(m, lp[i]);
        m = half_max(m);
#pragma unroll
        for (int i = 0; i < kCatPer; ++i) {
            pr[i] = (l + 32 * i < C) ? expf(lp[i] - m) : 0.f;
            s += pr[i];
        }
        s = half_sum(s);
        float best = -INFINITY;
        int arg = 0x7fffffff;
#pragma unroll
        for (int i = 0; i < kCatPer; ++i) {
            const int c = l + 32 * i;
            if (c < C) {
                pr[i] /= s;
                probs[base + c] = pr[i];
                const float r = pr[i] / q[base + c];
                if (r > best) { best = r; arg = c; }       // first maximum wins (argmax)
            }
        }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o, 64);
            const int oa = __shfl_xor(arg, o, 64);
            if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
        }
#pragma unroll
        for (int i = 0; i < kCatPer; ++i) {
            const int c = l + 32 * i;
            if (c < C) state[base + c] = (c == arg) ? 1.f : 0.f;
        }
    }
}

// straight-through: d state / d probs = I, probs = softmax(logits): dlogits = p * (g - sum_c p g)
__global__ __launch_bounds__(256) void cat_head_bwd_kernel(const float* __restrict__ dstate, const float* __restrict__ probs,
                                                           size_t groups, int C, float* __restrict__ dlogits) {
    const int l = threadIdx.x & 31;
    for (size_t g = (size_t)blockIdx.x * 8 + (threadIdx.x >> 5); g < groups; g += (size_t)gridDim.x * 8) {
        const size_t base = g * C;
        float pr[kCatPer], gr[kCatPer], dot = 0.f;
#pragma unroll
        for (int i = 0; i < kCatPer; ++i) {
            const int c = l + 32 * i;
            pr[i] = c < C ? probs[base + c] : 0.f;
            gr[i] = c < C ? dstate[base + c] : 0.f;
            dot += pr[i] * gr[i];
        }
        dot = half_sum(dot);
#pragma unroll
        for (int i = 0; i < kCatPer; ++i) {
            const int c = l + 32 * i;
            if (c < C) dlogits[base + c] = pr[i] * (gr[i] - dot);
        }
    }
}

// KL(q || p) of one group from normalised log-probabilities (torch.distributions.kl._kl_categorical_categorical:
// t = q.probs * (q.logits - p.logits); t[p.probs == 0] = inf; t[q.probs == 0] = 0)
__device__ __forceinline__ float group_kl(const float (&lq)[kCatPer], const float (&lp)[kCatPer], int C, int l) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < kCatPer; ++i) {
        if (l + 32 * i < C) {
            const float qv = expf(lq[i]), pv = expf(lp[i]);
            float t = qv * (lq[i] - lp[i]);
            if (pv == 0.f) t = INFINITY;
            if (qv == 0.f) t = 0.f;
            s += t;
        }
    }
    return half_sum(s);
}

// unit of work = one group (balanced form: mean over all rows x D groups) or one row (sum form: max(sum_D, free_nats))
__global__ __launch_bounds__(256) void cat_kl_fwd_kernel(const float* __restrict__ ql, const float* __restrict__ pl, int rows,
                                                         int D, int C, float free_nats, int sum_form,
                                                         double* __restrict__ partials) {
    __shared__ double red[kWaves > 4 ? kWaves : 4];
    const int l = threadIdx.x & 31;
    const size_t units = sum_form ? (size_t)rows : (size_t)rows * D;
    const int per = sum_form ? D : 1;
    double acc = 0.0;
    for (size_t u = (size_t)blockIdx.x * 8 + (threadIdx.x >> 5); u < units; u += (size_t)gridDim.x * 8) {
        float rs = 0.f;
        for (int d = 0; d < per; ++d) {
            const size_t base = (u * per + d) * C;
            float lq[kCatPer], lp[kCatPer];
            group_log_softmax(ql + base, C, l, lq);
            group_log_softmax(pl + base, C, l, lp);
            rs += group_kl(lq, lp, C, l);
        }
        if (l == 0) acc += (double)(sum_form ? fmaxf(rs, free_nats) : rs);
    }
    acc = block_sum_d(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}

// balanced: lhs = KL(sg(post) || prior) -> d prior = p - q;  rhs = KL(post || sg(prior)) -> d post = q (lq - lp - KL)
__global__ __launch_bounds__(256) void cat_kl_bwd_kernel(const float* __restrict__ ql, const float* __restrict__ pl, int rows,
                                                         int D, int C, float free_nats, float kl_balance, float weight,
                                                         float inv_count, const float* __restrict__ scalars, int slot,
                                                         float* __restrict__ dql, float* __restrict__ dpl) {
    const int l = threadIdx.x & 31;
    const bool sum_form = kl_balance == -1.f;
    const size_t units = sum_form ? (size_t)rows : (size_t)rows * D;
    const int per = sum_form ? D : 1;
    float fq, fp;
    if (!sum_form) {
        const float f = max_grad(scalars[slot] * inv_count, free_nats) * weight * inv_count;
        fp = kl_balance * f;
        fq = (1.f - kl_balance) * f;
    } else {
        fq = fp = weight / (float)rows;
    }
    for (size_t u = (size_t)blockIdx.x * 8 + (threadIdx.x >> 5); u < units; u += (size_t)gridDim.x * 8) {
        float f = 1.f;
        if (sum_form) {
            float rs = 0.f;
            for (int d = 0; d < per; ++d) {
                const size_t base = (u * per + d) * C;
                float lq[kCatPer], lp[kCatPer];
                group_log_softmax(ql + base, C, l, lq);
                group_log_softmax(pl + base, C, l, lp);
                rs += group_kl(lq, lp, C, l);
            }
            f = max_grad(rs, free_nats);
        }
        for (int d = 0; d < per; ++d) {
            const size_t base = (u * per + d) * C;
            float lq[kCatPer], lp[kCatPer];
            group_log_softmax(ql + base, C, l, lq);
            group_log_softmax(pl + base, C, l, lp);
            const float kl = group_kl(lq, lp, C, l);
#pragma unroll
            for (int i = 0; i < kCatPer; ++i) {
                const int c = l + 32 * i;
                if (c < C) {
                    const float qv = expf(lq[i]), pv = expf(lp[i]);
                    dql[base + c] = f * fq * (qv == 0.f ? 0.f : qv * (lq[i] - lp[i] - kl));
                    dpl[base + c] = f * fp * (pv - qv);
                }
            }
        }
    }
}

static int finish(double* partials, int nb, float* scalars, int slot, hipStream_t s, const char* name) {
    hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(256), 0, s, partials, nb, scalars, slot);
    BD_CHECK_LAUNCH(name);
    return 0;
}

}  // namespace bd

extern "C" {
using namespace bd;

size_t bd_reduce_ws_floats(void) { return 2 * (size_t)kRedBlocks; }

int bd_sum(const float* x, size_t n, float* scalars, int slot, float* ws, void* stream) {
    BD_REQUIRE(x && scalars && ws && n > 0, "bd_sum: bad arguments");
    const int nb = red_blocks(n);
    hipLaunchKernelGGL(sum_kernel<0>, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, n, (double*)ws);
    BD_CHECK_LAUNCH("bd_sum");
    return finish((double*)ws, nb, scalars, slot, (hipStream_t)stream, "bd_sum(final)");
}

int bd_sumsq(const float* x, size_t n, float* scalars, int slot, float* ws, void* stream) {
    BD_REQUIRE(x && scalars && ws && n > 0, "bd_sumsq: bad arguments");
    const int nb = red_blocks(n);
    hipLaunchKernelGGL(sum_kernel<1>, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, n, (double*)ws);
    BD_CHECK_LAUNCH("bd_sumsq");
    return finish((double*)ws, nb, scalars, slot, (hipStream_t)stream, "bd_sumsq(final)");
}

int bd_normal_nll(const float* pred, int ldp, const float* target, int ldt, int rows, int D, float grad_scale,
                  float* dpred, int ldd, float* scalars, int slot, float* ws, void* stream) {
    BD_REQUIRE(pred && target && scalars && ws && rows > 0 && D > 0, "bd_normal_nll: bad arguments");
    BD_REQUIRE(ldp >= D && ldt >= D && (!dpred || ldd >= D), "bd_normal_nll: leading dimension too small");
    const int nb = red_blocks((size_t)rows * D);
    hipLaunchKernelGGL(normal_nll_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, pred, ldp, target, ldt, rows, D,
                       grad_scale, dpred, ldd, (double*)ws);
    BD_CHECK_LAUNCH("bd_normal_nll");
    return finish((double*)ws, nb, scalars, slot, (hipStream_t)stream, "bd_normal_nll(final)");
}

int bd_bernoulli_nll(const float* logits, const float* target, size_t n, float grad_scale, float* dlogits, float* scalars,
                     int slot, float* ws, void* stream) {
    BD_REQUIRE(logits && target && scalars && ws && n > 0, "bd_bernoulli_nll: bad arguments");
    const int nb = red_blocks(n);
    hipLaunchKernelGGL(bernoulli_nll_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, logits, target, n, grad_scale,
                       dlogits, (double*)ws);
    BD_CHECK_LAUNCH("bd_bernoulli_nll");
    return finish((double*)ws, nb, scalars, slot, (hipStream_t)stream, "bd_bernoulli_nll(final)");
}

int bd_kl_forward(const float* qm, const float* qs, const float* pm, const float* ps, int rows, int S, float free_nats,
                  int sum_form, float* scalars, int slot, float* ws, void* stream) {
    BD_REQUIRE(qm && qs && pm && ps && scalars && ws && rows > 0 && S > 0, "bd_kl_forward: bad arguments");
    const int nb = red_blocks(sum_form ? (size_t)rows : (size_t)rows * S);
    hipLaunchKernelGGL(kl_fwd_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, qm, qs, pm, ps, rows, S, free_nats,
                       sum_form, (double*)ws);
    BD_CHECK_LAUNCH("bd_kl_forward");
    return finish((double*)ws, nb, scalars, slot, (hipStream_t)stream, "bd_kl_forward(final)");
}

int bd_kl_backward(const float* qm, const float* qs, const float* pm, const float* ps, int rows, int S, float free_nats,
                   float kl_balance, float weight, float inv_count, const float* scalars, int slot, float* dqm,
                   float* dqs, float* dpm, float* dps, void* stream) {
    BD_REQUIRE(qm && qs && pm && ps && scalars && dqm && dqs && dpm && dps && rows > 0 && S > 0,
               "bd_kl_backward: bad arguments");
    hipLaunchKernelGGL(kl_bwd_kernel, dim3(red_blocks((size_t)rows * S)), dim3(256), 0, (hipStream_t)stream, qm, qs, pm,
                       ps, rows, S, free_nats, kl_balance, weight, inv_count, scalars, slot, dqm, dqs, dpm, dps);
    BD_CHECK_LAUNCH("bd_kl_backward");
    return 0;
}

int bd_gauss_head_forward(const float* out, const float* eps, int M, int S, float min_std, float* mean, float* std,
                          float* state, void* stream) {
    BD_REQUIRE(out && M > 0 && S > 0 && (!state || eps), "bd_gauss_head_forward: bad arguments");
    hipLaunchKernelGGL(gauss_head_fwd_kernel, dim3(red_blocks((size_t)M * S)), dim3(256), 0, (hipStream_t)stream, out, eps,
                       M, S, min_std, mean, std, state);
    BD_CHECK_LAUNCH("bd_gauss_head_forward");
    return 0;
}

int bd_gauss_head_backward(const float* out, const float* eps, const float* dstate, const float* dmean,
                           const float* dstd, int M, int S, float* dout, void* stream) {
    BD_REQUIRE(out && dout && M > 0 && S > 0 && (!dstate || eps), "bd_gauss_head_backward: bad arguments");
    hipLaunchKernelGGL(gauss_head_bwd_kernel, dim3(red_blocks((size_t)M * S)), dim3(256), 0, (hipStream_t)stream, out, eps,
                       dstate, dmean, dstd, M, S, dout);
    BD_CHECK_LAUNCH("bd_gauss_head_backward");
    return 0;
}

int bd_lambda_return_forward(const float* reward, const float* value, int Hm, int N, float discount, float lambda_,
                             float* returns, void* stream) {
    BD_REQUIRE(reward && value && returns && Hm > 0 && N > 0, "bd_lambda_return_forward: bad arguments");
    hipLaunchKernelGGL(lambda_fwd_kernel, dim3(red_blocks(N)), dim3(256), 0, (hipStream_t)stream, reward, value, Hm, N,
                       discount, lambda_, returns);
    BD_CHECK_LAUNCH("bd_lambda_return_forward");
    return 0;
}

int bd_lambda_return_backward(const float* dreturns, float dret_const, int Hm, int N, float discount, float lambda_,
                              float* dreward, float* dvalue, void* stream) {
    BD_REQUIRE(dreward && dvalue && Hm > 0 && N > 0, "bd_lambda_return_backward: bad arguments");
    hipLaunchKernelGGL(lambda_bwd_kernel, dim3(red_blocks(N)), dim3(256), 0, (hipStream_t)stream, dreturns, dret_const, Hm,
                       N, discount, lambda_, dreward, dvalue);
    BD_CHECK_LAUNCH("bd_lambda_return_backward");
    return 0;
}

int bd_adam_step(float* p, float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps,
                 float weight_decay, int step, float max_norm, const float* scalars, int sqnorm_slot, void* stream) {
    BD_REQUIRE(p && g && m && v && scalars && n > 0 && step >= 1, "bd_adam_step: bad arguments");
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    hipLaunchKernelGGL(adam_kernel, dim3(red_blocks(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, beta1, beta2,
                       eps, weight_decay, (float)(lr / bc1), (float)(1.0 / sqrt(bc2)), max_norm, scalars, sqnorm_slot);
    BD_CHECK_LAUNCH("bd_adam_step");
    return 0;
}

int bd_polyak(float* target, const float* src, size_t n, float weight, void* stream) {
    BD_REQUIRE(target && src && n > 0, "bd_polyak: bad arguments");
    hipLaunchKernelGGL(polyak_kernel, dim3(red_blocks(n)), dim3(256), 0, (hipStream_t)stream, target, src, n, weight);
    BD_CHECK_LAUNCH("bd_polyak");
    return 0;
}

static inline int cat_blocks(size_t units) {
    size_t b = (units + 7) / 8;
    return (int)(b < 1 ? 1 : (b > (size_t)kRedBlocks ? (size_t)kRedBlocks : b));
}

int bd_categorical_head_forward(const float* logits, const float* q_noise, int rows, int D, int C, float* state,
                                float* probs, void* stream) {
    BD_REQUIRE(logits && q_noise && state && probs && rows > 0 && D > 0, "bd_categorical_head_forward: bad arguments");
    BD_REQUIRE(C > 0 && C <= 32 * kCatPer, "bd_categorical_head_forward: %d classes (max %d)", C, 32 * kCatPer);
    const size_t groups = (size_t)rows * D;
    hipLaunchKernelGGL(cat_head_fwd_kernel, dim3(cat_blocks(groups)), dim3(256), 0, (hipStream_t)stream, logits, q_noise,
                       groups, C, state, probs);
    BD_CHECK_LAUNCH("bd_categorical_head_forward");
    return 0;
}

int bd_categorical_head_backward(const float* dstate, const float* probs, int rows, int D, int C, float* dlogits,
                                 void* stream) {
    BD_REQUIRE(dstate && probs && dlogits && rows > 0 && D > 0, "bd_categorical_head_backward: bad arguments");
    BD_REQUIRE(C > 0 && C <= 32 * kCatPer, "bd_categorical_head_backward: %d classes (max %d)", C, 32 * kCatPer);
    const size_t groups = (size_t)rows * D;
    hipLaunchKernelGGL(cat_head_bwd_kernel, dim3(cat_blocks(groups)), dim3(256), 0, (hipStream_t)stream, dstate, probs,
                       groups, C, dlogits);
    BD_CHECK_LAUNCH("bd_categorical_head_backward");
    return 0;
}

int bd_kl_categorical_forward(const float* post_logits, const float* prior_logits, int rows, int D, int C, float free_nats,
                              int sum_form, float* scalars, int slot, float* ws, void* stream) {
    BD_REQUIRE(post_logits && prior_logits && scalars && ws && rows > 0 && D > 0, "bd_kl_categorical_forward: bad arguments");
    BD_REQUIRE(C > 0 && C <= 32 * kCatPer, "bd_kl_categorical_forward: %d classes (max %d)", C, 32 * kCatPer);
    const int nb = cat_blocks(sum_form ? (size_t)rows : (size_t)rows * D);
    hipLaunchKernelGGL(cat_kl_fwd_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, post_logits, prior_logits, rows, D, C,
                       free_nats, sum_form, (double*)ws);
    BD_CHECK_LAUNCH("bd_kl_categorical_forward");
    return finish((double*)ws, nb, scalars, slot, (hipStream_t)stream, "bd_kl_categorical_forward(final)");
}

int bd_kl_categorical_backward(const float* post_logits, const float* prior_logits, int rows, int D, int C, float free_nats,
                               float kl_balance, float weight, float inv_count, const float* scalars, int slot,
                               float* dpost, float* dprior, void* stream) {
    BD_REQUIRE(post_logits && prior_logits && scalars && dpost && dprior && rows > 0 && D > 0,
               "bd_kl_categorical_backward: bad arguments");
    BD_REQUIRE(C > 0 && C <= 32 * kCatPer, "bd_kl_categorical_backward: %d classes (max %d)", C, 32 * kCatPer);
    const bool sum_form = kl_balance == -1.f;
    hipLaunchKernelGGL(cat_kl_bwd_kernel, dim3(cat_blocks(sum_form ? (size_t)rows : (size_t)rows * D)), dim3(256), 0,
                       (hipStream_t)stream, post_logits, prior_logits, rows, D, C, free_nats, kl_balance, weight, inv_count,
                       scalars, slot, dpost, dprior);
    BD_CHECK_LAUNCH("bd_kl_categorical_backward");
    return 0;
}

}  // extern "C"
