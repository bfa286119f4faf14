/*
 * bigdreamer_hip.h -- C ABI of the MI355X-native Dreamer world-model training step.
 *
 * The reference (jgsimard/big-dreamer) has no FFI / plugin interface: its hot path is a chain of
 * ATen ops issued from Python.  This library is what a binding for that path
 * binds instead; each entry point names the reference code it replaces.  The Python host in
 * big_dreamer_amd/ (ctypes) mirrors the reference's own call surface on top of it
 * (TransitionModel.forward, Dreamer.imagine_ahead, lambda_return, Dreamer.train_step ...).
 *
 * Conventions
 *   - plain C, no torch types: raw device pointers (fp32 unless said otherwise), explicit sizes,
 *     hipStream_t passed as void*;
 *   - nothing allocates, frees or synchronises: every buffer (inputs, outputs, saved activations,
 *     workspaces) is caller-owned device memory, every launch is asynchronous on `stream`;
 *   - return value 0 = launched, negative = rejected (bd_last_error() has the text); never throws;
 *   - tensors are contiguous row-major "rows x features" with an explicit leading dimension where one
 *     is given; time-major (time, batch, feature) arrays are passed flattened to rows = time*batch;
 *   - re-entrant per device: one host thread per GPU (one process per rank).
 *
 * Weight layout.  Kernels read weights in a packed MFMA-fragment layout produced by
 * bd_pack_weights() from the PyTorch (out, in) row-major tensors: for W[N][K],
 *   packed[nb][kb][lane][i] = W[nb*16 + (lane&15)][kb*16 + 4*(lane>>4) + i]   (zero padded),
 * i.e. one coalesced 1 KiB read per 16x16 block feeds four v_mfma_f32_16x16x4_f32.
 */
#ifndef BIGDREAMER_HIP_H
#define BIGDREAMER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BD_MAX_LAYERS 6
#define BD_ACT_NONE 0
#define BD_ACT_ELU 1
#define BD_ACT_ELU_GRAD 2   /* conv entry points only: out = value * ELU'(aux), aux = the SAVED ELU output at the same
                            * index as out (the dgrad of a layer and the ELU backward of the layer below in one pass) */

const char* bd_last_error(void);
int bd_version(void);

/* ---- weight packing ------------------------------------------------------------------------- */
typedef struct {
    const float* src; /* PyTorch-layout matrix, element (n,k) at src[n*ld + k]                  */
    float* dst;       /* packed destination, bd_packed_floats(N,K) floats (transposed: (K,N))    */
    int ld;           /* leading dimension of src                                                 */
    int N, K;         /* logical sub-block to pack (rows n < N, cols k < K of src)                */
    int transpose;    /* 0: pack W (out=N,in=K); 1: pack W^T (out=K,in=N) for the dgrad kernels  */
} bd_pack_desc;

size_t bd_packed_floats(int N, int K);
/* descs: DEVICE array of n descriptors. */
int bd_pack_weights(const bd_pack_desc* descs, int n, void* stream);

/* ---- dense chains: DenseModel / build_mlp (src/models.py:365-408, src/utils.py:368-404) ------ */
typedef struct {
    const float* w;    /* packed weights (out=N, in=K)                                            */
    const float* bias; /* [N] or NULL                                                             */
    int N, K;
    int act;           /* BD_ACT_*: applied to this layer's output                                */
    float* save;       /* optional [M x N] post-activation output kept for the backward, or NULL  */
} bd_layer;

typedef struct {
    int M;                          /* rows                                                       */
    const float* in0; int ld0, w0;  /* input = [in0 | in1] (torch.cat(..., dim=-1)); in1 optional  */
    const float* in1; int ld1, w1;
    int n_layers;
    bd_layer layer[BD_MAX_LAYERS];
    float* out; int ldo;            /* last layer's output [M x N_last]                            */
    /* Optional one-hot (Categorical) input segment BESIDE [in0 | in1]: gD factors of gC classes given as class indices
     * gidx [M x gD] (uint8).  Layer 0 then computes  act(W0 [in0 | in1] + sum_f gWT[f*gC + gidx[m][f]] + b)  with
     * gWT the plain row-major TRANSPOSE [gD*gC x N0] of the layer's one-hot columns: a gather of gD rows instead of a
     * K = gD*gC contraction (DenseModel(belief, state) on [h; one-hot s]).  gD = 0: absent. */
    const unsigned char* gidx; const float* gWT; int gD, gC;
} bd_mlp_fwd_args;
int bd_mlp_forward(const bd_mlp_fwd_args* a, void* stream);

typedef struct {
    const float* wt;   /* packed W^T (out=K, in=N) -- unused for layer 0 unless din requested      */
    const float* saved;/* this layer's saved post-activation output [M x N] (needed iff act!=NONE) */
    int N, K;
    int act;
    float* dpre;       /* optional [M x N] gradient w.r.t. the pre-activation (for bd_wgrad)        */
} bd_layer_bwd;

typedef struct {
    int M;
    const float* dout; int lddo;    /* gradient w.r.t. the last layer's output [M x N_last]        */
    float dout_scale;               /* multiplies dout on load (1.0f for none)                     */
    int n_layers;
    bd_layer_bwd layer[BD_MAX_LAYERS];
    float* din0; int ld0, w0;       /* optional gradient w.r.t. the input, split like the forward   */
    float* din1; int ld1, w1;
    int accumulate;                 /* 1: din += , 0: din =                                         */
} bd_mlp_bwd_args;
int bd_mlp_backward(const bd_mlp_bwd_args* a, void* stream);

/* Tall form of the two chain entry points (csrc/mlp.hip, mlp_*_tall_kernel): for M >= 8192 rows and layer widths of at
 * most 15 column blocks, 48-row workgroups with one in-place LDS image and (row tile, column block) pairs balanced over
 * the four waves, accumulators in the transposed (row x 4 consecutive columns) form; layer widths must be multiples of
 * 4 floats and saved / dpre buffers 16-byte aligned, otherwise the 16-row form runs.  Same arguments, same results
 * (summation order within a dot product is unchanged).  mode 1: on (the default), 0: always the 16/32-row form,
 * 2: on for every M (diagnostics), -1: as the environment says (BD_MLP_TALL=0 switches it off). */
int bd_mlp_set_tall(int mode);

/* dW[N x K] (+)= dpre^T[N x M] * act[M x K],  db[N] (+)= column sums of dpre (db may be NULL).
 * Deterministic split-M (slabs in `ws`, then a fixed-order reduction); accumulate=1 adds to dW/db;
 * ws must hold bd_wgrad_ws_floats(M,N,K) floats (its capacity ws_floats is checked).
 * (What autograd's AddmmBackward computes for every nn.Linear on the path.) */
size_t bd_wgrad_ws_floats(int M, int N, int K);
int bd_wgrad(const float* dpre, int ldp, const float* act, int lda, int M, int N, int K,
             float* dW, int ldw, float* db, int accumulate, float* ws, size_t ws_floats, void* stream);

/* Grouped form: every weight-gradient GEMM of one backward pass in ONE launch (+ one grouped reduce).  Fill the
 * caller fields of each descriptor on the host, let bd_wgrad_plan add the launch plan (and tell the slab workspace
 * size), copy the table to device memory once, then call bd_wgrad_grouped every step.  Rows [0, M1) take their
 * activations from act1, rows [M1, M) from act2[row - M1] (set M1 = M and act2 = NULL for a single source). */
typedef struct {
    const float* dpre; int ldp;               /* [M x N] pre-activation gradients                            */
    const float* act1; int lda1; int M1;
    const float* act2; int lda2;
    int M, N, K;
    float* dW; int ldw;                        /* [N x K] output                                              */
    float* db;                                 /* [N] or NULL                                                 */
    /* filled by bd_wgrad_plan */
    int splits, rows_per, tiles_n, tiles_k, block_begin, red_begin;
    unsigned long long ws_off;
    /* optional gathered `act` (conv weight gradients, pattern F of bd_conv_gemm): when g_nseg > 0, row m = (img, y, x)
     * over g_gh x g_gw per image takes act(m, k) = act1[((img*g_IH + 2y + s)*g_IW + 2x)*g_C + off], s = k / g_seglen,
     * off = k % g_seglen (K = g_nseg * g_seglen; lda1 / act2 unused, M1 = M). */
    int g_nseg, g_seglen, g_gh, g_gw, g_IH, g_IW, g_C, g_pad;
} bd_wgrad_desc;
int bd_wgrad_plan(bd_wgrad_desc* descs_host, int n, int* total_blocks, int* total_red_blocks, size_t* ws_floats);
int bd_wgrad_grouped(const bd_wgrad_desc* descs_dev, int n, int total_blocks, int total_red_blocks, float* ws,
                     void* stream);
/* The two launches of bd_wgrad_grouped separately (phase 1: the slab GEMM kernel; phase 2: the fixed-order reduce), so
 * that a benchmark can bracket the GEMM kernel alone with HIP events.  phase 0 = both = bd_wgrad_grouped. */
int bd_wgrad_grouped_phase(const bd_wgrad_desc* descs_dev, int n, int total_blocks, int total_red_blocks, float* ws,
                           int phase, void* stream);

/* ---- conv stacks of the pixel configurations (CnnImageEncoder src/models.py:527-564, ObservationModel
 * src/models.py:319-362; F.conv2d / F.conv_transpose2d and their autograd backward) as gather-GEMMs on NHWC images:
 *   out[m][n] = act( sum_k A(m, k) W[n][k] + bias[n] ),  m = (img, y, x) over an imgs x gh x gw row grid,
 *   A(m, k): s = k / seglen, off = k % seglen;  iy = y*sy + y0 + s*ss;  ix0 = x*sx + x0;
 *            value = in[((img*IH + iy)*IW + ix0)*C + off]   (0 outside the image when mask = 1),
 *   out row of m: out + ((img*OH + y*osy + oy0)*OW + x*osx + ox0)*ldo.
 * Pattern F (stride-2 VALID conv forward, transposed-conv dgrad): nseg = k, seglen = k*C, sy = sx = 2, ss = 1, dense output.
 * Pattern T (one parity class (py,px) of a stride-2 transposed conv forward / conv dgrad): nseg = Ta, seglen = Tb*C,
 *   sy = sx = 1, ss = -1, x0 = -(Tb-1), mask = 1, osy = osx = 2, oy0 = py, ox0 = px; weights from bd_conv_pack_class. */
typedef struct {
    const float* in; float* out;
    const float* w;       /* packed (bd_pack_weights / bd_conv_pack_class): out = N, in = K                 */
    const float* bias;    /* [N] or NULL                                                                   */
    int imgs, gh, gw, N, K;
    int nseg, seglen, C, IH, IW, sy, y0, ss, sx, x0, mask, cshift, vec4;
    int OH, OW, osy, oy0, osx, ox0, ldo;
    int act;              /* BD_ACT_*                                                                      */
    int fuse_cq;          /* pattern T with its four parity classes fused: N = 4*fuse_cq columns, column n = cls*fuse_cq
                           * + c goes to pixel (2y + (cls>>1), 2x + (cls&1)), channel c (osy = osx = 2, oy0 = ox0 = 0;
                           * gh x gw = the class-(0,0) grid; bias indexed by c); 0 = one class per call                */
    const float* aux;     /* BD_ACT_ELU_GRAD: saved outputs, same layout as `out`; else unused                         */
} bd_conv_args;
int bd_conv_gemm(const bd_conv_args* a, void* stream);
/* Stride-2 VALID convolution of a THIN image (C <= 4 channels) into 32 channels, NHWC: out (imgs, OH, OW, 32) =
 * act(conv(in (imgs, IH, IW, C), W) + bias), W plain row-major [32][(ky, kx, c)] with row stride ldw (the stored parameter
 * as it lies in the buffer; bias may be NULL).  Conv2d(3 -> 32, k4) forward (src/models.py:538) and the dgrad of
 * ConvTranspose2d(32 -> 3, k6) (src/models.py:347).  k*k*C <= 108, OW <= 32. */
int bd_conv_thin_forward(const float* in, int imgs, int IH, int IW, int C, int k, const float* W, int ldw, const float* bias,
                         int act, const float* aux, float* out, void* stream);   /* aux: BD_ACT_ELU_GRAD only (else NULL) */
/* dst (packed) [n = inner][k = (a, b', outer)] = src[outer][py+2a][px+2(Tb-1-b')][inner], src stored (outer, ky, kx, inner) */
int bd_conv_pack_class(const float* src, float* dst, int Couter, int Cinner, int ksz, int py, int px, int Ta, int Tb,
                       void* stream);
/* all four parity classes at once (bd_conv_args.fuse_cq): dst [n = cls*Cinner + c][k = (a, b', outer)], T x T taps with
 * T = (ksz+1)/2, zero where the tap falls outside the kernel (odd ksz, parity 1) */
int bd_conv_pack_fused(const float* src, float* dst, int Couter, int Cinner, int ksz, void* stream);
/* g *= ELU'(y) in place from saved ELU outputs (n a multiple of 4) */
int bd_elu_backward(float* g, const float* y, size_t n, void* stream);
/* out[n] = sum_m rows[m][n] of an [M x N] row-major matrix, N <= 256 (bias gradient of a transposed-conv layer);
 * ws: bd_colsum_ws_floats(N) floats; fixed summation order */
size_t bd_colsum_ws_floats(int N);
int bd_colsum(const float* rows, size_t M, int N, float* out, float* ws, void* stream);
/* (imgs, C, HW) -> (imgs, HW, C) when to_nhwc, the reverse otherwise */
int bd_image_layout(const float* src, float* dst, int imgs, int C, int HW, int to_nhwc, void* stream);

/* ---- RSSM observe scan: TransitionModel.forward with embeddings (src/models.py:191-299) ------
 * One persistent launch walks all T steps; a workgroup owns 16 batch rows (rows are independent, so
 * there is no inter-workgroup synchronisation).  The prior head (src/models.py:256) does not feed the
 * recurrence when embeddings are given, so the host runs it batched over all T*B rows afterwards
 * (bd_mlp_forward + bd_gauss_head_forward).  The embedding half of the posterior's first layer is
 * hoisted out of the loop: pre_emb = embeddings @ W_q1[:, Be:]^T. */
typedef struct {
    int T, B, Be, S, A, Hd;
    /* packed weights */
    const float* w_embed_s; const float* w_embed_a; const float* b_embed; /* fc_embed_state_action.0[:, :S] / [:, S:] */
    const float* w_ir; const float* w_iz; const float* w_in;   /* rnn.weight_ih rows r,z,n (Be,Be) */
    const float* w_hr; const float* w_hz; const float* w_hn;   /* rnn.weight_hh rows r,z,n (Be,Be) */
    const float* b_ih; const float* b_hh;                       /* [3*Be] each                      */
    const float* w_q1h; const float* b_q1;         /* belief_posterior.model.0[:, :Be]: (Hd, Be)   */
    const float* w_q2m; const float* w_q2s;        /* belief_posterior.model.2 rows [0,S) / [S,2S) */
    const float* b_q2;                             /* [2*S]                                        */
    /* inputs */
    const float* init_belief;  /* [B x Be]                                                        */
    const float* init_state;   /* [B x S]                                                         */
    const float* actions;      /* [T x B x A]                                                     */
    const float* nonterm;      /* [T x B] or NULL (nonterminals=None, src/models.py:247)          */
    const float* pre_emb;      /* [T x B x Hd] hoisted embedding projection (no bias)              */
    const float* eps_post;     /* [T x B x S] standard normal                                     */
    float min_std;             /* 0.1                                                             */
    /* outputs */
    float* feat;      /* [T x B x (Be+S)]: [belief_{t+1} | posterior_state_{t+1}]                  */
    float* post_mean; /* [T x B x S]                                                              */
    float* post_std;  /* [T x B x S]                                                              */
    /* saved for the backward (all NULL for inference) */
    float* sv_s;      /* [T x B x S]  masked previous state (input of the embed layer)             */
    float* sv_x;      /* [T x B x Be] embed output                                                 */
    float* sv_gates;  /* [T x B x 4*Be]: r, z, n, (W_hn h + b_hn)                                  */
    float* sv_q;      /* [T x B x Hd] posterior hidden (post-ELU)                                  */
} bd_observe_fwd_args;
int bd_observe_forward(const bd_observe_fwd_args* a, void* stream);

typedef struct {
    int T, B, Be, S, A, Hd;
    /* transposed packed weights */
    const float* wt_embed_s;                                 /* (S, Be)                            */
    const float* wt_ir; const float* wt_iz; const float* wt_in;
    const float* wt_hr; const float* wt_hz; const float* wt_hn;
    const float* wt_q1h;                                     /* (Be, Hd)                           */
    const float* wt_q2m; const float* wt_q2s;                /* (Hd, S) each                        */
    /* forward inputs / saved */
    const float* init_belief; const float* nonterm; const float* eps_post;
    const float* feat; const float* post_std;
    const float* sv_x; const float* sv_gates; const float* sv_q;
    /* incoming gradients */
    const float* dfeat;      /* [T x B x (Be+S)] from the obs/reward heads (+ prior head on the belief part) */
    const float* dpost_mean; /* [T x B x S] from the KL term (or NULL)                              */
    const float* dpost_std;  /* [T x B x S] (or NULL)                                               */
    float min_std;
    /* outputs: pre-activation gradients for bd_wgrad */
    float* d_embed_pre;  /* [T x B x Be]                                                            */
    float* d_gi;         /* [T x B x 3*Be] (r,z,n) gradient w.r.t. W_ih x + b_ih                     */
    float* d_gh;         /* [T x B x 3*Be] gradient w.r.t. W_hh h + b_hh                             */
    float* d_q1_pre;     /* [T x B x Hd]   (also the gradient of pre_emb)                            */
    float* d_q2_out;     /* [T x B x 2*S]                                                           */
} bd_observe_bwd_args;
int bd_observe_backward(const bd_observe_bwd_args* a, void* stream);

/* Cluster variant of the observe scan for small batches (B=50 -> only 4 row tiles): C = bd_observe_cluster_size(B, Be)
 * workgroups (CUs) share each 16-row tile.  The GRU contraction is split by output column blocks over the members,
 * everything small is computed redundantly, and one all-gather per time step goes through `ws` (write-through
 * stores + per-member flags + sc1 loads, bounded spins).  Same arguments and results as bd_observe_forward /
 * bd_observe_backward; `ws` holds bd_observe_cluster_ws_floats(B, Be) floats, is zero-filled ONCE by the caller when
 * it is allocated, and its flags are zeroed by a memset node on `stream` ahead of each launch.  The error word inside
 * it (float index bd_observe_cluster_err_offset(B), a u32) is STICKY: a member that times out waiting for its peers
 * ORs 1 (forward) / 2 (backward) into it, launches never clear it, so a time-out of any launch stays visible until
 * bd_observe_cluster_status reads it.  bd_observe_cluster_status synchronises `stream`, returns non-zero with the
 * error text if the word is set, and clears it.  A host that already copies results back every step (the engine's
 * log fetch) reads the word in the same transfer instead.  bd_observe_cluster_set_spin_limit (0 = default, about
 * seconds) exists so that tests can force a time-out.  The launches return an error (never abort the queue) when the
 * kernel's static + dynamic LDS would exceed the CU's 160 KiB. */
int bd_observe_cluster_size(int B, int Be);   /* workgroups per 16-row tile; 0 = use bd_observe_forward/backward */
/* Two cluster forms sit behind the same calls.  Round 3 (csrc/observe_ksplit.hip, the default where the cluster has one
 * member per 16-column belief block and Hd <= Be): EVERY layer of the step is split along K over the members with the
 * weight slices resident in registers -- gate partials reduce-scatter, posterior-hidden partials reduce-scatter, head
 * partials all-reduce: three hand-offs per step, nothing computed redundantly.  Round 1 (csrc/observe_cluster.hip, wide
 * batches with two belief blocks per member, or bd_observe_cluster_set_ksplit(0)): only the GRU is split, by output
 * columns, one all-gather per step, the small layers recomputed by every member.  Identical arguments and results. */
int bd_observe_cluster_set_ksplit(int mode);  /* 3 = K-split, forward GRU split by output columns (two hand-offs per forward
                                               * step; the default); 1 = K-split, GRU split along K in both directions;
                                               * 2 = as 1 with granule hand-offs ("the data is the flag"); 0 = round-1 form;
                                               * -1 = default (environment BD_OBS_KSPLIT, else 3) */
size_t bd_observe_cluster_ws_floats(int B, int Be);
int bd_observe_forward_cluster(const bd_observe_fwd_args* a, float* ws, size_t ws_floats, void* stream);
int bd_observe_backward_cluster(const bd_observe_bwd_args* a, float* ws, size_t ws_floats, void* stream);
int bd_observe_cluster_status(float* ws, int B, void* stream);
size_t bd_observe_cluster_err_offset(int B);
int bd_observe_cluster_set_spin_limit(unsigned limit);   /* returns 0 */

/* GaussianBeliefModel tail (src/models.py:70-73): out[M x 2S] -> mean, std=softplus(raw)+min_std,
 * state = mean + std*eps.  Used for the batched prior of the observe scan. */
int bd_gauss_head_forward(const float* out, const float* eps, int M, int S, float min_std,
                          float* mean, float* std, float* state, void* stream);
/* d out = [dmean + dstate, (dstd + dstate*eps) * sigmoid(raw)];  dstate/dmean/dstd may be NULL */
int bd_gauss_head_backward(const float* out, const float* eps, const float* dstate, const float* dmean,
                           const float* dstd, int M, int S, float* dout, void* stream);

/* ---- imagination rollout: Dreamer.imagine_ahead + get_action (src/dreamer.py:179-237,429-444),
 *      ActorModel (src/models.py:506-517), SampleDist.entropy / TanhBijector (src/models.py:630-733).
 * One persistent launch walks all Hm = planning_horizon-1 steps; a workgroup owns 16 trajectories. */
typedef struct {
    int N, Hm, Be, S, A, Hd, n_samples;
    /* frozen world model (packed) */
    const float* w_embed_s; const float* w_embed_a; const float* b_embed;
    const float* w_ir; const float* w_iz; const float* w_in;
    const float* w_hr; const float* w_hz; const float* w_hn;
    const float* b_ih; const float* b_hh;
    const float* w_p1; const float* b_p1;          /* belief_prior.model.0 (Hd, Be)                 */
    const float* w_p2m; const float* w_p2s; const float* b_p2;
    /* actor (packed): layer 0 split in belief / state columns, 3 more hidden layers, output layer split
       in mean rows / std rows (model.8: (2A, Hd)) */
    const float* w_a0h; const float* w_a0s; const float* w_a[3]; const float* b_a[4];
    const float* w_a4m; const float* w_a4s; const float* b_a4;
    /* inputs */
    const float* start_feat;   /* [N x (Be+S)] detached posterior features                          */
    const float* eps_action;   /* [Hm x N x A]                                                      */
    const float* eps_entropy;  /* [Hm x n_samples x N x A]                                          */
    const float* eps_prior;    /* [Hm x N x S]                                                      */
    float min_std, act_raw_init_std, act_min_std, act_mean_scale;
    /* outputs */
    float* feat;          /* [Hm x N x (Be+S)] imagined [belief | prior_state]                      */
    float* prior_mean;    /* [Hm x N x S] (may be NULL)                                             */
    float* prior_std;     /* [Hm x N x S]                                                           */
    float* entropy;       /* [Hm x N]                                                               */
    float* action;        /* [Hm x N x A] tanh-squashed sample                                      */
    /* saved for the backward (all NULL for inference) */
    float* sv_actor;      /* [4][Hm x N x Hd] actor hidden activations                              */
    float* sv_act_stats;  /* [Hm x N x 4A]: tanh(m/scale), sigmoid(raw+init), d ent/d mean, d ent/d std */
    float* sv_x;          /* [Hm x N x Be]                                                          */
    float* sv_gates;      /* [Hm x N x 4*Be]                                                        */
    float* sv_p;          /* [Hm x N x Hd]                                                          */
    size_t sv_actor_stride; /* floats between the layers of sv_actor; 0 = Hm*N*Hd.  Lets a rollout be launched in
                               two time segments (second segment: Hm, start_feat and every [Hm x ...] pointer shifted) */
} bd_imagine_fwd_args;
int bd_imagine_forward(const bd_imagine_fwd_args* a, void* stream);
/* The two launches of bd_imagine_forward separately (a caller that saves the actor statistics may run the entropy
 * estimate on another stream: it is off the recurrence).  bd_imagine_forward_scan leaves (mean, std) of every action in
 * slots 2, 3 of sv_act_stats and writes no entropy when sv_act_stats != NULL (with NULL it computes the entropy in the
 * scan and the second call is not needed); bd_actor_entropy turns them into entropy[Hm x N] and d entropy / d mean, / d std
 * in the same slots (SampleDist.entropy / TanhBijector, src/models.py:630-733). */
int bd_imagine_forward_scan(const bd_imagine_fwd_args* a, void* stream);
int bd_actor_entropy(const float* eps_entropy, float* act_stats, float* entropy, int Hm, int N, int A, int n_samples,
                     void* stream);

typedef struct {
    int N, Hm, Be, S, A, Hd;
    const float* wt_embed_s; const float* wt_embed_a;        /* (S, Be), (A, Be)                    */
    const float* wt_ir; const float* wt_iz; const float* wt_in;
    const float* wt_hr; const float* wt_hz; const float* wt_hn;
    const float* wt_p1; const float* wt_p2m; const float* wt_p2s;
    const float* wt_a[3];                 /* transposes of actor layers 1..3 (layer 0: input detached) */
    const float* wt_a4m; const float* wt_a4s;                /* (Hd, A) each                        */
    /* forward tensors */
    const float* start_feat; const float* feat; const float* prior_std; const float* action;
    const float* eps_action; const float* eps_prior;
    const float* sv_actor; const float* sv_act_stats; const float* sv_x; const float* sv_gates;
    const float* sv_p;
    float min_std;
    /* incoming gradients */
    const float* dfeat;     /* [Hm x N x (Be+S)] from reward / value heads                           */
    float dentropy;         /* d loss / d entropy[t][n] (constant: -entropy_weight/(Hm*N))           */
    /* outputs for bd_wgrad */
    float* d_actor_pre;     /* [4][Hm x N x Hd], or NULL: the actor's hidden layers are off the recurrence (detached
                             * input); the caller then runs them as one chain over all rows from d_actor_out:
                             * bd_mlp_backward(layers = actor, dout = d_actor_out, saved = sv_actor, dpre outputs)  */
    float* d_actor_out;     /* [Hm x N x 2A]                                                        */
    const float* ent_weight; /* optional [Hm x N] per-element factor on dentropy (use_discount=True: the cumulative
                              * discount weights of the actor objective, src/dreamer.py:346-351), or NULL          */
} bd_imagine_bwd_args;
int bd_imagine_backward(const bd_imagine_bwd_args* a, void* stream);

/* ---- lambda-return (src/dreamer.py:447-471); bootstrap = value[Hm-1] as at dreamer.py:332 ----- */
int bd_lambda_return_forward(const float* reward, const float* value, int Hm, int N, float discount,
                             float lambda_, float* returns, void* stream);
/* gradient of the scan: d returns = dreturns[Hm x N] if non-NULL else the constant dret_const.
 * Writes dreward and dvalue (dvalue[0] = 0: value[0] is never used; the bootstrap duplicate of
 * value[Hm-1] is folded into dvalue[Hm-1]). */
int bd_lambda_return_backward(const float* dreturns, float dret_const, int Hm, int N, float discount,
                              float lambda_, float* dreward, float* dvalue, void* stream);

/* ---- perf-mode noise: Philox4x32-10, counter-based (csrc/bd_rng.h).  key = seed, counter = (index of a group of four
 * values, stream id, step): any element of any stream of any step is computable on its own.  The parity path never uses
 * this: tests pass the reference's draws as explicit arrays.
 * bd_rng_fill: up to BD_RNG_MAX_TENSORS noise tensors in ONE launch (standard normals: src/models.py:72 randn_like,
 * src/dreamer.py:443 rsample; Exp(1): the variates torch.multinomial's single-draw path consumes, src/models.py:114-115). */
#define BD_RNG_MAX_TENSORS 6
#define BD_RNG_NORMAL 0
#define BD_RNG_EXPONENTIAL 1
typedef struct {
    int n;
    unsigned long long seed;
    unsigned long long step;
    struct {
        float* p;
        size_t count;
        int kind;                 /* BD_RNG_NORMAL | BD_RNG_EXPONENTIAL */
        unsigned stream_id;       /* distinct per tensor */
    } t[BD_RNG_MAX_TENSORS];
} bd_rng_fill_args;
int bd_rng_fill(const bd_rng_fill_args* a, void* stream);
/* The generator's core on the host: out4 = Philox4x32-10(counter ctr4, key key2) (known-answer tests, no GPU needed). */
int bd_philox4x32_10(const unsigned* ctr4, const unsigned* key2, unsigned* out4);
/* bd_actor_entropy with the n_samples draws per (row, action dim) generated IN the kernel (stream `stream_id` of `seed`,
 * `step`): the (Hm x n_samples x N x A) entropy noise tensor -- 13.7 MB per step at configs[1], 233 MB at A = 17 -- is never
 * written to or read from HBM.  Same estimator, same outputs as bd_actor_entropy (SampleDist.entropy, src/models.py:725-733). */
int bd_actor_entropy_rng(unsigned long long seed, unsigned long long step, unsigned stream_id, float* act_stats,
                         float* entropy, int Hm, int N, int A, int n_samples, void* stream);

/* ---- plain GEMM  C[M x N] (+)= A[M x K] B[N x K]^T  (csrc/gemm.hip; fp32 MFMA, exact fp32 products and sums).
 * The pixel decoder's one plain GEMM: the dgrad of ConvTranspose2d(E -> 128, k5, s2) on a 1 x 1 map
 * (src/models.py:338-341), d l0[M x E] = g[M x 3200] W[E x 3200]^T.  Any K / leading dimensions / alignment (16-byte
 * loads per operand where its base, leading dimension and K allow them). */
int bd_gemm_nt(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K, int accumulate,
               void* stream);

/* ---- Categorical latents: CategoricalBeliefModel tail (src/models.py:108-117) and the Categorical branch of
 * Dreamer._kl_loss (src/dreamer.py:102-106,131-144).  logits / state / probs are [rows x D*C], D groups of C classes.
 * Forward: probs = softmax(logits) per group; state = one_hot(argmax(probs / q_noise)) with q_noise ~ Exp(1), which is
 * torch.multinomial's single-draw algorithm behind OneHotCategoricalStraightThrough.rsample(); the straight-through
 * term adds exactly zero.  Backward: dlogits = probs * (dstate - sum_c probs * dstate) per group. */
int bd_categorical_head_forward(const float* logits, const float* q_noise, int rows, int D, int C, float* state,
                                float* probs, void* stream);
int bd_categorical_head_backward(const float* dstate, const float* probs, int rows, int D, int C, float* dlogits,
                                 void* stream);
/* KL(post || prior) between the D categorical factors.  Forward writes the RAW sum into scalars[slot] (sum over all
 * rows*D groups; with sum_form -- kl_balance == -1 -- the sum over rows of max(sum_D KL, free_nats)); ws as for the
 * other reductions.  Backward (same conventions as bd_kl_backward): balanced form lhs -> dprior, rhs -> dpost, both
 * gated by the free-nats clamp of the mean scalars[slot] * inv_count. */
int bd_kl_categorical_forward(const float* post_logits, const float* prior_logits, int rows, int D, int C, float free_nats,
                              int sum_form, float* scalars, int slot, float* ws, void* stream);
int bd_kl_categorical_backward(const float* post_logits, const float* prior_logits, int rows, int D, int C, float free_nats,
                               float kl_balance, float weight, float inv_count, const float* scalars, int slot,
                               float* dpost, float* dprior, void* stream);

/* ---- Categorical RSSM scans: latent_distribution="Categorical" (BASELINE configs[4], algorithm=dreamerV2) ----
 * TransitionModel.forward (src/models.py:191-299, Categorical branches :226-228,258-260,269-271) and
 * Dreamer.imagine_ahead (src/dreamer.py:179-237, :205-206,224-227) with CategoricalBeliefModel heads
 * (src/models.py:76-117): the state is D one-hot factors of C classes, S = D*C (src/planet.py:56-57), sampled with
 * straight-through gradients (OneHotCategoricalStraightThrough.rsample()).
 * Same persistent 16-row-tile design as the Gaussian scans.  What changes with the latent:
 *   - the state is carried as D class indices per row; W [s; a] of the embed layer (and W0 [h; s] of the actor) is the
 *     MFMA contraction over the action / belief columns PLUS A GATHER of D rows of the transposed state weights
 *     (`*_sT`: plain row-major [S x out], row k = W[:, k]) -- 32 x out adds instead of a K = 1024 contraction;
 *   - the head is hidden -> S logits (MFMA, 256 columns at a time through LDS), then per factor softmax and
 *     sample = argmax(probs / q), q ~ Exp(1): torch.multinomial's single-draw path, q is an explicit input [rows x S];
 *   - backward: d logits = probs * (g - sum_c probs * g) per factor (straight-through: d state / d probs = I) with
 *     g = d loss / d state = heads' gradient + W_es^T (d embed pre-activation of the next step) (* nonterminal).
 * Outputs: feat [rows x (Be+S)] = [h; one-hot s] (dense: the reward / value / decoder chains read it as any other
 * feature matrix), the class indices sidx [rows x D] (uint8), the logits [rows x S].
 * C <= 256; S <= 256, or 256 % C == 0 and S % 16 == 0. */
typedef struct {
    int T, B, Be, D, C, A, Hd;
    const float* w_embed_sT;                                   /* plain [S x Be]: row k = W_e[:, k]           */
    const float* w_embed_a; const float* b_embed;              /* packed (Be, A), [Be]                        */
    const float* w_ir; const float* w_iz; const float* w_in;
    const float* w_hr; const float* w_hz; const float* w_hn;
    const float* b_ih; const float* b_hh;
    const float* w_q1h; const float* b_q1;                     /* posterior layer 0, belief columns           */
    const float* w_q2; const float* b_q2;                      /* packed (S, Hd), [S]: posterior logits       */
    const float* init_belief;                                  /* [B x Be]                                    */
    const float* init_state;                                   /* [B x S]: zeros, or one-hot per factor       */
    const float* actions;                                      /* [T x B x A]                                 */
    const float* nonterm;                                      /* [T x B] or NULL                             */
    const float* pre_emb;                                      /* [T x B x Hd] hoisted embeddings @ W_q1[:, Be:]^T */
    const float* q_post;                                       /* [T x B x S] Exp(1) draws of the sampler     */
    float* feat;                                               /* out [T x B x (Be+S)]                        */
    float* post_logits;                                        /* out [T x B x S]                             */
    unsigned char* sidx;                                       /* out [T x B x D] sampled class per factor    */
    float* sv_s;       /* [T x B x S] masked input state of every step (dense; embed weight gradient), or NULL */
    float* sv_x; float* sv_gates; float* sv_q;                 /* as bd_observe_fwd_args, or NULL              */
} bd_observe_cat_fwd_args;
int bd_observe_cat_forward(const bd_observe_cat_fwd_args* a, void* stream);

typedef struct {
    int T, B, Be, D, C, A, Hd;
    const float* wt_embed_s;                                   /* packed transpose (S, Be): d state = W_es^T d pre */
    const float* wt_ir; const float* wt_iz; const float* wt_in;
    const float* wt_hr; const float* wt_hz; const float* wt_hn;
    const float* wt_q1h;                                       /* (Be, Hd)                                    */
    const float* wt_q2;                                        /* packed transpose (Hd, S)                    */
    const float* init_belief; const float* nonterm;
    const float* feat; const float* post_logits;
    const float* sv_x; const float* sv_gates; const float* sv_q;
    const float* dfeat;          /* [T x B x (Be+S)] from the obs / reward heads (+ prior head on the belief part) */
    const float* dpost_logits;   /* [T x B x S] from the KL term, or NULL                                        */
    float* d_embed_pre; float* d_gi; float* d_gh; float* d_q1_pre;   /* as bd_observe_bwd_args                   */
    float* d_q2_out;             /* [T x B x S] gradient of the posterior logits                                */
} bd_observe_cat_bwd_args;
int bd_observe_cat_backward(const bd_observe_cat_bwd_args* a, void* stream);

/* Cluster variant of the Categorical observe scan (csrc/observe_cat_cluster.hip): Cm = bd_observe_cat_cluster_size(B, Be,
 * D, C, max_wgs) workgroups, one per CU, share each 16-row tile -- GRU column blocks and groups of D / Cm factors of the
 * posterior head are split over the members, two hand-offs per time step through `ws` (forward: new belief, sampled class
 * indices; backward: d posterior-hidden all-reduce, [belief-gradient carry | d embed pre-activation]).  Same arguments,
 * same results as bd_observe_cat_forward / bd_observe_cat_backward (TransitionModel.forward, src/models.py:191-299,
 * Categorical branches).  cluster size 0 = shape not supported or tiles * Cm > max_wgs: use the one-workgroup-per-tile
 * calls.  `ws`: bd_observe_cat_cluster_ws_floats(B, Be, Hd, D, Cm) floats, zero-filled ONCE by the caller; header, sticky
 * error word (bd_observe_cluster_err_offset / bd_observe_cluster_status) and time-out behaviour as for
 * bd_observe_forward_cluster. */
int bd_observe_cat_cluster_size(int B, int Be, int D, int C, int max_wgs);
size_t bd_observe_cat_cluster_ws_floats(int B, int Be, int Hd, int D, int Cm);
int bd_observe_cat_forward_cluster(const bd_observe_cat_fwd_args* a, int Cm, float* ws, size_t ws_floats, void* stream);
int bd_observe_cat_backward_cluster(const bd_observe_cat_bwd_args* a, int Cm, float* ws, size_t ws_floats, void* stream);

typedef struct {
    int N, Hm, Be, D, C, A, Hd, n_samples;
    const float* w_embed_sT; const float* w_embed_a; const float* b_embed;
    const float* w_ir; const float* w_iz; const float* w_in;
    const float* w_hr; const float* w_hz; const float* w_hn;
    const float* b_ih; const float* b_hh;
    const float* w_p1; const float* b_p1;                      /* belief_prior.model.0                        */
    const float* w_p2; const float* b_p2;                      /* packed (S, Hd), [S]: prior logits           */
    const float* w_a0h;                                        /* actor layer 0, belief columns, packed       */
    const float* w_a0sT;                                       /* actor layer 0, state columns: plain [S x Hd] */
    const float* w_a[3]; const float* b_a[4];
    const float* w_a4m; const float* w_a4s; const float* b_a4;
    const float* start_feat;                                   /* [N x (Be+S)] posterior features (one-hot s)  */
    const unsigned char* start_sidx;                           /* [N x D] class indices of start_feat's one-hot state, or NULL: every
                                                                  factor of start_feat[:, Be:] is then all-zero (fed as zeros, like
                                                                  the reference does with its initial state) or (scaled) one-hot   */
    const float* eps_action; const float* eps_entropy;         /* as bd_imagine_fwd_args                      */
    const float* q_prior;                                      /* [Hm x N x S] Exp(1) draws                   */
    float act_raw_init_std, act_min_std, act_mean_scale;
    float* feat;                                               /* out [Hm x N x (Be+S)]                       */
    unsigned char* sidx;                                       /* out [Hm x N x D]                            */
    float* prior_logits;                                       /* out [Hm x N x S]                            */
    float* entropy; float* action;
    float* sv_actor; float* sv_act_stats; float* sv_x; float* sv_gates; float* sv_p;   /* or NULL            */
} bd_imagine_cat_fwd_args;
int bd_imagine_cat_forward(const bd_imagine_cat_fwd_args* a, void* stream);
/* (eps_entropy == NULL with sv_act_stats != NULL: the scan alone -- the caller runs the entropy estimate itself,
 * bd_actor_entropy or bd_actor_entropy_rng on sv_act_stats, as after bd_imagine_forward_scan) */

typedef struct {
    int N, Hm, Be, D, C, A, Hd;
    const float* wt_embed_s; const float* wt_embed_a;
    const float* wt_ir; const float* wt_iz; const float* wt_in;
    const float* wt_hr; const float* wt_hz; const float* wt_hn;
    const float* wt_p1;                                        /* (Be, Hd)                                    */
    const float* wt_p2;                                        /* packed transpose (Hd, S)                    */
    const float* wt_a[3]; const float* wt_a4m; const float* wt_a4s;
    const float* start_feat; const float* feat; const float* prior_logits; const float* action;
    const float* eps_action;
    const float* sv_actor; const float* sv_act_stats; const float* sv_x; const float* sv_gates; const float* sv_p;
    const float* dfeat;          /* [Hm x N x (Be+S)] from the reward / value heads                            */
    float dentropy;
    float* d_actor_pre; float* d_actor_out;                    /* as bd_imagine_bwd_args                      */
    const float* ent_weight;     /* as bd_imagine_bwd_args, or NULL                                              */
} bd_imagine_cat_bwd_args;
int bd_imagine_cat_backward(const bd_imagine_cat_bwd_args* a, void* stream);

/* ---- CEM planner: MPCPlanner.forward (src/planner.py:28-90) -------------------------------------
 * One CEM iteration = bd_plan_rollout + bd_cem_refit.  rows = B * cand candidate action sequences (row = b * cand + c);
 * the rollout forms a_t = act_mean[t][b] + act_std[t][b] * eps_action[t][row] (src/planner.py:60-62), runs the
 * prior-only RSSM step (src/models.py:241-256, embeddings=None, nonterminals=None) and the reward model
 * (src/models.py:365-408) per step in LDS and writes the actions [H x rows x A] and the summed predicted reward per
 * candidate (src/planner.py:68-72).  w_r[0] is the reward model's first layer packed over K = Be+S. */
typedef struct {
    int rows, H, cand, Be, S, A, Hd;
    const float* w_embed_s; const float* w_embed_a; const float* b_embed;
    const float* w_ir; const float* w_iz; const float* w_in;
    const float* w_hr; const float* w_hz; const float* w_hn;
    const float* b_ih; const float* b_hh;
    const float* w_p1; const float* b_p1;                 /* belief_prior.model.0 */
    const float* w_p2m; const float* w_p2s; const float* b_p2;   /* belief_prior.model.2 rows [:S] / [S:] */
    const float* w_r[5]; const float* b_r[5];             /* reward_model.model.{0,2,4,6,8} */
    float min_std;
    const float* init_belief;   /* [B x Be]  (every candidate of environment b starts from row b, src/planner.py:37) */
    const float* init_state;    /* [B x S]   */
    const float* act_mean;      /* [H x B x A] */
    const float* act_std;       /* [H x B x A] */
    const float* eps_action;    /* [H x rows x A]  the torch.randn draws of src/planner.py:53-59 */
    const float* eps_state;     /* [H x rows x S]  the prior-state draws (src/models.py:72) */
    float* actions;             /* out [H x rows x A] */
    float* returns;             /* out [rows]; NULL: skip the reward model and write `feat` instead */
    float* feat;                /* out [H x rows x (Be+S)] ([h'; s'] per step) or NULL */
} bd_plan_args;
int bd_plan_rollout(const bd_plan_args* a, void* stream);
/* Re-fit the action belief to the `top` best candidates of every environment (src/planner.py:74-87):
 * mean / stdev [H x B x A] <- mean and biased standard deviation over the selected action sequences.
 * returns is [ret_steps x B*cand]: ret_steps = 1 for summed returns, H for per-step reward predictions (summed here). */
int bd_cem_refit(const float* returns, int ret_steps, const float* actions, int H, int B, int cand, int top, int A,
                 float* mean, float* stdev, void* stream);

/* ---- losses (src/planet.py:252-284, src/dreamer.py:110-146,342-383) ---------------------------
 * Reductions write RAW SUMS into a small device "scalar board" (float array); the host turns them
 * into the logged means after one D2H copy per step, and multi-GPU runs all-reduce the board's KL slot
 * before the free-nats clamp.  Each reduction is two launches: per-workgroup fp64 partials in `ws`
 * (bd_reduce_ws_floats() floats), then a fixed-order final sum (deterministic, no float atomics). */
/* scalars[slot] = sum over all elements of 0.5 (t-p)^2 + 0.5 ln 2pi;  dpred = (p-t) * grad_scale
 * (-Independent(Normal(pred,1)).log_prob(target); dpred may be NULL) */
int bd_normal_nll(const float* pred, int ldp, const float* target, int ldt, int rows, int D,
                  float grad_scale, float* dpred, int ldd, float* scalars, int slot, float* ws, void* stream);
/* Dreamer._discount_loss (src/dreamer.py:239-251, use_discount=True): RAW sum over n elements of
 * -log Bernoulli(logits).prob(target) into scalars[slot]; dlogits (may be NULL) = (sigmoid(logit) - target) * grad_scale. */
int bd_bernoulli_nll(const float* logits, const float* target, size_t n, float grad_scale, float* dlogits, float* scalars,
                     int slot, float* ws, void* stream);

/* balanced form (sum_form=0): scalars[slot] = sum of elementwise KL(N(qm,qs) || N(pm,ps));
 * summed form  (sum_form=1, kl_balance == -1): scalars[slot] = sum_rows max(sum_S KL, free_nats) */
int bd_kl_forward(const float* qm, const float* qs, const float* pm, const float* ps, int rows, int S,
                  float free_nats, int sum_form, float* scalars, int slot, float* ws, void* stream);
/* Gradients of weight * kl_loss.  Balanced: the clamp decision uses mean = scalars[slot] * inv_count
 * (inv_count = 1 / global element count, after any all-reduce of the slot). */
int bd_kl_backward(const float* qm, const float* qs, const float* pm, const float* ps, int rows, int S,
                   float free_nats, float kl_balance, float weight, float inv_count, const float* scalars,
                   int slot, float* dqm, float* dqs, float* dpm, float* dps, void* stream);
/* scalars[slot] = sum(x) / sum(x^2) */
int bd_sum(const float* x, size_t n, float* scalars, int slot, float* ws, void* stream);
int bd_sumsq(const float* x, size_t n, float* scalars, int slot, float* ws, void* stream);

/* ---- clip_grad_norm_ + Adam (src/dreamer.py:299-302,362-367,386-391; torch.optim.Adam with
 *      weight_decay = L2-in-gradient).  total_norm = sqrt(scalars[sqnorm_slot]); the gradient buffer
 *      is left clipped, as the reference leaves .grad. */
int bd_adam_step(float* p, float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2,
                 float eps, float weight_decay, int step, float max_norm, const float* scalars,
                 int sqnorm_slot, void* stream);
/* polyak_update (src/utils.py:56-78): target = src*weight + target*(1-weight) */
int bd_polyak(float* target, const float* src, size_t n, float weight, void* stream);

/* ---- replay gather: ExperienceReplay._retrieve_batch (src/memory.py:70-85) --------------------- */
int bd_replay_gather(const float* src, const int64_t* idx, int n_idx, int width, float* dst, void* stream);

/* pixel replay: gather uint8 frames (pixels bytes per row, multiple of 4) and dequantise in one pass:
 * out = floor(u8 / 2^(8-bits)) / 2^bits - 0.5 + noise / 2^bits  (preprocess_observation_, src/utils.py:299-317;
 * noise ~ U[0,1) is an explicit [n_idx x pixels] input) */
int bd_replay_gather_pixels(const unsigned char* src, const int64_t* idx, int n_idx, int pixels, int bit_depth,
                            const float* noise, float* dst, void* stream);
/* The same with the dequantisation noise U[0, 1) drawn in the kernel (Philox4x32-10, csrc/bd_rng.h; stream 6 of `seed`,
 * counter = element quad, `step`): perf mode -- no noise tensor, no library RNG launch. */
int bd_replay_gather_pixels_rng(const unsigned char* src, const int64_t* idx, int n_idx, int pixels, int bit_depth,
                                unsigned long long seed, unsigned long long step, float* dst, void* stream);

size_t bd_reduce_ws_floats(void);

#ifdef __cplusplus
}
#endif
#endif /* BIGDREAMER_HIP_H */
