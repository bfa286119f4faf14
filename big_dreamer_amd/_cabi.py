"""ctypes binding of ``libbigdreamer_hip.so`` (C ABI declared in ``include/bigdreamer_hip.h``).

The product path has no CPU fallback: importing this module without the built HIP library raises.
Build it with ``python -c "import __graft_entry__ as g; g.build()"`` (or ``make -C big_dreamer_amd/csrc``).
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# BD_LIB selects an alternative build of the same library (kernel tuning experiments only)
LIB_PATH = os.environ.get("BD_LIB") or os.path.join(_HERE, "libbigdreamer_hip.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: the MI355X HIP extension has not been built "
        "(run `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")

lib = C.CDLL(LIB_PATH)

BD_MAX_LAYERS = 6
ACT_NONE, ACT_ELU, ACT_ELU_GRAD = 0, 1, 2
P = C.c_void_p
F32 = C.c_float
I32 = C.c_int


class PackDesc(C.Structure):
    _fields_ = [("src", P), ("dst", P), ("ld", I32), ("N", I32), ("K", I32), ("transpose", I32)]


class Layer(C.Structure):
    _fields_ = [("w", P), ("bias", P), ("N", I32), ("K", I32), ("act", I32), ("save", P)]


class MlpFwdArgs(C.Structure):
    _fields_ = [("M", I32), ("in0", P), ("ld0", I32), ("w0", I32), ("in1", P), ("ld1", I32), ("w1", I32),
                ("n_layers", I32), ("layer", Layer * BD_MAX_LAYERS), ("out", P), ("ldo", I32),
                ("gidx", P), ("gWT", P), ("gD", I32), ("gC", I32)]


class LayerBwd(C.Structure):
    _fields_ = [("wt", P), ("saved", P), ("N", I32), ("K", I32), ("act", I32), ("dpre", P)]


class MlpBwdArgs(C.Structure):
    _fields_ = [("M", I32), ("dout", P), ("lddo", I32), ("dout_scale", F32), ("n_layers", I32),
                ("layer", LayerBwd * BD_MAX_LAYERS), ("din0", P), ("ld0", I32), ("w0", I32),
                ("din1", P), ("ld1", I32), ("w1", I32), ("accumulate", I32)]


BD_RNG_MAX_TENSORS = 6
BD_RNG_NORMAL, BD_RNG_EXPONENTIAL = 0, 1


class RngTensor(C.Structure):
    _fields_ = [("p", P), ("count", C.c_size_t), ("kind", I32), ("stream_id", C.c_uint)]


class RngFillArgs(C.Structure):
    _fields_ = [("n", I32), ("seed", C.c_ulonglong), ("step", C.c_ulonglong), ("t", RngTensor * BD_RNG_MAX_TENSORS)]


class WgradDesc(C.Structure):
    _fields_ = [("dpre", P), ("ldp", I32), ("act1", P), ("lda1", I32), ("M1", I32), ("act2", P), ("lda2", I32),
                ("M", I32), ("N", I32), ("K", I32), ("dW", P), ("ldw", I32), ("db", P),
                ("splits", I32), ("rows_per", I32), ("tiles_n", I32), ("tiles_k", I32), ("block_begin", I32),
                ("red_begin", I32), ("ws_off", C.c_ulonglong),
                ("g_nseg", I32), ("g_seglen", I32), ("g_gh", I32), ("g_gw", I32), ("g_IH", I32), ("g_IW", I32), ("g_C", I32),
                ("g_pad", I32)]


def _ptr_fields(names):
    return [(n, P) for n in names]


class ObserveFwdArgs(C.Structure):
    _fields_ = ([(n, I32) for n in ("T", "B", "Be", "S", "A", "Hd")] + _ptr_fields(
        ["w_embed_s", "w_embed_a", "b_embed", "w_ir", "w_iz", "w_in", "w_hr", "w_hz", "w_hn", "b_ih", "b_hh",
         "w_q1h", "b_q1", "w_q2m", "w_q2s", "b_q2", "init_belief", "init_state", "actions", "nonterm", "pre_emb",
         "eps_post"]) + [("min_std", F32)] + _ptr_fields(
        ["feat", "post_mean", "post_std", "sv_s", "sv_x", "sv_gates", "sv_q"]))


class ObserveBwdArgs(C.Structure):
    _fields_ = ([(n, I32) for n in ("T", "B", "Be", "S", "A", "Hd")] + _ptr_fields(
        ["wt_embed_s", "wt_ir", "wt_iz", "wt_in", "wt_hr", "wt_hz", "wt_hn", "wt_q1h", "wt_q2m", "wt_q2s",
         "init_belief", "nonterm", "eps_post", "feat", "post_std", "sv_x", "sv_gates", "sv_q", "dfeat", "dpost_mean",
         "dpost_std"]) + [("min_std", F32)] + _ptr_fields(
        ["d_embed_pre", "d_gi", "d_gh", "d_q1_pre", "d_q2_out"]))


class ImagineFwdArgs(C.Structure):
    _fields_ = ([(n, I32) for n in ("N", "Hm", "Be", "S", "A", "Hd", "n_samples")] + _ptr_fields(
        ["w_embed_s", "w_embed_a", "b_embed", "w_ir", "w_iz", "w_in", "w_hr", "w_hz", "w_hn", "b_ih", "b_hh", "w_p1",
         "b_p1", "w_p2m", "w_p2s", "b_p2", "w_a0h", "w_a0s"]) + [("w_a", P * 3), ("b_a", P * 4)] + _ptr_fields(
        ["w_a4m", "w_a4s", "b_a4", "start_feat", "eps_action", "eps_entropy", "eps_prior"]) + [
        ("min_std", F32), ("act_raw_init_std", F32), ("act_min_std", F32), ("act_mean_scale", F32)] + _ptr_fields(
        ["feat", "prior_mean", "prior_std", "entropy", "action", "sv_actor", "sv_act_stats", "sv_x", "sv_gates",
         "sv_p"]) + [("sv_actor_stride", C.c_size_t)])


class ImagineBwdArgs(C.Structure):
    _fields_ = ([(n, I32) for n in ("N", "Hm", "Be", "S", "A", "Hd")] + _ptr_fields(
        ["wt_embed_s", "wt_embed_a", "wt_ir", "wt_iz", "wt_in", "wt_hr", "wt_hz", "wt_hn", "wt_p1", "wt_p2m",
         "wt_p2s"]) + [("wt_a", P * 3)] + _ptr_fields(
        ["wt_a4m", "wt_a4s", "start_feat", "feat", "prior_std", "action", "eps_action", "eps_prior", "sv_actor",
         "sv_act_stats", "sv_x", "sv_gates", "sv_p"]) + [("min_std", F32)] + _ptr_fields(["dfeat"]) + [
        ("dentropy", F32)] + _ptr_fields(["d_actor_pre", "d_actor_out", "ent_weight"]))


U8P = C.c_void_p


class ObserveCatFwdArgs(C.Structure):
    _fields_ = ([(n, I32) for n in ("T", "B", "Be", "D", "C", "A", "Hd")] + _ptr_fields(
        ["w_embed_sT", "w_embed_a", "b_embed", "w_ir", "w_iz", "w_in", "w_hr", "w_hz", "w_hn", "b_ih", "b_hh",
         "w_q1h", "b_q1", "w_q2", "b_q2", "init_belief", "init_state", "actions", "nonterm", "pre_emb", "q_post",
         "feat", "post_logits", "sidx", "sv_s", "sv_x", "sv_gates", "sv_q"]))


class ObserveCatBwdArgs(C.Structure):
    _fields_ = ([(n, I32) for n in ("T", "B", "Be", "D", "C", "A", "Hd")] + _ptr_fields(
        ["wt_embed_s", "wt_ir", "wt_iz", "wt_in", "wt_hr", "wt_hz", "wt_hn", "wt_q1h", "wt_q2", "init_belief", "nonterm",
         "feat", "post_logits", "sv_x", "sv_gates", "sv_q", "dfeat", "dpost_logits", "d_embed_pre", "d_gi", "d_gh",
         "d_q1_pre", "d_q2_out"]))


class ImagineCatFwdArgs(C.Structure):
    _fields_ = ([(n, I32) for n in ("N", "Hm", "Be", "D", "C", "A", "Hd", "n_samples")] + _ptr_fields(
        ["w_embed_sT", "w_embed_a", "b_embed", "w_ir", "w_iz", "w_in", "w_hr", "w_hz", "w_hn", "b_ih", "b_hh", "w_p1",
         "b_p1", "w_p2", "b_p2", "w_a0h", "w_a0sT"]) + [("w_a", P * 3), ("b_a", P * 4)] + _ptr_fields(
        ["w_a4m", "w_a4s", "b_a4", "start_feat", "start_sidx", "eps_action", "eps_entropy", "q_prior"]) + [
        ("act_raw_init_std", F32), ("act_min_std", F32), ("act_mean_scale", F32)] + _ptr_fields(
        ["feat", "sidx", "prior_logits", "entropy", "action", "sv_actor", "sv_act_stats", "sv_x", "sv_gates", "sv_p"]))


class ImagineCatBwdArgs(C.Structure):
    _fields_ = ([(n, I32) for n in ("N", "Hm", "Be", "D", "C", "A", "Hd")] + _ptr_fields(
        ["wt_embed_s", "wt_embed_a", "wt_ir", "wt_iz", "wt_in", "wt_hr", "wt_hz", "wt_hn", "wt_p1", "wt_p2"]) + [
        ("wt_a", P * 3)] + _ptr_fields(
        ["wt_a4m", "wt_a4s", "start_feat", "feat", "prior_logits", "action", "eps_action", "sv_actor", "sv_act_stats",
         "sv_x", "sv_gates", "sv_p", "dfeat"]) + [("dentropy", F32)] + _ptr_fields(["d_actor_pre", "d_actor_out", "ent_weight"]))


class PlanArgs(C.Structure):
    _fields_ = ([(n, I32) for n in ("rows", "H", "cand", "Be", "S", "A", "Hd")] + _ptr_fields(
        ["w_embed_s", "w_embed_a", "b_embed", "w_ir", "w_iz", "w_in", "w_hr", "w_hz", "w_hn", "b_ih", "b_hh", "w_p1",
         "b_p1", "w_p2m", "w_p2s", "b_p2"]) + [("w_r", P * 5), ("b_r", P * 5), ("min_std", F32)] + _ptr_fields(
        ["init_belief", "init_state", "act_mean", "act_std", "eps_action", "eps_state", "actions", "returns", "feat"]))


class ConvArgs(C.Structure):
    _fields_ = [("in_", P), ("out", P), ("w", P), ("bias", P),
                ("imgs", I32), ("gh", I32), ("gw", I32), ("N", I32), ("K", I32),
                ("nseg", I32), ("seglen", I32), ("C", I32), ("IH", I32), ("IW", I32), ("sy", I32), ("y0", I32),
                ("ss", I32), ("sx", I32), ("x0", I32), ("mask", I32), ("cshift", I32), ("vec4", I32),
                ("OH", I32), ("OW", I32), ("osy", I32), ("oy0", I32), ("osx", I32), ("ox0", I32), ("ldo", I32),
                ("act", I32), ("fuse_cq", I32), ("aux", P)]


# every symbol include/bigdreamer_hip.h declares, with its signature
_SIGS = {
    "bd_last_error": (C.c_char_p, []),
    "bd_version": (I32, []),
    "bd_packed_floats": (C.c_size_t, [I32, I32]),
    "bd_pack_weights": (I32, [P, I32, P]),
    "bd_mlp_forward": (I32, [C.POINTER(MlpFwdArgs), P]),
    "bd_mlp_backward": (I32, [C.POINTER(MlpBwdArgs), P]),
    "bd_mlp_set_tall": (I32, [I32]),
    "bd_wgrad_ws_floats": (C.c_size_t, [I32, I32, I32]),
    "bd_wgrad": (I32, [P, I32, P, I32, I32, I32, I32, P, I32, P, I32, P, C.c_size_t, P]),
    "bd_wgrad_plan": (I32, [C.POINTER(WgradDesc), I32, C.POINTER(I32), C.POINTER(I32), C.POINTER(C.c_size_t)]),
    "bd_wgrad_grouped": (I32, [P, I32, I32, I32, P, P]),
    "bd_wgrad_grouped_phase": (I32, [P, I32, I32, I32, P, I32, P]),
    "bd_observe_forward": (I32, [C.POINTER(ObserveFwdArgs), P]),
    "bd_observe_backward": (I32, [C.POINTER(ObserveBwdArgs), P]),
    "bd_observe_cluster_size": (I32, [I32, I32]),
    "bd_observe_cluster_ws_floats": (C.c_size_t, [I32, I32]),
    "bd_observe_forward_cluster": (I32, [C.POINTER(ObserveFwdArgs), P, C.c_size_t, P]),
    "bd_observe_backward_cluster": (I32, [C.POINTER(ObserveBwdArgs), P, C.c_size_t, P]),
    "bd_observe_cluster_status": (I32, [P, I32, P]),
    "bd_observe_cluster_err_offset": (C.c_size_t, [I32]),
    "bd_observe_cluster_set_spin_limit": (I32, [C.c_uint]),
    "bd_observe_cluster_set_ksplit": (I32, [I32]),
    "bd_gauss_head_forward": (I32, [P, P, I32, I32, F32, P, P, P, P]),
    "bd_gauss_head_backward": (I32, [P, P, P, P, P, I32, I32, P, P]),
    "bd_observe_cat_forward": (I32, [C.POINTER(ObserveCatFwdArgs), P]),
    "bd_observe_cat_backward": (I32, [C.POINTER(ObserveCatBwdArgs), P]),
    "bd_observe_cat_cluster_size": (I32, [I32, I32, I32, I32, I32]),
    "bd_observe_cat_cluster_ws_floats": (C.c_size_t, [I32, I32, I32, I32, I32]),
    "bd_observe_cat_forward_cluster": (I32, [C.POINTER(ObserveCatFwdArgs), I32, P, C.c_size_t, P]),
    "bd_observe_cat_backward_cluster": (I32, [C.POINTER(ObserveCatBwdArgs), I32, P, C.c_size_t, P]),
    "bd_imagine_cat_forward": (I32, [C.POINTER(ImagineCatFwdArgs), P]),
    "bd_imagine_cat_backward": (I32, [C.POINTER(ImagineCatBwdArgs), P]),
    "bd_imagine_forward": (I32, [C.POINTER(ImagineFwdArgs), P]),
    "bd_imagine_forward_scan": (I32, [C.POINTER(ImagineFwdArgs), P]),
    "bd_actor_entropy": (I32, [P, P, P, I32, I32, I32, I32, P]),
    "bd_imagine_backward": (I32, [C.POINTER(ImagineBwdArgs), P]),
    "bd_lambda_return_forward": (I32, [P, P, I32, I32, F32, F32, P, P]),
    "bd_rng_fill": (I32, [C.POINTER(RngFillArgs), P]),
    "bd_philox4x32_10": (I32, [C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.POINTER(C.c_uint)]),
    "bd_actor_entropy_rng": (I32, [C.c_ulonglong, C.c_ulonglong, C.c_uint, P, P, I32, I32, I32, I32, P]),
    "bd_gemm_nt": (I32, [P, I32, P, I32, P, I32, I32, I32, I32, I32, P]),
    "bd_categorical_head_forward": (I32, [P, P, I32, I32, I32, P, P, P]),
    "bd_categorical_head_backward": (I32, [P, P, I32, I32, I32, P, P]),
    "bd_kl_categorical_forward": (I32, [P, P, I32, I32, I32, F32, I32, P, I32, P, P]),
    "bd_kl_categorical_backward": (I32, [P, P, I32, I32, I32, F32, F32, F32, F32, P, I32, P, P, P]),
    "bd_plan_rollout": (I32, [C.POINTER(PlanArgs), P]),
    "bd_cem_refit": (I32, [P, I32, P, I32, I32, I32, I32, I32, P, P, P]),
    "bd_lambda_return_backward": (I32, [P, F32, I32, I32, F32, F32, P, P, P]),
    "bd_normal_nll": (I32, [P, I32, P, I32, I32, I32, F32, P, I32, P, I32, P, P]),
    "bd_bernoulli_nll": (I32, [P, P, C.c_size_t, F32, P, P, I32, P, P]),
    "bd_kl_forward": (I32, [P, P, P, P, I32, I32, F32, I32, P, I32, P, P]),
    "bd_kl_backward": (I32, [P, P, P, P, I32, I32, F32, F32, F32, F32, P, I32, P, P, P, P, P]),
    "bd_sum": (I32, [P, C.c_size_t, P, I32, P, P]),
    "bd_sumsq": (I32, [P, C.c_size_t, P, I32, P, P]),
    "bd_adam_step": (I32, [P, P, P, P, C.c_size_t, F32, F32, F32, F32, F32, I32, F32, P, I32, P]),
    "bd_polyak": (I32, [P, P, C.c_size_t, F32, P]),
    "bd_replay_gather": (I32, [P, P, I32, I32, P, P]),
    "bd_replay_gather_pixels": (I32, [P, P, I32, I32, I32, P, P, P]),
    "bd_replay_gather_pixels_rng": (I32, [P, P, I32, I32, I32, C.c_ulonglong, C.c_ulonglong, P, P]),
    "bd_reduce_ws_floats": (C.c_size_t, []),
    "bd_conv_gemm": (I32, [C.POINTER(ConvArgs), P]),
    "bd_conv_thin_forward": (I32, [P, I32, I32, I32, I32, I32, P, I32, P, I32, P, P, P]),
    "bd_conv_pack_class": (I32, [P, P, I32, I32, I32, I32, I32, I32, I32, P]),
    "bd_conv_pack_fused": (I32, [P, P, I32, I32, I32, P]),
    "bd_elu_backward": (I32, [P, P, C.c_size_t, P]),
    "bd_image_layout": (I32, [P, P, I32, I32, I32, I32, P]),
    "bd_colsum_ws_floats": (C.c_size_t, [I32]),
    "bd_colsum": (I32, [P, C.c_size_t, I32, P, P, P]),
}

for _name, (_res, _args) in _SIGS.items():
    _fn = getattr(lib, _name)          # AttributeError here = header/library mismatch
    _fn.restype = _res
    _fn.argtypes = _args

EXPORTED = tuple(_SIGS)


def check(rc: int) -> None:
    if rc != 0:
        raise RuntimeError("bigdreamer_hip: " + lib.bd_last_error().decode())


def ptr(t) -> int:
    """Device pointer of a tensor (None -> NULL).  The kernels read every operand as dense row-major with the leading
    dimension the caller states, so a tensor must be contiguous or a row-strided 2-D view of one (unit inner stride,
    e.g. ``W[:, :S]`` or ``feat[:, :Be]``): a permuted or column-strided view would be read with the wrong strides."""
    if t is None:
        return None
    if not (t.dtype in (torch.float32, torch.int64, torch.uint8) and t.is_cuda):
        raise TypeError("expected a CUDA fp32/int64/uint8 tensor")
    if not (t.is_contiguous() or (t.dim() == 2 and t.stride(1) == 1 and t.stride(0) >= t.shape[1])):
        raise ValueError(f"non-contiguous tensor (shape {tuple(t.shape)}, strides {t.stride()}) passed to a HIP kernel: "
                         "call .contiguous() first")
    return t.data_ptr()


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def packed_floats(N: int, K: int) -> int:
    return ((N + 15) // 16) * ((K + 15) // 16) * 256
