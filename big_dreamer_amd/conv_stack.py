"""Conv encoder / decoder of the pixel configurations on the hand-written gather-GEMM kernels (csrc/conv.hip).

CnnImageEncoder (src/models.py:527-564): 4 x (Conv2d k4 s2 + ELU), Flatten, Identity | Linear(1024, E).
ObservationModel (src/models.py:319-362): Linear(Be+S, E), ConvT(E,128,k5,s2)+ELU on a 1x1 image (= a Linear to
(5,5,128)), ConvT(128,64,k5)+ELU, ConvT(64,32,k6)+ELU, ConvT(32,3,k6).

Activations are NHWC; conv weights are stored (d0, ky, kx, d1) (engine.ParamGroup).  Forward, dgrad and the weight
gradients of every layer are kernels of this library (bd_conv_gemm patterns F / T, bd_mlp_forward / backward for
the Linear-shaped layers, bd_wgrad_grouped with gathered windows, and bd_gemm_nt (csrc/gemm.hip) for the dgrad of the
1x1 -> 5x5 transposed convolution, a plain (M x 3200) x (3200 x E) GEMM whose K does not fit the LDS-resident row tiles).  Results match the reference's autograd path to the fp32 tolerance of tests/test_hip_parity.py
(pixel golden cases).
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

from . import _cabi as cabi
from . import conv

import os

lib = cabi.lib
ptr = cabi.ptr
THIN = os.environ.get("BD_CONV_THIN", "1") != "0"      # "0": the 3-channel-image layers on the row-tile gather kernels (A/B)
# "1": the ELU backward as the epilogue of the dgrad kernels (BD_ACT_ELU_GRAD; parity-tested).  Off: measured 13.2 vs 13.1 ms
# per step at configs[2] -- the scattered loads of the saved outputs cost the gather kernels more than the separate pass
FUSE_ELU = os.environ.get("BD_CONV_FUSE_ELU", "0") != "0"

ENC = [(3, 32, 4), (32, 64, 4), (64, 128, 4), (128, 256, 4)]          # (ci, co, k); input 64 -> 31 -> 14 -> 6 -> 2
ENC_SIZES = [64, 31, 14, 6, 2]
DEC = [(128, 64, 5), (64, 32, 6), (32, 3, 6)]                          # after the 1x1 -> 5x5 layer; 5 -> 13 -> 30 -> 64
DEC_SIZES = [5, 13, 30, 64]
DEC_IDX = [4, 6, 8]                                                    # decoder.{idx} of the three conv layers above


class ConvStacks:
    def __init__(self, eng) -> None:
        self.e = eng
        d = eng.d
        dev = eng.dev
        self.E = d.E
        self.lin_tail = d.E != 1024
        z = lambda n: torch.zeros(n, dtype=torch.float32, device=dev)
        # ---- packed weight copies (rebuilt by pack() after every model optimiser step) ----
        self.pk_enc_f = [z(cabi.packed_floats(co, k * k * ci)) for ci, co, k in ENC]
        self.pk_enc_t = [None] + [z(conv.fused_pack_floats(co, ci, k)) for ci, co, k in ENC[1:]]     # four classes fused
        self.pk_dec1 = z(cabi.packed_floats(25 * 128, d.E))                       # (ky, kx, co) x ci
        self.bias_dec1 = z(25 * 128)
        self.pk_dec_t = [z(conv.fused_pack_floats(ci, co, k)) for ci, co, k in DEC]
        self.pk_dec_f = [z(cabi.packed_floats(ci, k * k * co)) for ci, co, k in DEC]
        self.pk_dec0 = z(cabi.packed_floats(d.E, d.Be + d.S))
        self.pk_dec0_t = z(cabi.packed_floats(d.E, d.Be + d.S))
        if self.lin_tail:
            self.pk_lin = z(cabi.packed_floats(d.E, 1024))
            self.pk_lin_t = z(cabi.packed_floats(d.E, 1024))
        self._table = None
        self.colsum_ws = z(int(lib.bd_colsum_ws_floats(256)))

    # ------------------------------------------------------------------------------------------ packing
    def _matrix_descs(self):
        e, d = self.e, self.e.d
        out = []

        def add(src2d, dst, N, K, tr=False):
            out.append(cabi.PackDesc(src2d.data_ptr(), dst.data_ptr(), src2d.stride(0), N, K, int(tr)))

        for i, (ci, co, k) in enumerate(ENC):
            add(e.Ws("encoder", f"model.{2 * i}.weight").view(co, k * k * ci), self.pk_enc_f[i], co, k * k * ci)
        # 1x1 -> 5x5 transposed conv as a Linear: out[(ky,kx,co)] = sum_ci x[ci] W[ci][(ky,kx,co)]  (W^T of the stored matrix)
        add(e.Ws("observation_model", "decoder.2.weight").view(d.E, 25 * 128), self.pk_dec1, d.E, 25 * 128, tr=True)
        for j, (ci, co, k) in enumerate(DEC):
            add(e.Ws("observation_model", f"decoder.{DEC_IDX[j]}.weight").view(ci, k * k * co), self.pk_dec_f[j], ci, k * k * co)
        w0 = e.W("observation_model", "decoder.0.weight")
        add(w0, self.pk_dec0, d.E, d.Be + d.S)
        add(w0, self.pk_dec0_t, d.E, d.Be + d.S, tr=True)
        if self.lin_tail:
            wl = e.W("encoder", "model.9.weight")
            add(wl, self.pk_lin, d.E, 1024)
            add(wl, self.pk_lin_t, d.E, 1024, tr=True)
        return out

    def pack(self) -> None:
        e = self.e
        if self._table is None:
            descs = self._matrix_descs()
            arr = (cabi.PackDesc * len(descs))(*descs)
            self._table = (torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(e.dev), len(descs))
        raw, n = self._table
        cabi.check(lib.bd_pack_weights(raw.data_ptr(), n, cabi.stream()))
        for i, (ci, co, k) in enumerate(ENC):
            if i > 0:          # dgrad of conv i (pattern T): stored (co, ky, kx, ci), outer = co
                conv.pack_fused(e.Ws("encoder", f"model.{2 * i}.weight"), self.pk_enc_t[i], co, ci, k)
        for j, (ci, co, k) in enumerate(DEC):      # forward of transposed conv j: stored (ci, ky, kx, co), outer = ci
            conv.pack_fused(e.Ws("observation_model", f"decoder.{DEC_IDX[j]}.weight"), self.pk_dec_t[j], ci, co, k)
        # the 1x1 -> 5x5 layer's bias, once per output pixel
        self.bias_dec1.view(25, 128).copy_(e.W("observation_model", "decoder.2.bias").unsqueeze(0).expand(25, 128))

    # ------------------------------------------------------------------------------------------ forward
    def encode(self, obs4d: torch.Tensor, tag: str = "") -> torch.Tensor:
        """(M, 3, 64, 64) NCHW observations -> embeddings (M, E).  Keeps the NHWC activations for the backward."""
        e = self.e
        M = obs4d.shape[0]
        x = e.buf(tag + "cv_x0", M, 64, 64, 3)
        cabi.check(lib.bd_image_layout(ptr(obs4d.contiguous()), ptr(x), M, 3, 64 * 64, 1, cabi.stream()))
        self.acts_enc = [x]
        for i, (ci, co, k) in enumerate(ENC):
            sz = ENC_SIZES[i + 1]
            y = e.buf(tag + f"cv_a{i + 1}", M, sz, sz, co)
            if i == 0 and THIN:         # the 3-channel image: wave-private band pipelines (bd_conv_thin_forward)
                conv.thin_f(self.acts_enc[-1], y, e.Ws("encoder", "model.0.weight").view(co, k * k * ci),
                            e.W("encoder", "model.0.bias"), M, ENC_SIZES[0], ENC_SIZES[0], ci, k, cabi.ACT_ELU)
            else:
                conv.pattern_f(self.acts_enc[-1], y, self.pk_enc_f[i], e.W("encoder", f"model.{2 * i}.bias"), M, ENC_SIZES[i],
                               ENC_SIZES[i], ci, k, co, cabi.ACT_ELU)
            self.acts_enc.append(y)
        flat = e.buf(tag + "cv_flat", M, 1024)                 # the reference's Flatten order (c, h, w)
        cabi.check(lib.bd_image_layout(ptr(self.acts_enc[-1]), ptr(flat), M, 256, 4, 0, cabi.stream()))
        if not self.lin_tail:
            return flat
        emb = e.buf(tag + "cv_emb", M, self.E)
        e.mlp_forward(M, flat, 1024, 1024, [(self.pk_lin, e.W("encoder", "model.9.bias"), self.E, 1024, cabi.ACT_NONE)], None,
                      emb, self.E, raw_packs=True)
        return emb

    def decode(self, feat: torch.Tensor, tag: str = "") -> torch.Tensor:
        """(M, Be+S) features -> predicted images (M, 64, 64, 3) NHWC.  Keeps the activations for the backward."""
        e, d = self.e, self.e.d
        M, F = feat.shape[0], d.Be + d.S
        l0 = e.buf(tag + "cv_l0", M, self.E)
        e.mlp_forward(M, feat, F, F, [(self.pk_dec0, e.W("observation_model", "decoder.0.bias"), self.E, F, cabi.ACT_NONE)],
                      None, l0, self.E, raw_packs=True)
        d1 = e.buf(tag + "cv_d1", M, 5, 5, 128)
        e.mlp_forward(M, l0, self.E, self.E, [(self.pk_dec1, self.bias_dec1, 25 * 128, self.E, cabi.ACT_ELU)], None, d1,
                      25 * 128, raw_packs=True)
        self.acts_dec = [l0, d1]
        for j, (ci, co, k) in enumerate(DEC):
            sz = DEC_SIZES[j + 1]
            y = e.buf(tag + f"cv_d{j + 2}", M, sz, sz, co)
            conv.pattern_t_fused(self.acts_dec[-1], y, self.pk_dec_t[j], e.W("observation_model", f"decoder.{DEC_IDX[j]}.bias"),
                                 M, DEC_SIZES[j], DEC_SIZES[j], ci, k, co, sz, sz, cabi.ACT_ELU if j < 2 else cabi.ACT_NONE)
            self.acts_dec.append(y)
        return self.acts_dec[-1]

    # ------------------------------------------------------------------------------------------ backward
    def _colsum(self, wb, rows2d: torch.Tensor, M: int, N: int, db: torch.Tensor) -> None:
        """db[n] = sum_m rows[m][n]: bias gradient of a transposed-conv layer (sum over every output pixel)."""
        cabi.check(lib.bd_colsum(ptr(rows2d), M, N, ptr(db), ptr(self.colsum_ws), cabi.stream()))

    def backward_decoder(self, g_pred: torch.Tensor, feat: torch.Tensor, dfeat: torch.Tensor, wb) -> None:
        """g_pred (M, 64, 64, 3): gradient w.r.t. the prediction (consumed in place).  Adds d/d feat into `dfeat` and
        queues every weight-gradient GEMM of the decoder on `wb`."""
        e, d = self.e, self.e.d
        M, F = feat.shape[0], d.Be + d.S
        G = lambda n: e.Gs("observation_model", n)
        g = g_pred
        for j in (2, 1, 0):
            ci, co, k = DEC[j]
            isz, osz = DEC_SIZES[j], DEC_SIZES[j + 1]
            a_in = self.acts_dec[j + 1]                         # this layer's input (post-ELU output of the layer below)
            name = f"decoder.{DEC_IDX[j]}"
            # dW (ci, ky, kx, co) = sum over input pixels of in[m][ci] * window(g)(m, (ky, kx, co));  db = column sums of g
            wb.add(a_in, ci, g, 0, M * isz * isz, ci, k * k * co, G(name + ".weight"), k * k * co, None,
                   gather=(k, k * co, isz, isz, osz, osz, co))
            self._colsum(wb, g, M * osz * osz, co, G(name + ".bias"))
            # d input = strided conv of g with the stored matrix [ci][(ky, kx, co)], then through the ELU of the layer below
            # (the ELU backward is the dgrad kernels' epilogue: BD_ACT_ELU_GRAD with the saved outputs a_in)
            gi = e.buf(f"cv_gd{j + 1}", M, isz, isz, ci)
            if j == 2 and THIN:         # dgrad of ConvT(32 -> 3): a k6 convolution of the 3-channel image gradient
                conv.thin_f(g, gi, e.Ws("observation_model", f"decoder.{DEC_IDX[j]}.weight").view(ci, k * k * co), None, M, osz,
                            osz, co, k, cabi.ACT_ELU_GRAD if FUSE_ELU else cabi.ACT_NONE, a_in if FUSE_ELU else None)
            else:
                conv.pattern_f(g, gi, self.pk_dec_f[j], None, M, osz, osz, co, k, ci,
                               cabi.ACT_ELU_GRAD if FUSE_ELU else cabi.ACT_NONE, a_in if FUSE_ELU else None)
            if not FUSE_ELU:
                cabi.check(lib.bd_elu_backward(ptr(gi), ptr(a_in), gi.numel(), cabi.stream()))
            g = gi
        # the 1x1 -> 5x5 layer as a Linear: dW[ci][(ky,kx,co)] = sum_m l0[m][ci] * g[m][(ky,kx,co)]
        l0 = self.acts_dec[0]
        wb.add(l0, self.E, g, 25 * 128, M, self.E, 25 * 128, G("decoder.2.weight"), 25 * 128, None)
        self._colsum(wb, g, M * 25, 128, G("decoder.2.bias"))
        gl0 = e.buf("cv_gl0", M, self.E)
        # d l0 = g W^T with the stored matrix W (ci, (ky, kx, co)) as it lies in the parameter buffer: a plain K = 3200 GEMM
        cabi.check(lib.bd_gemm_nt(ptr(g), 25 * 128, ptr(e.Ws("observation_model", "decoder.2.weight")), 25 * 128, ptr(gl0),
                                  self.E, M, self.E, 25 * 128, 0, cabi.stream()))
        # Linear(Be+S, E)
        wb.add(gl0, self.E, feat, F, M, self.E, F, e.G("observation_model", "decoder.0.weight"), F,
               e.G("observation_model", "decoder.0.bias"))
        e.mlp_backward(M, gl0, self.E, [(self.pk_dec0_t, None, self.E, F, cabi.ACT_NONE)], [None], [None], din0=dfeat, ld0=F,
                       w0=F, accumulate=True, raw_packs=True)

    def backward_encoder(self, d_emb: torch.Tensor, wb) -> None:
        """d_emb (M, E): gradient w.r.t. the embeddings.  Queues every weight-gradient GEMM of the encoder on `wb`."""
        e = self.e
        M = d_emb.shape[0]
        G = lambda n: e.Gs("encoder", n)
        if self.lin_tail:
            flat = e._buf["cv_flat"]
            wb.add(d_emb, self.E, flat, 1024, M, self.E, 1024, e.G("encoder", "model.9.weight"), 1024,
                   e.G("encoder", "model.9.bias"))
            dflat = e.buf("cv_dflat", M, 1024)
            e.mlp_backward(M, d_emb, self.E, [(self.pk_lin_t, None, self.E, 1024, cabi.ACT_NONE)], [None], [None], din0=dflat,
                           ld0=1024, w0=1024, raw_packs=True)
        else:
            dflat = d_emb
        g = e.buf("cv_ga4", M, 2, 2, 256)
        cabi.check(lib.bd_image_layout(ptr(dflat), ptr(g), M, 256, 4, 1, cabi.stream()))
        for i in (3, 2, 1, 0):
            ci, co, k = ENC[i]
            isz, osz = ENC_SIZES[i], ENC_SIZES[i + 1]
            if i == 3 or not FUSE_ELU:  # d pre-activation (below the top layer it is the epilogue of the dgrad that produced g)
                cabi.check(lib.bd_elu_backward(ptr(g), ptr(self.acts_enc[i + 1]), g.numel(), cabi.stream()))
            wb.add(g, co, self.acts_enc[i], 0, M * osz * osz, co, k * k * ci, G(f"model.{2 * i}.weight"), k * k * ci,
                   G(f"model.{2 * i}.bias"), gather=(k, k * ci, osz, osz, isz, isz, ci))
            if i > 0:
                gi = e.buf(f"cv_ga{i}", M, isz, isz, ci)
                conv.pattern_t_fused(g, gi, self.pk_enc_t[i], None, M, osz, osz, co, k, ci, isz, isz,
                                     cabi.ACT_ELU_GRAD if FUSE_ELU else cabi.ACT_NONE, self.acts_enc[i] if FUSE_ELU else None)
                g = gi
