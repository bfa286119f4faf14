"""Host side of the hand-written conv stacks (csrc/conv.hip): geometry of the two gather patterns and the packed
weight copies.  Activations are NHWC fp32; Conv2d weights are stored (co, ky, kx, ci), ConvTranspose2d weights
(ci, ky, kx, co) -- see conv.hip.  All convolutions of the reference's stacks are stride 2, no padding
(src/models.py:319-362, 527-564)."""
from __future__ import annotations

import ctypes as C
from typing import List, Tuple

import torch

from . import _cabi as cabi

lib = cabi.lib
ptr = cabi.ptr


def _log2(c: int) -> int:
    return c.bit_length() - 1 if c & (c - 1) == 0 else 0


def conv_out(size: int, k: int) -> int:
    """Output size of a stride-2 VALID convolution."""
    return (size - k) // 2 + 1


def convT_out(size: int, k: int) -> int:
    return (size - 1) * 2 + k


def taps(k: int, parity: int) -> int:
    """Kernel taps ky = parity + 2a < k of one output parity class of a stride-2 transposed convolution."""
    return (k - parity + 1) // 2


def pattern_f(inp: torch.Tensor, out: torch.Tensor, w_packed: torch.Tensor, bias, imgs: int, IH: int, IW: int, Cin: int,
              k: int, N: int, act: int, aux=None) -> None:
    """out (imgs, OH, OW, N) = conv_stride2_valid(inp (imgs, IH, IW, Cin)) with packed W[N][(ky, kx, ci)] (+bias, act).
    Also the dgrad of a transposed convolution (inp = gradient of its output); act = ACT_ELU_GRAD multiplies by ELU' of the
    saved outputs `aux` (same shape as out): the ELU backward of the layer below in the same pass."""
    OH, OW = conv_out(IH, k), conv_out(IW, k)
    a = cabi.ConvArgs()
    a.in_, a.out, a.w, a.bias = ptr(inp), ptr(out), ptr(w_packed), ptr(bias)
    a.imgs, a.gh, a.gw, a.N, a.K = imgs, OH, OW, N, k * k * Cin
    a.nseg, a.seglen, a.C, a.IH, a.IW = k, k * Cin, Cin, IH, IW
    a.sy, a.y0, a.ss, a.sx, a.x0, a.mask = 2, 0, 1, 2, 0, 0
    a.vec4 = int(Cin % 4 == 0)
    a.cshift = _log2(Cin)
    a.OH, a.OW, a.osy, a.oy0, a.osx, a.ox0, a.ldo = OH, OW, 1, 0, 1, 0, N
    a.act, a.aux = act, ptr(aux)
    assert act != cabi.ACT_ELU_GRAD or (aux is not None and aux.numel() == out.numel())
    cabi.check(lib.bd_conv_gemm(C.byref(a), cabi.stream()))


def thin_f(inp: torch.Tensor, out: torch.Tensor, w_plain: torch.Tensor, bias, imgs: int, IH: int, IW: int, Cin: int, k: int,
           act: int, aux=None) -> None:
    """pattern_f for a THIN image (Cin <= 4) and 32 output channels: out (imgs, OH, OW, 32) from the PLAIN weight matrix
    [32][(ky, kx, c)] (a 2-D view of the stored parameter; bd_conv_thin_forward, csrc/conv.hip)."""
    assert w_plain.dim() == 2 and w_plain.shape[0] == 32 and w_plain.shape[1] == k * k * Cin and w_plain.stride(1) == 1
    assert act != cabi.ACT_ELU_GRAD or (aux is not None and aux.numel() == out.numel())
    cabi.check(lib.bd_conv_thin_forward(ptr(inp), imgs, IH, IW, Cin, k, w_plain.data_ptr(), w_plain.stride(0), ptr(bias), act,
                                        ptr(aux), ptr(out), cabi.stream()))


def pattern_t(inp: torch.Tensor, out: torch.Tensor, w_classes: List[torch.Tensor], bias, imgs: int, IH: int, IW: int,
              Cin: int, k: int, N: int, OH: int, OW: int, act: int) -> None:
    """out (imgs, OH, OW, N) = convT_stride2(inp (imgs, IH, IW, Cin)) restricted to the rows/cols < OH, OW, from the four
    parity-class weight packs (class_packs).  Also the dgrad of a stride-2 convolution (inp = gradient of its output,
    OH/OW = the convolution's input size: pixels no window covers get a zero gradient)."""
    assert Cin % 4 == 0 and Cin & (Cin - 1) == 0, "T pattern: channel count must be a power of two >= 4"
    for py in range(2):
        for px in range(2):
            Ta, Tb = taps(k, py), taps(k, px)
            gh, gw = (OH - py + 1) // 2, (OW - px + 1) // 2
            if gh <= 0 or gw <= 0:
                continue
            a = cabi.ConvArgs()
            a.in_, a.out, a.w, a.bias = ptr(inp), ptr(out), ptr(w_classes[2 * py + px]), ptr(bias)
            a.imgs, a.gh, a.gw, a.N, a.K = imgs, gh, gw, N, Ta * Tb * Cin
            a.nseg, a.seglen, a.C, a.IH, a.IW = Ta, Tb * Cin, Cin, IH, IW
            a.sy, a.y0, a.ss, a.sx, a.x0, a.mask = 1, 0, -1, 1, -(Tb - 1), 1
            a.vec4, a.cshift = 1, _log2(Cin)
            a.OH, a.OW, a.osy, a.oy0, a.osx, a.ox0, a.ldo = OH, OW, 2, py, 2, px, N
            a.act = act
            cabi.check(lib.bd_conv_gemm(C.byref(a), cabi.stream()))


def pattern_t_fused(inp: torch.Tensor, out: torch.Tensor, w_fused: torch.Tensor, bias, imgs: int, IH: int, IW: int, Cin: int,
                    k: int, N: int, OH: int, OW: int, act: int, aux=None) -> None:
    """pattern_t with the four parity classes in ONE launch: they read the same T x T window (T = (k+1)//2), so their
    weights are the columns of one matrix (fused_pack) and the window is gathered once instead of four times."""
    assert Cin % 4 == 0 and Cin & (Cin - 1) == 0, "T pattern: channel count must be a power of two >= 4"
    T = (k + 1) // 2
    a = cabi.ConvArgs()
    a.in_, a.out, a.w, a.bias = ptr(inp), ptr(out), ptr(w_fused), ptr(bias)
    a.imgs, a.gh, a.gw, a.N, a.K = imgs, (OH + 1) // 2, (OW + 1) // 2, 4 * N, T * T * Cin
    a.nseg, a.seglen, a.C, a.IH, a.IW = T, T * Cin, Cin, IH, IW
    a.sy, a.y0, a.ss, a.sx, a.x0, a.mask = 1, 0, -1, 1, -(T - 1), 1
    a.vec4, a.cshift = 1, _log2(Cin)
    a.OH, a.OW, a.osy, a.oy0, a.osx, a.ox0, a.ldo = OH, OW, 2, 0, 2, 0, N
    a.act, a.fuse_cq, a.aux = act, N, ptr(aux)
    assert act != cabi.ACT_ELU_GRAD or (aux is not None and aux.numel() == out.numel())
    cabi.check(lib.bd_conv_gemm(C.byref(a), cabi.stream()))


def fused_pack_floats(Couter: int, Cinner: int, k: int) -> int:
    T = (k + 1) // 2
    return cabi.packed_floats(4 * Cinner, T * T * Couter)


def pack_fused(stored: torch.Tensor, dst: torch.Tensor, Couter: int, Cinner: int, k: int) -> None:
    """stored: (Couter, k, k, Cinner) contiguous -> the fused four-class pack of pattern_t_fused."""
    cabi.check(lib.bd_conv_pack_fused(ptr(stored), ptr(dst), Couter, Cinner, k, cabi.stream()))


def class_pack_floats(Couter: int, Cinner: int, k: int) -> List[int]:
    return [cabi.packed_floats(Cinner, taps(k, py) * taps(k, px) * Couter) for py in range(2) for px in range(2)]


def pack_classes(stored: torch.Tensor, dsts: List[torch.Tensor], Couter: int, Cinner: int, k: int) -> None:
    """stored: (Couter, k, k, Cinner) contiguous -> the four parity-class packs of the T pattern."""
    for py in range(2):
        for px in range(2):
            cabi.check(lib.bd_conv_pack_class(ptr(stored), ptr(dsts[2 * py + px]), Couter, Cinner, k, py, px, taps(k, py),
                                              taps(k, px), cabi.stream()))


def pack_matrix(src2d: torch.Tensor, dst: torch.Tensor, N: int, K: int, transpose: bool = False) -> None:
    """bd_pack_weights on one (N, K) row-major matrix view (stride(0) = leading dimension)."""
    d = (cabi.PackDesc * 1)(cabi.PackDesc(src2d.data_ptr(), dst.data_ptr(), src2d.stride(0), N, K, int(transpose)))
    raw = torch.frombuffer(bytearray(bytes(d)), dtype=torch.uint8).to(src2d.device)
    cabi.check(lib.bd_pack_weights(raw.data_ptr(), 1, cabi.stream()))
    torch.cuda.current_stream().synchronize()      # `raw` must outlive the launch (test / setup helper only)


def to_nhwc(x: torch.Tensor) -> torch.Tensor:
    imgs, Cc, H, W = x.shape
    out = torch.empty(imgs, H, W, Cc, dtype=torch.float32, device=x.device)
    cabi.check(lib.bd_image_layout(ptr(x.contiguous()), ptr(out), imgs, Cc, H * W, 1, cabi.stream()))
    return out


def to_nchw(x: torch.Tensor) -> torch.Tensor:
    imgs, H, W, Cc = x.shape
    out = torch.empty(imgs, Cc, H, W, dtype=torch.float32, device=x.device)
    cabi.check(lib.bd_image_layout(ptr(x.contiguous()), ptr(out), imgs, Cc, H * W, 0, cabi.stream()))
    return out


def wgrad_desc(dpre2d: torch.Tensor, N: int, act_img: torch.Tensor, imgs: int, gh: int, gw: int, IH: int, IW: int, Cimg: int,
               k: int, dW: torch.Tensor, db) -> "cabi.WgradDesc":
    """Descriptor of a conv weight-gradient GEMM: dW[N][(ky, kx, c)] = sum_m dpre[m][n] * window_F(act_img)(m, (ky, kx, c)),
    m over imgs x gh x gw (gh, gw = the stride-2 VALID output grid of the IH x IW x Cimg image), db[n] = sum_m dpre[m][n].
    Conv2d: dpre = gradient of the conv output (N = co), act_img = the conv input.  ConvTranspose2d: dpre = the layer's
    INPUT rows (N = ci), act_img = gradient of its output -- the result is the (ci, ky, kx, co) stored layout."""
    dsc = cabi.WgradDesc()
    M = imgs * gh * gw
    dsc.dpre, dsc.ldp = ptr(dpre2d), N
    dsc.act1, dsc.lda1, dsc.M1, dsc.act2, dsc.lda2 = ptr(act_img), 0, M, None, 0
    dsc.M, dsc.N, dsc.K = M, N, k * k * Cimg
    dsc.dW, dsc.ldw, dsc.db = ptr(dW), k * k * Cimg, ptr(db)
    dsc.g_nseg, dsc.g_seglen, dsc.g_gh, dsc.g_gw, dsc.g_IH, dsc.g_IW, dsc.g_C = k, k * Cimg, gh, gw, IH, IW, Cimg
    return dsc


def run_wgrad(descs: list) -> None:
    """Plan + launch a list of descriptors once (tests / one-off use; the engine caches tables)."""
    n = len(descs)
    arr = (cabi.WgradDesc * n)(*descs)
    tb, tr, wsf = C.c_int(0), C.c_int(0), C.c_size_t(0)
    cabi.check(lib.bd_wgrad_plan(arr, n, C.byref(tb), C.byref(tr), C.byref(wsf)))
    dev = torch.device("cuda", torch.cuda.current_device())
    table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
    ws = torch.zeros(max(1, wsf.value), dtype=torch.float32, device=dev)
    cabi.check(lib.bd_wgrad_grouped(table.data_ptr(), n, tb.value, tr.value, ptr(ws), cabi.stream()))
    torch.cuda.current_stream().synchronize()
