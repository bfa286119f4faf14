"""GPU tests of the hand-written conv gather-GEMMs (csrc/conv.hip) against torch's F.conv2d / F.conv_transpose2d on the
layer shapes of the reference's CnnImageEncoder / ObservationModel (src/models.py:319-362, 527-564).
The reference values are computed by torch ON THE CPU in fp32 (the ops the reference's disable_cuda=True path runs), not
by MIOpen on the device: inputs are generated on the device, copied to the host for the reference, and the results
compared there."""
import numpy as np
import pytest
import torch
import torch.nn.functional as Fnn

pytestmark = pytest.mark.gpu

ENC = [(3, 32, 4, 64), (32, 64, 4, 31), (64, 128, 4, 14), (128, 256, 4, 6)]        # (Cin, Cout, k, input size)
DEC = [(128, 64, 5, 5), (64, 32, 6, 13), (32, 3, 6, 30)]                           # ConvT after the 1x1 -> 5x5 layer


def _cpu(*ts):
    return [None if t is None else t.detach().cpu() for t in ts]


def _close(got, want, tol=2e-5):
    got, want = got.double().cpu().numpy(), want.double().cpu().numpy()
    err = np.abs(got - want).max()
    assert err <= tol * (1.0 + np.abs(want).max()), (err, np.abs(want).max())


@pytest.mark.parametrize("Cin,Cout,k,size", ENC)
def test_pattern_f_matches_conv2d_and_layout_roundtrip(Cin, Cout, k, size):
    from big_dreamer_amd import _cabi as cabi, conv
    g = torch.Generator(device="cuda").manual_seed(1)
    imgs = 7
    x = torch.randn(imgs, Cin, size, size, device="cuda", generator=g)
    w = torch.randn(Cout, Cin, k, k, device="cuda", generator=g) * 0.1
    b = torch.randn(Cout, device="cuda", generator=g)
    want = Fnn.elu(Fnn.conv2d(*_cpu(x, w, b), stride=2))
    xs = conv.to_nhwc(x)
    assert torch.equal(conv.to_nchw(xs), x)
    stored = w.permute(0, 2, 3, 1).contiguous()                       # (co, ky, kx, ci)
    K = k * k * Cin
    wp = torch.zeros(cabi.packed_floats(Cout, K), device="cuda")
    conv.pack_matrix(stored.view(Cout, K), wp, Cout, K)
    OH = conv.conv_out(size, k)
    out = torch.zeros(imgs, OH, OH, Cout, device="cuda")
    conv.pattern_f(xs, out, wp, b, imgs, size, size, Cin, k, Cout, cabi.ACT_ELU)
    _close(conv.to_nchw(out), want)


@pytest.mark.parametrize("Cin,Cout,k,size", DEC)
def test_pattern_t_matches_conv_transpose2d(Cin, Cout, k, size):
    from big_dreamer_amd import _cabi as cabi, conv
    g = torch.Generator(device="cuda").manual_seed(2)
    imgs = 5
    x = torch.randn(imgs, Cin, size, size, device="cuda", generator=g)
    w = torch.randn(Cin, Cout, k, k, device="cuda", generator=g) * 0.1
    b = torch.randn(Cout, device="cuda", generator=g)
    want = Fnn.conv_transpose2d(*_cpu(x, w, b), stride=2)
    stored = w.permute(0, 2, 3, 1).contiguous()                       # (ci, ky, kx, co)
    packs = [torch.zeros(n, device="cuda") for n in conv.class_pack_floats(Cin, Cout, k)]
    conv.pack_classes(stored, packs, Cin, Cout, k)
    OH = conv.convT_out(size, k)
    out = torch.full((imgs, OH, OH, Cout), float("nan"), device="cuda")       # every pixel must be written
    xs = conv.to_nhwc(x)
    conv.pattern_t(xs, out, packs, b, imgs, size, size, Cin, k, Cout, OH, OH, cabi.ACT_NONE)
    _close(conv.to_nchw(out), want)
    # the four classes fused into one launch
    fused = torch.zeros(conv.fused_pack_floats(Cin, Cout, k), device="cuda")
    conv.pack_fused(stored, fused, Cin, Cout, k)
    out2 = torch.full((imgs, OH, OH, Cout), float("nan"), device="cuda")
    conv.pattern_t_fused(xs, out2, fused, b, imgs, size, size, Cin, k, Cout, OH, OH, cabi.ACT_NONE)
    _close(conv.to_nchw(out2), want)


@pytest.mark.parametrize("Cin,Cout,k,size", ENC[1:])
def test_conv2d_dgrad_is_pattern_t_and_convT_dgrad_is_pattern_f(Cin, Cout, k, size):
    from big_dreamer_amd import _cabi as cabi, conv
    g = torch.Generator(device="cuda").manual_seed(3)
    imgs = 4
    x = torch.randn(imgs, Cin, size, size, device="cuda", generator=g)
    w = torch.randn(Cout, Cin, k, k, device="cuda", generator=g) * 0.1
    OHc = (size - k) // 2 + 1
    gy = torch.randn(imgs, Cout, OHc, OHc, device="cuda", generator=g)
    xc = x.cpu().requires_grad_(True)
    (gx,) = torch.autograd.grad(Fnn.conv2d(xc, w.cpu(), None, stride=2), xc, gy.cpu())
    stored = w.permute(0, 2, 3, 1).contiguous()                       # (co, ky, kx, ci): outer = co = K side of the dgrad
    packs = [torch.zeros(n, device="cuda") for n in conv.class_pack_floats(Cout, Cin, k)]
    conv.pack_classes(stored, packs, Cout, Cin, k)
    OH = conv.conv_out(size, k)
    out = torch.full((imgs, size, size, Cin), float("nan"), device="cuda")
    gys = conv.to_nhwc(gy)
    conv.pattern_t(gys, out, packs, None, imgs, OH, OH, Cout, k, Cin, size, size, cabi.ACT_NONE)
    _close(conv.to_nchw(out), gx)
    fused = torch.zeros(conv.fused_pack_floats(Cout, Cin, k), device="cuda")
    conv.pack_fused(stored, fused, Cout, Cin, k)
    out2 = torch.full((imgs, size, size, Cin), float("nan"), device="cuda")
    conv.pattern_t_fused(gys, out2, fused, None, imgs, OH, OH, Cout, k, Cin, size, size, cabi.ACT_NONE)
    _close(conv.to_nchw(out2), gx)
    # transposed convolution with the same tensor as its (ci=Cout, co=Cin) weight: dgrad = strided conv of the gradient
    xt = torch.randn(imgs, Cout, OH, OH, device="cuda", generator=g)
    HTc = 2 * (OH - 1) + k
    gyt = torch.randn(imgs, Cin, HTc, HTc, device="cuda", generator=g)
    xtc = xt.cpu().requires_grad_(True)
    (gxt,) = torch.autograd.grad(Fnn.conv_transpose2d(xtc, w.cpu(), None, stride=2), xtc, gyt.cpu())   # w as (ci=Cout, co=Cin, k, k)
    Kt = k * k * Cin
    wp = torch.zeros(cabi.packed_floats(Cout, Kt), device="cuda")
    conv.pack_matrix(stored.view(Cout, Kt), wp, Cout, Kt)             # stored = (ci_T, ky, kx, co_T) for the transposed conv
    HT = conv.convT_out(OH, k)
    outt = torch.zeros(imgs, OH, OH, Cout, device="cuda")
    conv.pattern_f(conv.to_nhwc(gyt), outt, wp, None, imgs, HT, HT, Cin, k, Cout, cabi.ACT_NONE)
    _close(conv.to_nchw(outt), gxt)


@pytest.mark.parametrize("Cin,Cout,k,size", ENC + [(64, 32, 6, 30), (32, 3, 6, 64)])
def test_gathered_wgrad_matches_conv_weight_gradients(Cin, Cout, k, size):
    """Conv2d: dW, db from (dOut rows, input windows); ConvTranspose2d with the same geometry: (input rows, dOut windows)."""
    from big_dreamer_amd import conv
    g = torch.Generator(device="cuda").manual_seed(4)
    imgs = 6
    OH = conv.conv_out(size, k)
    x = torch.randn(imgs, Cin, size, size, device="cuda", generator=g)
    w = torch.randn(Cout, Cin, k, k, device="cuda", generator=g) * 0.1
    gy = torch.randn(imgs, Cout, OH, OH, device="cuda", generator=g)
    _, gw_ref, gb_ref = torch.ops.aten.convolution_backward(*_cpu(gy, x, w), [Cout], [2, 2], [0, 0], [1, 1], False, [0, 0], 1,
                                                            [False, True, True])
    dW = torch.zeros(Cout, k, k, Cin, device="cuda")
    db = torch.zeros(Cout, device="cuda")
    gys = conv.to_nhwc(gy).view(imgs * OH * OH, Cout)
    xs = conv.to_nhwc(x)                      # (descriptors hold raw pointers: keep the operands alive)
    conv.run_wgrad([conv.wgrad_desc(gys, Cout, xs, imgs, OH, OH, size, size, Cin, k, dW, db)])
    _close(dW.permute(0, 3, 1, 2), gw_ref, 1e-4)
    _close(db, gb_ref, 1e-4)
    # transposed conv (ci_T = Cout -> co_T = Cin) from OH x OH to size' = 2(OH-1)+k: weight gradient in (ci, ky, kx, co)
    HT = conv.convT_out(OH, k)
    xt = torch.randn(imgs, Cout, OH, OH, device="cuda", generator=g)
    gyt = torch.randn(imgs, Cin, HT, HT, device="cuda", generator=g)
    _, gwt_ref, _ = torch.ops.aten.convolution_backward(*_cpu(gyt, xt, w), None, [2, 2], [0, 0], [1, 1], True, [0, 0], 1,
                                                        [False, True, False])
    dWt = torch.zeros(Cout, k, k, Cin, device="cuda")
    xts = conv.to_nhwc(xt).view(imgs * OH * OH, Cout)
    gyts = conv.to_nhwc(gyt)
    conv.run_wgrad([conv.wgrad_desc(xts, Cout, gyts, imgs, OH, OH, HT, HT, Cin, k, dWt, None)])
    _close(dWt.permute(0, 3, 1, 2), gwt_ref, 1e-4)


@pytest.mark.parametrize("M,N,K,acc", [(2450, 1024, 3200, False), (37, 70, 52, False), (333, 200, 1024, True), (16, 64, 32, False),
                                       (50, 33, 45, True), (100, 130, 48, False), (200, 64, 16, True), (45, 64, 80, False)])
def test_plain_gemm_nt_matches_cpu_fp32(M, N, K, acc):
    """bd_gemm_nt (csrc/gemm.hip): C (+)= A B^T -- the decoder's K = 3200 dgrad GEMM (first case: configs[2] sizes, 256
    workgroups of 160 x 64) and ragged shapes / every rows-per-workgroup variant, against a CPU fp32 matmul."""
    from big_dreamer_amd import _cabi as cabi
    g = torch.Generator().manual_seed(M + N)
    A = torch.randn(M, K + 4, generator=g)[:, :K]              # lda > K
    B = torch.randn(N, K, generator=g) / K ** 0.5
    C0 = torch.randn(M, N + 3, generator=g)
    Ad = A.cuda()
    Ad_full = torch.zeros(M, K + 4, device="cuda")
    Ad_full[:, :K] = Ad
    # B at a float offset of 1 when K is odd-ish (a weight inside the flat parameter buffer): the scalar-load path
    Bbuf = torch.zeros(N * K + 1, device="cuda")
    Bd = Bbuf[K % 2:K % 2 + N * K].view(N, K)
    Bd.copy_(B)
    Cd = C0.cuda().contiguous()
    cabi.check(cabi.lib.bd_gemm_nt(Ad_full.data_ptr(), K + 4, Bd.data_ptr(), K, Cd.data_ptr(), N + 3, M, N, K, int(acc),
                                   cabi.stream()))
    torch.cuda.synchronize()
    ref = A.double() @ B.double().t() + (C0[:, :N].double() if acc else 0)
    got = Cd.cpu()
    assert torch.equal(got[:, N:], C0[:, N:]), "columns beyond N were written"
    err = float((got[:, :N].double() - ref).abs().max())
    assert err <= 2e-5 * float(ref.abs().max()) + 1e-6, err


@pytest.mark.parametrize("imgs,k,bias,act", [(5, 4, True, True), (13, 6, False, False), (1, 4, False, True), (700, 6, True, False)])
def test_thin_image_conv_forward_matches_cpu_fp32(imgs, k, bias, act):
    """bd_conv_thin_forward (csrc/conv.hip): stride-2 VALID convolution of the 3-channel 64 x 64 image into 32 channels --
    Conv2d(3 -> 32, k4) forward and, with k6, the dgrad of ConvTranspose2d(32 -> 3, k6) -- against F.conv2d on the CPU."""
    from big_dreamer_amd import _cabi as cabi, conv
    g = torch.Generator().manual_seed(imgs + k)
    x = torch.randn(imgs, 3, 64, 64, generator=g)
    w = torch.randn(32, 3, k, k, generator=g) * 0.2
    b = torch.randn(32, generator=g) if bias else None
    ref = Fnn.conv2d(x, w, b, stride=2)
    if act:
        ref = Fnn.elu(ref)
    OH = conv.conv_out(64, k)
    xs = conv.to_nhwc(x.cuda())
    # the stored layout (co, ky, kx, ci) at an odd float offset of a larger buffer, as in the flat parameter buffer
    wbuf = torch.zeros(32 * k * k * 3 + 8, device="cuda")
    ws = wbuf[4:4 + 32 * k * k * 3].view(32, k * k * 3)
    ws.copy_(w.permute(0, 2, 3, 1).reshape(32, -1))
    out = torch.full((imgs, OH, OH, 32), float("nan"), device="cuda")
    conv.thin_f(xs, out, ws, b.cuda() if bias else None, imgs, 64, 64, 3, k, cabi.ACT_ELU if act else cabi.ACT_NONE)
    torch.cuda.synchronize()
    _close(out.permute(0, 3, 1, 2), ref, 2e-5)


@pytest.mark.parametrize("Cin,Cout,k,size", ENC[1:])
def test_dgrad_with_fused_elu_backward(Cin, Cout, k, size):
    """BD_ACT_ELU_GRAD: the dgrad kernels multiply by ELU'(saved output of the layer below) in their epilogue -- the same
    numbers as the plain dgrad followed by bd_elu_backward, on the T (fused classes) and the F pattern."""
    from big_dreamer_amd import _cabi as cabi, conv
    g = torch.Generator(device="cuda").manual_seed(11)
    imgs = 5
    w = torch.randn(Cout, Cin, k, k, device="cuda", generator=g) * 0.1
    stored = w.permute(0, 2, 3, 1).contiguous()
    OH = conv.conv_out(size, k)
    gys = torch.randn(imgs, OH, OH, Cout, device="cuda", generator=g)
    saved = Fnn.elu(torch.randn(imgs, size, size, Cin, device="cuda", generator=g))       # ELU outputs, both signs
    fused = torch.zeros(conv.fused_pack_floats(Cout, Cin, k), device="cuda")
    conv.pack_fused(stored, fused, Cout, Cin, k)
    plain = torch.full((imgs, size, size, Cin), float("nan"), device="cuda")
    conv.pattern_t_fused(gys, plain, fused, None, imgs, OH, OH, Cout, k, Cin, size, size, cabi.ACT_NONE)
    cabi.check(cabi.lib.bd_elu_backward(cabi.ptr(plain), cabi.ptr(saved), plain.numel(), cabi.stream()))
    got = torch.full((imgs, size, size, Cin), float("nan"), device="cuda")
    conv.pattern_t_fused(gys, got, fused, None, imgs, OH, OH, Cout, k, Cin, size, size, cabi.ACT_ELU_GRAD, saved)
    torch.cuda.synchronize()
    assert torch.equal(got, plain)
    # F pattern (dgrad of a transposed convolution): gradient image (imgs, HT, HT, Cin) -> (imgs, OH, OH, Cout)
    HT = conv.convT_out(OH, k)
    Kt = k * k * Cin
    wp = torch.zeros(cabi.packed_floats(Cout, Kt), device="cuda")
    conv.pack_matrix(stored.view(Cout, Kt), wp, Cout, Kt)
    gyt = torch.randn(imgs, HT, HT, Cin, device="cuda", generator=g)
    saved2 = Fnn.elu(torch.randn(imgs, OH, OH, Cout, device="cuda", generator=g))
    plain2 = torch.full((imgs, OH, OH, Cout), float("nan"), device="cuda")
    conv.pattern_f(gyt, plain2, wp, None, imgs, HT, HT, Cin, k, Cout, cabi.ACT_NONE)
    cabi.check(cabi.lib.bd_elu_backward(cabi.ptr(plain2), cabi.ptr(saved2), plain2.numel(), cabi.stream()))
    got2 = torch.full((imgs, OH, OH, Cout), float("nan"), device="cuda")
    conv.pattern_f(gyt, got2, wp, None, imgs, HT, HT, Cin, k, Cout, cabi.ACT_ELU_GRAD, saved2)
    torch.cuda.synchronize()
    assert torch.equal(got2, plain2)


@pytest.mark.parametrize("imgs", [3, 300])
def test_thin_image_dgrad_with_fused_elu_backward(imgs):
    """The same for bd_conv_thin_forward (dgrad of ConvTranspose2d(32 -> 3, k6): 64 x 64 x 3 gradient -> 30 x 30 x 32)."""
    from big_dreamer_amd import _cabi as cabi, conv
    g = torch.Generator(device="cuda").manual_seed(imgs)
    k = 6
    x = torch.randn(imgs, 64, 64, 3, device="cuda", generator=g)
    ws = torch.randn(32, k * k * 3, device="cuda", generator=g) * 0.2
    OH = conv.conv_out(64, k)
    saved = Fnn.elu(torch.randn(imgs, OH, OH, 32, device="cuda", generator=g))
    plain = torch.full((imgs, OH, OH, 32), float("nan"), device="cuda")
    conv.thin_f(x, plain, ws, None, imgs, 64, 64, 3, k, cabi.ACT_NONE)
    cabi.check(cabi.lib.bd_elu_backward(cabi.ptr(plain), cabi.ptr(saved), plain.numel(), cabi.stream()))
    got = torch.full((imgs, OH, OH, 32), float("nan"), device="cuda")
    conv.thin_f(x, got, ws, None, imgs, 64, 64, 3, k, cabi.ACT_ELU_GRAD, saved)
    torch.cuda.synchronize()
    assert torch.equal(got, plain)
