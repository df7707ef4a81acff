"""Split-bf16 ("bf16x3") mode: sq_linear_x3 (csrc/gemm_x3.hip) and the ResNet-50 embedder in SQ_DTYPE_BF16X3.

Every fp32 tensor travels as a bf16 hi plane and a bf16 lo plane, a product is three bf16 MFMAs
(a_hi.b_hi + a_hi.b_lo + a_lo.b_hi) with fp32 accumulation.  Two checks per shape:
  * against the EXACT value of that three-term formula in fp64 (only fp32 accumulation may differ: 2e-6), which pins the
    kernel -- loader, swizzle, fragment order, epilogue, hi/lo output split -- independently of the approximation;
  * against the true fp64 product of the unsplit operands: the approximation itself, <= 3e-5 (2^-17 per operand).
The network-level tests hold the mode to the reference's own tolerance (north_star: 1e-4 relative, labels bit-exact)."""
import ctypes
import os

import numpy as np
import pytest
import torch

from gpu_util import assert_allclose_rel, rel_err

pytestmark = pytest.mark.gpu

from oracle import resnet_oracle as ro  # noqa: E402  (checker only)
from sequoia_pub_amd import _lib, synth  # noqa: E402
from sequoia_pub_amd.resnet import resnet50  # noqa: E402


PLANE = {0: torch.bfloat16, 1: torch.float16}          # fmt 0: SQ_DTYPE_BF16X3 planes, fmt 1: SQ_DTYPE_F16X3 planes
WSCALE = {0: 1.0, 1: 2.0 ** 11}                        # fp16 planes: weight rows pre-scaled by a power of two (resnet.split_planes)


def split(t, fmt=0):
    hi = t.to(PLANE[fmt])
    lo = (t - hi.float()).to(PLANE[fmt])
    return hi, lo


def planes(t, fmt=0, dev="cuda"):
    """hi / lo planes in ONE allocation (as the embedder lays them out), returned as two views."""
    hi, lo = split(t, fmt)
    buf = torch.stack([hi, lo]).to(dev).contiguous()
    return buf[0], buf[1]


def run_x3(A, W, bias, res, act, f32_out=False, conv=None, M=None, fmt=0):
    Ah, Al = planes(A, fmt)
    Wh, Wl = planes(W * WSCALE[fmt], fmt)
    cs = torch.full((W.shape[0],), 1.0 / WSCALE[fmt], device="cuda") if fmt else None
    N, K = W.shape
    M = A.shape[0] if M is None else M
    bd = bias.cuda() if bias is not None else None
    rh = rl = None
    if res is not None:
        rh, rl = planes(res, fmt)
    out = torch.full((2, M, N), float("nan"), device="cuda", dtype=PLANE[fmt])
    o32 = torch.full((M, N), float("nan"), device="cuda") if f32_out else None
    geom = (ctypes.c_int * 9)(*conv) if conv else None
    lda = A.shape[-1] if conv is None else 0
    _lib.check(_lib.lib().sq_linear_x3(fmt, _lib.ptr(Ah), _lib.ptr(Al), lda, _lib.ptr(Wh), _lib.ptr(Wl), K, _lib.ptr(bd), _lib.ptr(cs),
                                       _lib.ptr(rh), _lib.ptr(rl), N, act,
                                       None if f32_out else _lib.ptr(out[0]), None if f32_out else _lib.ptr(out[1]), _lib.ptr(o32), N,
                                       M, N, K, geom, _lib.stream_ptr()))
    torch.cuda.synchronize()
    if f32_out:
        return o32.cpu().double()
    return out[0].cpu().double() + out[1].cpu().double()


def three_term(A, W, fmt=0):
    Ah, Al = (t.double() for t in split(A, fmt))
    Wh, Wl = (t.double() for t in split(W * WSCALE[fmt], fmt))
    return (Ah @ Wh.T + Ah @ Wl.T + Al @ Wh.T) / WSCALE[fmt]


def finish(y, bias, res, act, fmt=0):
    if bias is not None:
        y = y + bias.double()
    if res is not None:
        rh, rl = split(res, fmt)
        y = y + rh.double() + rl.double()
    return torch.relu(y) if act == 2 else y


SHAPES = [(256, 128, 64), (1000, 64, 256), (300, 520, 1032), (129, 72, 40), (4100, 256, 152), (513, 192, 2304), (37, 64, 64)]


# (approximation bound vs the true product, bound of one more output split): bf16 planes 2^-18 per operand; fp16 planes
# 2^-23 -- below the fp32 accumulation noise the first check already allows
TOL = {0: (3e-5, 1.6e-5), 1: (2e-6, 5e-7)}


@pytest.fixture(params=["auto", "rows256", "rows128"])
def block_shape(request):
    """Both block shapes of gemm_x3.hip on every problem: the 8-wave 256-row ping-pong kernel and the 4-wave 128-row,
    two-blocks-per-CU kernel (the launcher picks by K; sq_dbg_set key 7 forces one)."""
    lib = _lib.lib()
    lib.sq_dbg_set.argtypes = [ctypes.c_int, ctypes.c_int]
    lib.sq_dbg_set(7, {"auto": -1, "rows256": 0, "rows128": 1 << 30}[request.param])
    yield request.param
    lib.sq_dbg_set(7, -1)


@pytest.mark.parametrize("fmt", [0, 1])
@pytest.mark.parametrize("M,N,K", SHAPES)
@pytest.mark.parametrize("act,use_res", [(0, False), (2, True)])
def test_linear_x3_matches_three_term_formula_and_true_product(M, N, K, act, use_res, fmt, block_shape):
    _lib.require_gpu()
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K + act)
    A = torch.randn(M, K, generator=g) * torch.rand(M, 1, generator=g) * 3
    W = torch.randn(N, K, generator=g) * 0.1 + torch.arange(N)[:, None] * 1e-3       # asymmetric rows
    bias = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g) if use_res else None
    exact = finish(three_term(A, W, fmt), bias, res, act, fmt)
    true = finish(A.double() @ W.double().T, bias, res, act, fmt)
    out32 = run_x3(A, W, bias, res, act, f32_out=True, fmt=fmt)
    assert torch.isfinite(out32).all()
    assert rel_err(out32, exact) < 2e-6, rel_err(out32, exact)
    assert rel_err(out32, true) < TOL[fmt][0], rel_err(out32, true)
    out = run_x3(A, W, bias, res, act, fmt=fmt)            # hi / lo output planes: one more split on top
    assert rel_err(out, out32) < TOL[fmt][1], rel_err(out, out32)
    assert rel_err(out, true) < TOL[fmt][0] + TOL[fmt][1]


def _rel_per_element(out, exact):
    return float(((out - exact).abs() / exact.abs().clamp_min(1e-300)).max())


def test_f16x3_subnormal_planes_reach_the_matrix_pipe_bit_for_bit(block_shape):
    """Does gfx950's v_mfma_f32_32x32x16_f16 honour fp16 SUBNORMAL inputs?  csrc/x3_fmt.h's error statement ("absolute error
    <= 2^-25 below 2^-3") rests on it: below 2^-3 the lo plane of a value is an fp16 subnormal (|lo| < 2^-14).  Rows whose
    results are known bit for bit: a = 2^-10 + 2^-22 splits into hi = 2^-10 and lo = 2^-22 (a subnormal: 4 x 2^-24); against a
    weight of 2^-11 (a stored plane of exactly 1) the result must be a 2^-11 -- a flushed lo plane would return 2^-21 alone.
    Likewise a SUBNORMAL HI plane (a = 2^-20), the smallest subnormal (2^-24), a negative subnormal lo plane, and a weight
    whose lo plane is subnormal."""
    _lib.require_gpu()
    K, N, M = 64, 64, 128
    u = 2.0 ** -11                                        # run_x3 stores fp16 weight rows times 2^11 and undoes it in the epilogue
    A = torch.zeros(M, K, dtype=torch.float64)
    W = torch.zeros(N, K, dtype=torch.float64)
    A[0, 0] = 2.0 ** -10 + 2.0 ** -22                      # hi 2^-10, lo 2^-22 (subnormal)
    A[1, 0] = 2.0 ** -20                                   # hi subnormal, lo 0
    A[2, 0] = 2.0 ** -24                                   # the smallest fp16 subnormal
    A[3, 0] = 2.0 ** -3 - 2.0 ** -15                       # just under 2^-3: hi rounds to 2^-3, lo = -2^-15 (subnormal, negative)
    A[4, 0], A[4, 1] = 2.0 ** -12 + 2.0 ** -24, 2.0 ** -13 + 2.0 ** -24      # two terms, both lo planes the smallest subnormal
    W[:, 0] = u                                            # stored plane exactly 1: hi 1, lo 0
    W[:, 1] = 2 * u
    W[1, 0] = (1.0 + 2.0 ** -20) * u                       # a weight whose lo plane is subnormal: hi 1, lo 2^-20
    Af, Wf = A.float(), W.float()
    exact = three_term(Af, Wf, 1)                          # fp64 value of a_hi.w_hi + a_hi.w_lo + a_lo.w_hi on the planes as stored
    ah, al = split(Af, 1)
    assert float(al[0, 0]) == 2.0 ** -22 and float(ah[1, 0]) == 2.0 ** -20 and float(al[4, 1]) == 2.0 ** -24      # the planes ARE subnormal
    assert float(al[3, 0]) == -2.0 ** -15 and float(split(Wf / u, 1)[1][1, 0]) == 2.0 ** -20
    out = run_x3(Af, Wf, None, None, 0, f32_out=True, fmt=1)
    # every product here is exact in fp32 and every partial sum fits 24 bits: the result is known bit for bit
    assert float(out[0, 0]) == (2.0 ** -10 + 2.0 ** -22) * u, f"lo plane flushed? got {float(out[0, 0]):.10e}, the hi plane alone gives {2.0 ** -21:.10e}"
    assert float(out[1, 0]) == 2.0 ** -20 * u and float(out[2, 5]) == 2.0 ** -24 * u, "subnormal hi plane flushed"
    assert float(out[4, 7]) == ((2.0 ** -12 + 2.0 ** -24) + 2 * (2.0 ** -13 + 2.0 ** -24)) * u
    assert float(out[0, 1]) == (2.0 ** -10 + 2.0 ** -22 + 2.0 ** -30) * u and float(out[3, 0]) == (2.0 ** -3 - 2.0 ** -15) * u
    # ... with ONE exception the run of round 6 found: a product whose two factors are BOTH subnormal (a_hi = 2^-20 against the
    # weight's lo plane 2^-20: 2^-40; 2^-24 x 2^-20) does not reach the accumulator, while subnormal x normal does (rows above).
    # Such a product is < 2^-28 in the stored (scaled) units -- 2^-3 of the representation's own absolute error: harmless, but
    # pinned here so that x3_fmt.h's statement is what the hardware does.
    def sub(t):
        return (t != 0) & (t.abs() < 2.0 ** -14)
    Ah, Al = split(Af, 1)
    Wh, Wl = split(Wf / u, 1)
    dropped = torch.zeros_like(exact)
    for a_, w_ in ((Ah, Wh), (Ah, Wl), (Al, Wh)):
        dropped += (a_.double() * sub(a_)) @ (w_.double() * sub(w_)).T
    exact_hw = exact - dropped * u
    assert float(dropped.abs().max()) > 0 and float(dropped[1, 1]) == 2.0 ** -40
    bad = (out.float() != exact_hw.float()).nonzero()
    assert bad.numel() == 0, [(int(r), int(c), float(out[r, c]).hex(), float(exact_hw[r, c]).hex(), float(exact[r, c]).hex()) for r, c in bad[:12]]
    kept = int((out.float() == exact.float()).sum()), out.numel()
    print(f"fp16 MFMA, subnormal inputs: {kept[0]} of {kept[1]} results equal the full three-term value; subnormal x subnormal products are dropped "
          f"({int((dropped != 0).sum())} results differ by them)")


@pytest.mark.parametrize("lo_exp,hi_exp,wlo,whi,K", [(-14, -6, -6, 0, 256), (-14, -6, -6, 0, 64), (4, 15, -8, -3, 64), (-24, -14, -3, 0, 128)])
def test_f16x3_small_and_large_operands_per_element(lo_exp, hi_exp, wlo, whi, K, block_shape):
    """sq_linear_x3 on fp16 planes against the fp64 three-term formula PER ELEMENT (relative, not max-norm) where the O(1)
    operands of the other tests say nothing: activations in [2^-14, 2^-6) -- every lo plane entirely subnormal --, in
    [2^-24, 2^-14) -- the HI plane subnormal too --, and in [2^4, 2^15) near the top of fp16's range.  Operands are positive, so
    no cancellation hides a lost term: a flushed lo plane costs ~2^-12 relative (1.3e-4 ... 2.3e-4 on these draws, a flushed hi
    plane everything); the bound is fp32 accumulation.  Against the TRUE product: x3_fmt.h's statement -- 2^-22 relative per
    operand, or 2^-25 absolute per activation below 2^-3."""
    _lib.require_gpu()
    M, N = 300, 128
    g = torch.Generator().manual_seed(K + hi_exp * 7 + lo_exp)
    A = 2.0 ** (lo_exp + (hi_exp - lo_exp) * torch.rand(M, K, generator=g, dtype=torch.float64))
    W = 2.0 ** (wlo + (whi - wlo) * torch.rand(N, K, generator=g, dtype=torch.float64))
    A, W = A.float(), W.float()
    ah, al = split(A, 1)
    if hi_exp <= -6:
        assert float(al.float().abs().max()) < 2.0 ** -14, "the lo plane must be subnormal (or zero) throughout"
    exact = three_term(A, W, 1)
    true = A.double() @ W.double().T
    out = run_x3(A, W, None, None, 0, f32_out=True, fmt=1)
    assert torch.isfinite(out).all()
    e_exact = _rel_per_element(out, exact)
    slack = (out - true).abs() - (4e-6 * true + (2.0 ** -25 * W.double().sum(1))[None, :] * (1.0 if hi_exp <= -3 else 0.0))
    print(f"f16x3 operands in [2^{lo_exp}, 2^{hi_exp}), K={K}: worst per-element rel err vs the three-term formula {e_exact:.2e}, "
          f"vs the true product {_rel_per_element(out, true):.2e}")
    assert e_exact < 2e-6 * max(1.0, (K / 64) ** 0.5), e_exact
    assert float(slack.max()) <= 0, float(slack.max())


@pytest.mark.parametrize("fmt", [0, 1])
@pytest.mark.parametrize("Cin,Cout,k,stride,pad,H", [(64, 64, 3, 1, 1, 14), (128, 128, 3, 2, 1, 28), (256, 512, 1, 2, 0, 14), (32, 72, 3, 1, 1, 9)])
def test_conv_x3_implicit_gemm(Cin, Cout, k, stride, pad, H, fmt, block_shape):
    """The implicit-GEMM loader (taps gathered per K-tile, zero padding through the buffer descriptor) against
    torch.nn.functional.conv2d in fp64 on the joined operands, image borders inside a tile (n = 3 images)."""
    _lib.require_gpu()
    g = torch.Generator().manual_seed(Cin + Cout + k + H)
    n = 3
    x = torch.randn(n, H, H, Cin, generator=g)                                        # NHWC
    w = torch.randn(Cout, k, k, Cin, generator=g) * (1.0 / np.sqrt(k * k * Cin))     # [cout][kh][kw][cin]
    bias = torch.randn(Cout, generator=g)
    OH = (H + 2 * pad - k) // stride + 1
    xh, xl = split(x, fmt)
    wh, wl = split(w * WSCALE[fmt], fmt)
    xj, wj = xh.double() + xl.double(), (wh.double() + wl.double()) / WSCALE[fmt]
    ref = torch.nn.functional.conv2d(xj.permute(0, 3, 1, 2), wj.permute(0, 3, 1, 2), bias.double(), stride=stride, padding=pad)
    ref = torch.relu(ref).permute(0, 2, 3, 1).reshape(n * OH * OH, Cout)
    out = run_x3(x.reshape(-1, Cin), w.reshape(Cout, k * k * Cin), bias, None, 2, f32_out=True,
                 conv=(n, H, H, Cin, OH, OH, k, stride, pad), M=n * OH * OH, fmt=fmt)
    assert rel_err(out, ref) < TOL[fmt][0], rel_err(out, ref)


@pytest.mark.parametrize("fmt", [0, 1])
@pytest.mark.parametrize("halo", [-1, 0])
@pytest.mark.parametrize("Cin,Cout,H,n", [(128, 128, 28, 2), (256, 256, 14, 3), (64, 128, 7, 11), (32, 256, 31, 1), (96, 128, 5, 23),
                                           (64, 64, 56, 1), (64, 64, 40, 2), (32, 64, 14, 3), (64, 128, 33, 1)])
def test_conv_x3_halo_staged_3x3(Cin, Cout, H, n, halo, fmt):
    """conv_halo_x3.hip (input tile resident in LDS, all nine taps from one staged copy; taken for 3x3 / stride 1 / pad 1,
    maps <= 63 wide, N % 64 == 0: all four tile-width / staged-row instantiations) against conv2d in fp64 on the joined operands -- image borders and image boundaries
    inside a 256-pixel tile, a ragged last tile -- and the implicit-GEMM form (sq_dbg_set key 8 = 0) on the same problem."""
    _lib.require_gpu()
    lib = _lib.lib()
    lib.sq_dbg_set.argtypes = [ctypes.c_int, ctypes.c_int]
    g = torch.Generator().manual_seed(Cin + Cout + H + n)
    x = torch.randn(n, H, H, Cin, generator=g)
    w = torch.randn(Cout, 3, 3, Cin, generator=g) * (1.0 / np.sqrt(9 * Cin))
    bias = torch.randn(Cout, generator=g)
    xh, xl = split(x, fmt)
    wh, wl = split(w * WSCALE[fmt], fmt)
    xj, wj = xh.double() + xl.double(), (wh.double() + wl.double()) / WSCALE[fmt]
    ref = torch.nn.functional.conv2d(xj.permute(0, 3, 1, 2), wj.permute(0, 3, 1, 2), bias.double(), stride=1, padding=1)
    ref = torch.relu(ref).permute(0, 2, 3, 1).reshape(n * H * H, Cout)
    lib.sq_dbg_set(8, halo)
    try:
        out = run_x3(x.reshape(-1, Cin), w.reshape(Cout, 9 * Cin), bias, None, 2, f32_out=True, conv=(n, H, H, Cin, H, H, 3, 1, 1), M=n * H * H, fmt=fmt)
        planes_out = run_x3(x.reshape(-1, Cin), w.reshape(Cout, 9 * Cin), bias, None, 2, conv=(n, H, H, Cin, H, H, 3, 1, 1), M=n * H * H, fmt=fmt)
    finally:
        lib.sq_dbg_set(8, -1)
    assert torch.isfinite(out).all()
    assert rel_err(out, ref) < TOL[fmt][0], rel_err(out, ref)
    assert rel_err(planes_out, out) < TOL[fmt][1], rel_err(planes_out, out)


def _linear_x3_planes(fmt, A, W, cs, bias, res, M, N, K):
    """sq_linear_x3 on plane tensors [2, ., .] already on the device; returns the output planes [2, M, N]."""
    out = torch.full((2, M, N), float("nan"), device="cuda", dtype=PLANE[fmt])
    _lib.check(_lib.lib().sq_linear_x3(fmt, _lib.ptr(A[0]), _lib.ptr(A[1]), K, _lib.ptr(W[0]), _lib.ptr(W[1]), K, _lib.ptr(bias), _lib.ptr(cs),
                                       _lib.ptr(res[0]) if res is not None else None, _lib.ptr(res[1]) if res is not None else None, N, 2,
                                       _lib.ptr(out[0]), _lib.ptr(out[1]), None, N, M, N, K, None, _lib.stream_ptr()))
    return out


@pytest.mark.parametrize("fmt", [1, 0])
@pytest.mark.parametrize("C,N2,P", [(256, 256, 128), (256, 256, 1000), (256, 256, 3 * 196 + 5), (128, 128, 256), (128, 128, 1000), (128, 128, 2 * 784 + 77),
                                    (128, 256, 128), (128, 256, 1000), (128, 256, 784 + 300)])
def test_chain_x3w_is_bit_identical_to_the_two_launches(C, N2, P, fmt):
    """chain_x3w.hip (28 x 28 / 14 x 14 stages: expand 1x1 + identity + ReLU + the next block's reduce 1x1 in one launch, products
    transposed so that y stays in the wave that made it) against the two gemm_x3.hip launches it replaces, on the same planes:
    y and t1' must be equal bit for bit -- ragged last tiles, per-channel power-of-two weight scales, both plane formats."""
    _lib.require_gpu()
    lib = _lib.lib()
    vp = ctypes.c_void_p
    lib.sq_dbg_chain_x3w.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_longlong] + [vp] * 16 + [ctypes.c_int, vp]
    g = torch.Generator().manual_seed(C * 13 + N2 + P + fmt)
    N1 = 4 * C
    t2 = torch.relu(torch.randn(P, C, generator=g)) * torch.rand(P, 1, generator=g) * 2
    res = torch.relu(torch.randn(P, N1, generator=g)) * 1.5
    w3 = torch.randn(N1, C, generator=g) * (1.0 / np.sqrt(C)) + torch.arange(N1)[:, None] * 1e-4
    w1 = torch.randn(N2, N1, generator=g) * (1.0 / np.sqrt(N1)) + torch.arange(N2)[:, None] * 1e-4
    s3 = 2.0 ** (torch.arange(N1) % 5 + 7).float() if fmt else torch.ones(N1)
    s1 = 2.0 ** (torch.arange(N2) % 3 + 9).float() if fmt else torch.ones(N2)
    b3, b1 = torch.randn(N1, generator=g).cuda(), torch.randn(N2, generator=g).cuda()
    cs3, cs1 = (1.0 / s3).cuda(), (1.0 / s1).cuda()
    T2, R = torch.stack(split(t2, fmt)).cuda(), torch.stack(split(res, fmt)).cuda()
    # both weights in ONE allocation per plane (the fused launch takes one plane distance for both)
    wall = torch.cat([(w3 * s3[:, None]).reshape(-1), (w1 * s1[:, None]).reshape(-1)])
    Wp = torch.stack(split(wall, fmt)).cuda().contiguous()
    W3 = Wp[:, :N1 * C].view(2, N1, C)
    W1 = Wp[:, N1 * C:].view(2, N2, N1)
    assert W3[1].data_ptr() - W3[0].data_ptr() == W1[1].data_ptr() - W1[0].data_ptr()
    W3c, W1c = W3.contiguous(), W1.contiguous()                 # (sq_linear_x3 wants each weight's planes in their own allocation)
    y_ref = _linear_x3_planes(fmt, T2, W3c, cs3, b3, R, P, N1, C)
    t1_ref = _linear_x3_planes(fmt, y_ref, W1c, cs1, b1, None, P, N2, N1)
    y = torch.full((2, P, N1), float("nan"), device="cuda", dtype=PLANE[fmt])
    t1 = torch.full((2, P, N2), float("nan"), device="cuda", dtype=PLANE[fmt])
    rc = lib.sq_dbg_chain_x3w(fmt, C, N2, P, _lib.ptr(T2[0]), _lib.ptr(T2[1]), _lib.ptr(R[0]), _lib.ptr(R[1]), _lib.ptr(y[0]), _lib.ptr(y[1]),
                              _lib.ptr(t1[0]), _lib.ptr(t1[1]), _lib.ptr(W3[0]), _lib.ptr(W3[1]), _lib.ptr(W1[0]), _lib.ptr(W1[1]),
                              _lib.ptr(b3), _lib.ptr(cs3), _lib.ptr(b1), _lib.ptr(cs1), 0, _lib.stream_ptr())
    _lib.check(rc)
    torch.cuda.synchronize()
    yi, yr = y.view(torch.int16), y_ref.view(torch.int16)
    ti, tr = t1.view(torch.int16), t1_ref.view(torch.int16)
    assert torch.isfinite(y_ref.float()).all() and float(y_ref[0].float().abs().max()) > 0.5
    bad_y, bad_t = int((yi != yr).sum()), int((ti != tr).sum())
    assert bad_y == 0, (bad_y, yi.numel(), rel_err(y[0].float().cpu(), y_ref[0].float().cpu()))
    assert bad_t == 0, (bad_t, ti.numel(), rel_err(t1[0].float().cpu(), t1_ref[0].float().cpu()))


def _model(mode):
    sd = ro.init_resnet50_state_dict(seed=99, perturb_bn=True)
    m = resnet50(pretrained=False, compute_dtype=mode)
    full = m.state_dict()
    full.update(sd)
    m.load_state_dict(full)
    return m.to("cuda:0").eval(), sd


@pytest.mark.parametrize("mode,tol", [("bf16x3", 1e-4), ("f16x3", 1e-5)])
def test_resnet50_x3_features_match_reference_golden(golden_dir, mode, tol):
    """The reference's resnet50 (tests/golden/resnet50.npz) at the reference's tolerance (1e-4 relative); the fp16-plane
    mode is held to 1e-5 (measured: fp32-class)."""
    _lib.require_gpu()
    z = np.load(os.path.join(golden_dir, "resnet50.npz"))
    m, sd = _model(mode)
    f224 = m.extract_patches_u8(synth.patches_u8(0, n_patches=2, size=224)).cpu().numpy()
    f256 = m.extract_patches_u8(synth.patches_u8(1, n_patches=1, size=256)).cpu().numpy()
    e224, e256 = rel_err(f224, z["feat224"]), rel_err(f256, z["feat256"])
    print(f"resnet50 {mode}: rel err 224px {e224:.3e}  256px {e256:.3e}")
    assert e224 < tol and e256 < tol
    assert_allclose_rel(f224, z["feat224"], tol, f"{mode} features, allclose form")


@pytest.mark.parametrize("mode", ["bf16x3", "f16x3"])
def test_resnet50_x3_batch_and_stream_consistency(monkeypatch, mode):
    """A batch of 5 == five single-patch calls; two sub-batch chains in flight == one (same bits): tiles never mix patches'
    arithmetic and the plane layout does not depend on the sub-batch size."""
    _lib.require_gpu()
    m, sd = _model(mode)
    p = torch.from_numpy(synth.patches_u8(3, n_patches=5, size=224)).cuda()
    whole = m.extract_patches_u8(p)
    singles = torch.cat([m.extract_patches_u8(p[i:i + 1]) for i in range(5)])
    chunks = m.extract_patches_u8(p, sub_batch=2)
    torch.cuda.synchronize()
    assert torch.isfinite(whole).all()
    assert torch.equal(whole, singles) and torch.equal(whole, chunks)


@pytest.mark.parametrize("mode", ["f16x3", "bf16x3"])
def test_resnet50_x3_fused_chain_is_bit_identical(monkeypatch, mode):
    """chain_x3.hip (56 x 56 stage: [3x3 +] expand 1x1 + identity [or the downsample product] + ReLU + the next block's reduce
    1x1 in one launch, t2 and y handed over through LDS) walks K in the same order with the same MFMA sequence and epilogue
    arithmetic as the gemm_x3.hip / conv_halo_x3.hip launches it replaces: the features must not change by a bit.
    SQ_RESNET_NO_TAIL=1: the 3x3 as its own launch; SQ_RESNET_NO_CHAIN_DS=1: the downsample branch too; SQ_RESNET_NO_CHAIN=1:
    nothing fused in the 56 x 56 stage.  The downsample branches of layers 2-4 (and of layer 1 when the chain is off) ride in the
    expand launch as a second product (gemm_x3.hip dual form) unless SQ_RESNET_NO_DUAL=1: same requirement."""
    _lib.require_gpu()
    m, sd = _model(mode)
    p = torch.from_numpy(synth.patches_u8(5, n_patches=3, size=224)).cuda()
    p256 = torch.from_numpy(synth.patches_u8(6, n_patches=1, size=256)).cuda()          # 64 x 64 maps (256-px patches): the WIDE tail form (208-row planes, three-stage ring, tap-major K like the implicit GEMM these maps take unfused)
    outs = {}
    for tag, env in (("tail", {}), ("no_chainw", {"SQ_RESNET_NO_CHAINW": "1"}), ("no_stem_reduce", {"SQ_RESNET_NO_STEM_REDUCE": "1"}), ("chain", {"SQ_RESNET_NO_TAIL": "1"}), ("chain_no_ds", {"SQ_RESNET_NO_TAIL": "1", "SQ_RESNET_NO_CHAIN_DS": "1"}),
                     ("no_dual", {"SQ_RESNET_NO_DUAL": "1"}), ("dual_everywhere", {"SQ_RESNET_NO_CHAIN": "1"}),
                     ("plain", {"SQ_RESNET_NO_CHAIN": "1", "SQ_RESNET_NO_DUAL": "1", "SQ_RESNET_NO_CHAINW": "1", "SQ_RESNET_NO_STEM_REDUCE": "1"})):
        for k in ("SQ_RESNET_NO_TAIL", "SQ_RESNET_NO_CHAIN_DS", "SQ_RESNET_NO_CHAIN", "SQ_RESNET_NO_DUAL", "SQ_RESNET_NO_CHAINW", "SQ_RESNET_NO_STEM_REDUCE"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        outs[tag] = (m.extract_patches_u8(p), m.extract_patches_u8(p256))
    torch.cuda.synchronize()
    assert torch.isfinite(outs["tail"][0]).all()
    for tag in ("tail", "no_chainw", "no_stem_reduce", "chain", "chain_no_ds", "no_dual", "dual_everywhere"):
        assert torch.equal(outs[tag][0], outs["plain"][0]) and torch.equal(outs[tag][1], outs["plain"][1]), tag


@pytest.mark.parametrize("mode", ["f16x3"])
def test_tail_tile_walk_does_not_change_a_bit(monkeypatch, mode):
    """The 56 x 56 tails walk their tiles in chunks of 64 per XCD (chain_x3.hip: halo rows shared under one L2).  The walk only
    renames tiles: 11 patches = 539 tiles of 64 pixels (512 walked in chunks, the ragged 27 in plain order), 6 patches of 256
    px = 384 tiles (chunks of 16: walked; 64: plain order) -- chunked, one run per XCD and plain order give the same bits."""
    _lib.require_gpu()
    m, sd = _model(mode)
    p = torch.from_numpy(synth.patches_u8(11, n_patches=11, size=224)).cuda()
    p256 = torch.from_numpy(synth.patches_u8(12, n_patches=6, size=256)).cuda()
    outs = {}
    for walk in ("0", "64", "16", "1"):
        monkeypatch.setenv("SQ_X3_TAIL_XCD_WALK", walk)
        outs[walk] = (m.extract_patches_u8(p), m.extract_patches_u8(p256))
    monkeypatch.delenv("SQ_X3_TAIL_XCD_WALK")
    outs["default"] = (m.extract_patches_u8(p), m.extract_patches_u8(p256))
    torch.cuda.synchronize()
    assert torch.isfinite(outs["0"][0]).all() and torch.isfinite(outs["0"][1]).all()
    for k in ("64", "16", "1", "default"):
        assert torch.equal(outs[k][0], outs["0"][0]) and torch.equal(outs[k][1], outs["0"][1]), k
