"""sq_linear (MFMA NT GEMM engine) vs torch fp32/fp64 on the host: asymmetric operands,
ragged M/N/K tails, bias / residual / activation epilogues, both MFMA paths."""
import ctypes

import numpy as np
import pytest
import torch

from gpu_util import rel_err, to_bf16_f32

pytestmark = pytest.mark.gpu

from sequoia_pub_amd import _lib  # noqa: E402

SHAPES = [(128, 128, 64), (200, 136, 96), (64, 20820 // 10, 128), (6400, 1024, 1024), (37, 64, 64),
          (1000, 64, 256), (300, 520, 1032), (129, 65, 40)]


def run_linear(dtype, A, W, bias, res, act, out_bf16=False, ws_bytes=0):
    dev = "cuda"
    M, K = A.shape
    N = W.shape[0]
    tdt = torch.bfloat16 if dtype == _lib.SQ_BF16 else torch.float32
    Ad, Wd = A.to(dev, tdt).contiguous(), W.to(dev, tdt).contiguous()
    bd = bias.to(dev) if bias is not None else None
    rd = res.to(dev) if res is not None else None
    C = torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16 if out_bf16 else torch.float32)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev) if ws_bytes else None
    _lib.check(_lib.lib().sq_linear(dtype, _lib.ptr(Ad), K, _lib.ptr(Wd), K, _lib.ptr(bd), _lib.ptr(rd), N, _lib.SQ_F32, act,
                                    _lib.ptr(C), _lib.SQ_BF16 if out_bf16 else _lib.SQ_F32, N, M, N, K,
                                    _lib.ptr(ws), ws_bytes, _lib.stream_ptr()))
    torch.cuda.synchronize()
    return C.float().cpu()


def ref_linear(A, W, bias, res, act):
    y = A.double() @ W.double().T
    if bias is not None:
        y = y + bias.double()
    if res is not None:
        y = y + res.double()
    if act == 1:
        y = torch.nn.functional.gelu(y)
    elif act == 2:
        y = torch.relu(y)
    return y


@pytest.mark.parametrize("M,N,K", SHAPES)
@pytest.mark.parametrize("act", [0, 1, 2])
def test_linear_fp32_exact_mfma(M, N, K, act):
    _lib.require_gpu()
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K + act)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) * 0.1 + torch.arange(N)[:, None] * 1e-3     # asymmetric rows
    bias = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g) if act != 1 else None
    out = run_linear(_lib.SQ_F32, A, W, bias, res, act)
    ref = ref_linear(A, W, bias, res, act)
    assert torch.isfinite(out).all()
    assert rel_err(out, ref) < 2e-6, rel_err(out, ref)


@pytest.mark.parametrize("M,N,K", [s for s in SHAPES if s[2] % 8 == 0])
def test_linear_bf16_mfma(M, N, K):
    _lib.require_gpu()
    g = torch.Generator().manual_seed(M + N + K)
    A = to_bf16_f32(torch.randn(M, K, generator=g))
    W = to_bf16_f32(torch.randn(N, K, generator=g) * 0.1)
    bias = torch.randn(N, generator=g)
    out = run_linear(_lib.SQ_BF16, A, W, bias, None, 0)
    ref = ref_linear(A, W, bias, None, 0)          # operands already bf16-exact -> only fp32 accumulation differs
    assert rel_err(out, ref) < 2e-6, rel_err(out, ref)
    out16 = run_linear(_lib.SQ_BF16, A, W, bias, None, 2, out_bf16=True)
    assert rel_err(out16, torch.relu(ref)) < 5e-3


def test_fast_gelu_epilogue_decays_for_large_negative_pre_activations():
    """bf16 mode's GELU epilogue (sq_common.h sq_gelu<true>: erf as a polynomial on an argument clamped to +-3) for pre-activations
    in [-30, -4.3], far left of the clamp: the exact value lies in (-4.7e-5, 0]; the clamped polynomial alone would return
    -1.1e-5 |x|.  fp32 outputs, so nothing hides behind a bf16 rounding."""
    _lib.require_gpu()
    N, K = 256, 64
    pre = -(4.3 + (30 - 4.3) * torch.arange(N) / (N - 1))
    pre = pre.to(torch.bfloat16).float()                      # exact in bf16: the product below reproduces it exactly
    A = torch.zeros(128, K)
    A[:, 0] = 1.0
    W = torch.zeros(N, K)
    W[:, 0] = pre
    out = run_linear(_lib.SQ_BF16, A, W, None, None, 1)
    ref = torch.nn.functional.gelu(pre.double()).float().expand(128, N)
    err = float((out - ref).abs().max())
    print(f"fast GELU, x in [-30, -4.3]: max abs error {err:.2e}, most negative output {float(out.min()):.2e}")
    assert err < 6e-5 and float(out.min()) > -6e-5
    # and around the clamp / in the bulk the polynomial is within its stated bound
    pre2 = torch.linspace(-4.2, 4.2, N).to(torch.bfloat16).float()
    W[:, 0] = pre2
    out2 = run_linear(_lib.SQ_BF16, A, W, None, None, 1)
    ref2 = torch.nn.functional.gelu(pre2.double()).float().expand(128, N)
    assert float((out2 - ref2).abs().max()) < 1e-4


@pytest.mark.parametrize("M,N,K", [(64, 1024, 1024), (1024, 1024, 6400), (64, 128, 20824), (200, 520, 4104)])
@pytest.mark.parametrize("dtype", [_lib.SQ_F32, _lib.SQ_BF16])
def test_split_k_path(M, N, K, dtype):
    """skinny problems with a workspace take the deterministic split-K path (same epilogue)."""
    _lib.require_gpu()
    g = torch.Generator().manual_seed(M + N + K)
    A = to_bf16_f32(torch.randn(M, K, generator=g))
    W = to_bf16_f32(torch.randn(N, K, generator=g) * 0.1)
    bias = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g)
    out = run_linear(dtype, A, W, bias, res, 1, ws_bytes=64 << 20)
    out2 = run_linear(dtype, A, W, bias, res, 1, ws_bytes=64 << 20)
    ref = ref_linear(A, W, bias, res, 1)
    assert rel_err(out, ref) < 1e-5, rel_err(out, ref)
    assert torch.equal(out, out2)                      # fixed reduction order -> bitwise repeatable


@pytest.mark.parametrize("T,NO,NI", [(6400, 1024, 1024), (64, 1024, 2048), (200, 20820, 128), (333, 64, 64), (1000, 136, 264)])
@pytest.mark.parametrize("dtype", [_lib.SQ_F32, _lib.SQ_BF16])
@pytest.mark.parametrize("ws_bytes", [0, 64 << 20])
def test_weight_grad_tn_kernel(T, NO, NI, dtype, ws_bytes):
    """dW = dY^T . X straight from token-major operands (TN kernel), ragged token counts and output rows."""
    _lib.require_gpu()
    g = torch.Generator().manual_seed(T + NO + NI)
    dY = to_bf16_f32(torch.randn(T, NO, generator=g))
    X = to_bf16_f32(torch.randn(T, NI, generator=g) + torch.arange(NI)[None, :] * 1e-3)      # asymmetric columns
    tdt = torch.bfloat16 if dtype == _lib.SQ_BF16 else torch.float32
    ldy = (NO + 7) // 8 * 8
    dYd = torch.zeros(T, ldy, dtype=tdt, device="cuda")
    dYd[:, :NO] = dY.to(tdt)
    Xd = X.to("cuda", tdt).contiguous()
    dW = torch.full((NO, NI), float("nan"), device="cuda")
    db = torch.full((NO,), float("nan"), device="cuda")
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda") if ws_bytes else None
    _lib.check(_lib.lib().sq_linear_weight_grad(dtype, _lib.ptr(dYd), ldy, _lib.ptr(Xd), NI, _lib.ptr(dW), NI, _lib.ptr(db), NO, NI, T,
                                                _lib.ptr(ws), ws_bytes, _lib.stream_ptr()))
    torch.cuda.synchronize()
    ref = dY.double().T @ X.double()
    assert rel_err(dW.cpu(), ref) < 1e-5, rel_err(dW.cpu(), ref)
    # the bias gradient rides along in the same launch
    assert rel_err(db.cpu(), dY.double().sum(0)) < 1e-5, rel_err(db.cpu(), dY.double().sum(0))
    dW2 = torch.full((NO, NI), float("nan"), device="cuda")
    _lib.check(_lib.lib().sq_linear_weight_grad(dtype, _lib.ptr(dYd), ldy, _lib.ptr(Xd), NI, _lib.ptr(dW2), NI, None, NO, NI, T,
                                                _lib.ptr(ws), ws_bytes, _lib.stream_ptr()))
    assert rel_err(dW2.cpu(), dW.cpu()) < 1e-6          # (the K-slice count, hence the summation order, may differ)


@pytest.mark.parametrize("T,NO,NI,members", [(6400, 1024, 1024, 4), (6390, 1024, 1024, 1), (777, 256, 384, 3), (1500, 512, 2048, 2), (520, 128, 256, 4)])
@pytest.mark.parametrize("with_bias", [True, False])
def test_weight_grad_group_ring_form(T, NO, NI, members, with_bias):
    """Up to four same-shape weight gradients in one launch (sq_linear_weight_grad_group): the four-stage ring form of the TN
    kernel (bf16, extents multiples of 128: 8 waves per tile with an in-block K split, XCD-contiguous tile walk, bias gradient
    inside the tiles) against fp64, against the two-buffer kernel, ragged token counts, and bitwise repeatable."""
    _lib.require_gpu()
    lib = _lib.lib()
    g = torch.Generator().manual_seed(T + NO + NI + members)
    dYs = [to_bf16_f32(torch.randn(T, NO, generator=g) + 0.05 * i) for i in range(members)]
    Xs = [to_bf16_f32(torch.randn(T, NI, generator=g) + torch.arange(NI)[None, :] * 1e-3) for i in range(members)]
    dYd = [t.to("cuda", torch.bfloat16).contiguous() for t in dYs]
    Xd = [t.to("cuda", torch.bfloat16).contiguous() for t in Xs]

    def run(ring):
        lib.sq_dbg_set(15, ring)
        try:
            dW = [torch.full((NO, NI), float("nan"), device="cuda") for _ in range(members)]
            db = [torch.full((NO,), float("nan"), device="cuda") for _ in range(members)]
            arr = lambda ts: (ctypes.c_void_p * members)(*[t.data_ptr() for t in ts])
            _lib.check(lib.sq_linear_weight_grad_group(_lib.SQ_BF16, members, arr(dYd), arr(Xd), arr(dW), arr(db) if with_bias else None,
                                                       NO, NI, NI, NO, NI, T, _lib.stream_ptr()))
            torch.cuda.synchronize()
        finally:
            lib.sq_dbg_set(15, -1)
        return [w.cpu() for w in dW], [b.cpu() for b in db]

    dW1, db1 = run(1)
    dW1b, db1b = run(1)
    dW0, db0 = run(0)
    for i in range(members):
        ref = dYs[i].double().T @ Xs[i].double()
        assert rel_err(dW1[i], ref) < 1e-5, (i, rel_err(dW1[i], ref))
        assert rel_err(dW0[i], ref) < 1e-5, (i, rel_err(dW0[i], ref))
        assert torch.equal(dW1[i], dW1b[i])
        if with_bias:
            rb = dYs[i].double().sum(0)
            assert rel_err(db1[i], rb) < 1e-5, (i, rel_err(db1[i], rb))
            assert rel_err(db0[i], rb) < 1e-5
            assert torch.equal(db1[i], db1b[i])


def test_bad_arguments_fail_loudly():
    _lib.require_gpu()
    A = torch.zeros(4, 6, device="cuda")
    rc = _lib.lib().sq_linear(_lib.SQ_F32, _lib.ptr(A), 6, _lib.ptr(A), 6, None, None, 0, 0, 0, _lib.ptr(A), 0, 4, 4, 4, 6,
                              None, 0, _lib.stream_ptr())
    assert rc != 0 and b"multiple" in _lib.lib().sq_last_error()
    # grouped weight gradient: five members, a null member, a misaligned member
    X = torch.zeros(256, 128, device="cuda", dtype=torch.bfloat16)
    dW = torch.zeros(128, 128, device="cuda")
    arr = lambda ps: (ctypes.c_void_p * len(ps))(*ps)
    lib = _lib.lib()
    rc = lib.sq_linear_weight_grad_group(_lib.SQ_BF16, 5, arr([X.data_ptr()] * 5), arr([X.data_ptr()] * 5), arr([dW.data_ptr()] * 5), None,
                                         128, 128, 128, 128, 128, 256, _lib.stream_ptr())
    assert rc != 0 and b"members" in lib.sq_last_error()
    rc = lib.sq_linear_weight_grad_group(_lib.SQ_BF16, 2, arr([X.data_ptr(), 0]), arr([X.data_ptr()] * 2), arr([dW.data_ptr()] * 2), None,
                                         128, 128, 128, 128, 128, 256, _lib.stream_ptr())
    assert rc != 0 and b"member 1" in lib.sq_last_error()
    rc = lib.sq_linear_weight_grad_group(_lib.SQ_BF16, 2, arr([X.data_ptr(), X.data_ptr() + 2]), arr([X.data_ptr()] * 2), arr([dW.data_ptr()] * 2), None,
                                         128, 128, 128, 128, 128, 256, _lib.stream_ptr())
    assert rc != 0 and b"misaligned" in lib.sq_last_error()


@pytest.mark.parametrize("M,N,K,act,out", [(256 * 9 + 37, 512, 320, 2, "bf16"), (700, 256 + 64, 1024, 1, "bf16"), (3000, 768, 72, 0, "f32"),
                                           (256 * 40, 1024, 4096, 0, "f32"), (256 * 70 + 37, 1024, 320, 2, "bf16"), (256 * 33, 2048, 1088, 1, "f32")])
@pytest.mark.parametrize("sched,bn", [(0, 256), (1, 256), (1, 128)])
def test_eight_phase_256x256x64_variant(M, N, K, act, out, sched, bn):
    """gemm_p8.hip (256 x 256 x 64 tile, two buffers of four half-tiles, eight phases per pair of K-tiles, the two wave rows one
    barrier apart, 16x16x32 MFMAs), forced through the experiment knob: ragged M / N / K (an odd number of K-tiles, a K-tile
    with a single 16-byte chunk), bias, bf16 residual, ReLU / GELU, bf16 / fp32 output against the fp32 product of the same
    bf16 operands and against the default tile -- BIT-EQUAL, although the MFMA shape differs (16x16x32 against 32x32x16: on gfx950 both evidently
    accumulate K in the same order); run
    three times -- a schedule race would show as run-to-run differences.  Persistent form (one block per CU walking its tiles, the
    next tile's operands requested before the current tile's results are stored): the cases with more than 256 tiles."""
    _lib.require_gpu()
    lib = _lib.lib()
    lib.sq_dbg_set.argtypes = [ctypes.c_int, ctypes.c_int]
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).cuda().bfloat16()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).cuda().bfloat16()
    bias = torch.randn(N, generator=g).cuda()
    res = torch.randn(M, N, generator=g).cuda().bfloat16()
    odt, ocode = (torch.bfloat16, _lib.SQ_BF16) if out == "bf16" else (torch.float32, _lib.SQ_F32)
    outs = []
    lib.sq_dbg_set(10, sched)            # 0: one block per tile; 1: persistent blocks (when there are more tiles than CUs: the last two cases)
    lib.sq_dbg_set(13, bn)               # tile width: 256 (waves 2 x 4) or 128 (waves 4 x 2, one LDS-DMA instruction per B half-tile)
    try:
        for tile in (22, 88, 88, 88):
            lib.sq_dbg_set(0, tile)
            C = torch.full((M, N), float("nan"), device="cuda", dtype=odt)
            _lib.check(lib.sq_linear(_lib.SQ_BF16, _lib.ptr(A), K, _lib.ptr(W), K, _lib.ptr(bias), _lib.ptr(res), N, _lib.SQ_BF16, act,
                                     _lib.ptr(C), ocode, N, M, N, K, None, 0, _lib.stream_ptr()))
            torch.cuda.synchronize()
            outs.append(C.float())
    finally:
        lib.sq_dbg_set(0, 0)
        lib.sq_dbg_set(10, -1)
        lib.sq_dbg_set(13, -1)
    pre = A.float() @ W.float().T + bias + res.float()
    ref = torch.relu(pre) if act == 2 else torch.nn.functional.gelu(pre) if act == 1 else pre
    assert torch.isfinite(outs[1]).all()
    tol = 1e-2 if out == "bf16" else 2e-3 if act == 1 else 1e-5
    assert rel_err(outs[1].cpu(), ref.cpu()) < tol, rel_err(outs[1].cpu(), ref.cpu())
    assert rel_err(outs[1].cpu(), outs[0].cpu()) < tol
    assert torch.equal(outs[1], outs[2]) and torch.equal(outs[1], outs[3])
    same = bool(torch.equal(outs[0], outs[1]))
    print(f"gemm_p8 {M}x{N}x{K} act {act} {out} (sched {sched}, width {bn}): bit-equal to the default tile: {same}")
    assert same, "gemm_p8 left the bits of the default tile (the 16x16x32 and 32x32x16 bf16 MFMAs sum K identically on gfx950: measured equal in every case)"


@pytest.mark.parametrize("M,N,K", [(70000, 256, 1024), (65536 + 77, 128, 576)])
def test_ring_variant_matches_default_tile(M, N, K):
    """The three-stage 256 x 128 ring kernel (gemm_ring.hip) walks K in the same order with the same MFMA as the
    128 x 128 kernel: identical bits, including a ragged last M tile, with bias + bf16 residual + ReLU."""
    _lib.require_gpu()
    lib = _lib.lib()
    lib.sq_dbg_set.argtypes = [ctypes.c_int, ctypes.c_int]
    g = torch.Generator().manual_seed(M + K)
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).cuda()
    b = torch.randn(N, generator=g).cuda()
    R = torch.randn(M, N, generator=g).to(torch.bfloat16).cuda()
    ws = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    outs = []
    for tile in (22, 33):
        lib.sq_dbg_set(0, tile)
        C = torch.empty(M, N, device="cuda")
        _lib.check(lib.sq_linear(_lib.SQ_BF16, _lib.ptr(A), K, _lib.ptr(W), K, _lib.ptr(b), _lib.ptr(R), N, _lib.SQ_BF16, 2, _lib.ptr(C), 0, N, M, N, K,
                                 _lib.ptr(ws), ws.numel(), _lib.stream_ptr()))
        outs.append(C)
    lib.sq_dbg_set(0, 0)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1])
    ref = torch.relu(A.float() @ W.float().T + b + R.float())
    assert float((outs[1] - ref).abs().max() / ref.abs().max()) < 2e-2
