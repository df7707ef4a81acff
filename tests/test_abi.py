"""CPU-side checks: the C-ABI library builds, loads and exports every symbol that
include/sequoia_hip.h declares; host-side layout logic; no compute calls."""
import ctypes
import os
import re

import pytest
import torch

from sequoia_pub_amd import _lib
from sequoia_pub_amd.vis import ViS, tensor_map, vis_layout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "sequoia_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sq_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = _lib.lib()
    names = declared_symbols()
    assert len(names) >= 6
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/sequoia_hip.h but not exported"


def test_layout_is_aligned_and_dense():
    cfg = _lib.VisConfig(1024, 6, 16, 20820, 100)
    lay = vis_layout(cfg)
    tm = tensor_map(cfg, lay)
    assert len(tm) == 1013                                    # 1013 tensors at full size
    n = sum(int(torch.tensor(s).prod()) for _, s in tm.values())
    assert n == 53_758_292 or n > 53_000_000
    spans = sorted((off, off + int(torch.tensor(s).prod())) for off, s in tm.values())
    for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
        assert a1 <= b0                                       # no overlap
    assert lay.total >= spans[-1][1] and lay.total % 8 == 0
    for f in ("f_w", "s_w", "c_w", "proj_w", "ff1_w", "ff2_w"):
        assert getattr(lay.layer[0], f) % 8 == 0              # 16-byte aligned in bf16 as well


def test_gradient_buckets_tile_the_flat_buffer():
    """The all-reduce buckets (completion order: head, layers last to first) cover every gradient exactly once."""
    cfg = _lib.VisConfig(1024, 6, 16, 20820, 100)
    lay = vis_layout(cfg)
    cap = cfg.depth + 1
    lo, hi = (ctypes.c_int64 * cap)(), (ctypes.c_int64 * cap)()
    assert _lib.lib().sq_vis_grad_buckets(ctypes.byref(cfg), lo, hi, cap) == cap
    spans = sorted((lo[i], hi[i]) for i in range(cap))
    assert spans[0][0] == 0 and spans[-1][1] == lay.total
    for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
        assert a1 == b0
    assert (lo[0], hi[0]) == (lay.head_ln_g, lay.total)                    # head first
    assert lo[cap - 1] == 0 and hi[cap - 1] == lay.layer[1].f_w            # layer 0 (+ pos_emb1D) last
    assert _lib.lib().sq_vis_grad_buckets(ctypes.byref(cfg), lo, hi, 3) < 0


def test_bad_config_reports_error():
    cfg = _lib.VisConfig(100, 6, 16, 10, 100)                 # D not a multiple of 64
    lay = _lib.VisLayout()
    rc = _lib.lib().sq_vis_layout_init(ctypes.byref(cfg), ctypes.byref(lay))
    assert rc != 0 and b"input_dim" in _lib.lib().sq_last_error()


def test_model_without_gpu_raises_not_falls_back():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    m = ViS(8, 64, 1, 1, 64, 64, 64, device="cpu")
    with pytest.raises(_lib.SequoiaHipError):
        m(torch.zeros(1, 100, 64))


def test_split_planes_layout_matches_the_header():
    """include/sequoia_hip.h, split modes: weights = hi plane [w_total] then lo plane [w_total]; every convolution behind the
    stem K-tile-major inside its block -- element (n, k) at w_off + ((k // 32) * cout + n) * 32 + k % 32 -- the stem row-major;
    bias = [b_total] biases then [b_total] per-channel factors that undo the power-of-two pre-scaling (host logic, no GPU)."""
    import torch
    from sequoia_pub_amd import resnet as rn
    lay = rn.resnet50_layout()
    g = torch.Generator().manual_seed(0)
    w = torch.randn(lay.w_total, generator=g) * 0.05
    b = torch.randn(lay.b_total, generator=g)
    planes, bs = rn.split_planes(w, b, _lib.SQ_F16X3, lay)
    assert planes.numel() == 2 * lay.w_total and bs.numel() == 2 * lay.b_total
    hi = planes[:lay.w_total].view(torch.float16).float()
    lo = planes[lay.w_total:].view(torch.float16).float()
    scale = bs[lay.b_total:]
    assert torch.equal(bs[:lay.b_total], b)
    for i in (0, 1, 2, 4, 14, 52):
        d = lay.conv[i]
        K = d.k_padded
        rows = w[d.w_off:d.w_off + d.cout * K].view(d.cout, K)
        s = 1.0 / scale[d.b_off:d.b_off + d.cout]                  # the power of two each row was multiplied by
        assert torch.equal(torch.exp2(torch.round(torch.log2(s))), s)
        amax = (rows * s[:, None]).abs().amax(dim=1)
        assert bool(((amax > 128) & (amax <= 256)).all())
        got = (hi + lo)[d.w_off:d.w_off + d.cout * K]
        if i == 0:
            back = got.view(d.cout, K)                             # stem: [64][152] rows
        else:
            assert K % 32 == 0
            back = got.view(K // 32, d.cout, 32).permute(1, 0, 2).reshape(d.cout, K)
        # hi + lo carries 22 bits of the scaled weight
        assert torch.allclose(back / s[:, None], rows, rtol=3e-7, atol=1e-9)
