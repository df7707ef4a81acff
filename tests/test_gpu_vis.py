"""ViS forward on the HIP path vs the CPU oracle / reference golden vectors.

Tolerances: fp32 (exact-fp32 MFMA) mode 1e-4 relative -- the tolerance BASELINE.json's
north_star states; bf16 perf mode 3e-2 relative (bf16 operands, fp32 accumulation)."""
import os

import numpy as np
import pytest
import torch

from gpu_util import rel_err

pytestmark = pytest.mark.gpu

from oracle import vis_oracle  # noqa: E402  (checker only)
from sequoia_pub_amd import _lib, synth  # noqa: E402
from sequoia_pub_amd.vis import ViS  # noqa: E402

TOL = {"fp32": 1e-4, "bf16": 3e-2}


def _tiny(golden_dir):
    z = np.load(os.path.join(golden_dir, "vis_tiny.npz"))
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w::")}
    return z, sd


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_vis_tiny_matches_reference_golden(golden_dir, mode):
    _lib.require_gpu()
    z, sd = _tiny(golden_dir)
    m = ViS(num_outputs=50, input_dim=128, depth=2, nheads=2, dimensions_f=64, dimensions_s=64, dimensions_c=64,
            device="cuda:0", compute_dtype=mode)
    m.load_state_dict(sd)
    m.to("cuda:0").eval()
    with torch.no_grad():
        out = m(torch.from_numpy(z["x"]).to(m.device)).cpu().numpy()
    err = rel_err(out, z["pred"])
    print(f"vis_tiny {mode}: rel err {err:.3e}")
    assert err < TOL[mode]


def test_vis_tiny_2d_literal_quirk(golden_dir):
    _lib.require_gpu()
    z, sd = _tiny(golden_dir)
    m = ViS(50, 128, 2, 2, 64, 64, 64, device="cuda:0")
    m.load_state_dict(sd)
    m.to("cuda:0")
    with torch.no_grad():
        out = m.forward_literal_2d(torch.from_numpy(z["x"][0]).cuda()).cpu().numpy()
    assert rel_err(out, z["pred_2d_literal"]) < 1e-4


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_vis_full_size_matches_reference_golden(golden_dir, mode):
    """D=1024, 6 layers, 16 heads, G=20820 (BASELINE config 2 model), B=2."""
    _lib.require_gpu()
    z = np.load(os.path.join(golden_dir, "vis_full.npz"))
    cfg = dict(num_outputs=20820, input_dim=1024, depth=6, nheads=16, dimensions_f=64, dimensions_s=64, dimensions_c=64)
    sd = vis_oracle.perturb_norm_params(vis_oracle.init_vis_state_dict(**cfg, seed=99), seed=5)
    m = ViS(**cfg, device="cuda:0", compute_dtype=mode)
    m.load_state_dict(sd)
    m.to("cuda:0")
    x = torch.from_numpy(synth.cluster_tokens(99, 2, 1024)).cuda()
    with torch.no_grad():
        out = m(x).cpu().numpy()
    err = rel_err(out, z["pred"])
    print(f"vis_full {mode}: rel err {err:.3e}")
    assert err < TOL[mode]


@pytest.mark.parametrize("B", [1, 5, 64])
def test_vis_batch_sizes_vs_oracle(B):
    """ragged batch sizes incl. BASELINE's B=64, D=1024 on a 2-layer model (oracle finishes in seconds)."""
    _lib.require_gpu()
    cfg = dict(num_outputs=1000, input_dim=1024, depth=2, nheads=16, dimensions_f=64, dimensions_s=64, dimensions_c=64)
    sd = vis_oracle.perturb_norm_params(vis_oracle.init_vis_state_dict(**cfg, seed=3), seed=4)
    m = ViS(**cfg, device="cuda:0")
    m.load_state_dict(sd)
    m.to("cuda:0")
    x = torch.from_numpy(synth.cluster_tokens(5, B, 1024))
    with torch.no_grad():
        ref = vis_oracle.vis_forward(sd, x).numpy()
        out = m(x.cuda()).cpu().numpy()
    assert rel_err(out, ref) < 1e-4


def test_combiner_in_the_f_projection_epilogue_matches_the_two_launches(monkeypatch):
    """Inference in bf16 at large batch (the spatial path: M = B * 100 >= 65 536 rows): the per-head combiner runs in the f
    projection's epilogue (gemm_p8.hip, GemmArgs::comb_w) and the summary branch ahead of it on the same stream.  Against the two
    launches (SQ_FWD_NO_FUSED_COMB=1) on the same input -- same bf16 operands, the 64-deep sums in another MFMA shape: equal up to
    the last bit of a bf16 now and then -- and, for the first slides, against the fp32 oracle at the bf16 tolerance.  A ragged last
    tile (B = 701 -> 70 100 rows = 273 tiles of 256 + 212 rows) and D = 1024, depth 2."""
    _lib.require_gpu()
    cfg = dict(num_outputs=1000, input_dim=1024, depth=2, nheads=16, dimensions_f=64, dimensions_s=64, dimensions_c=64)
    sd = vis_oracle.perturb_norm_params(vis_oracle.init_vis_state_dict(**cfg, seed=31), seed=32)
    m = ViS(**cfg, device="cuda:0", compute_dtype="bf16")
    m.load_state_dict(sd)
    m.to("cuda:0").eval()
    B = 701
    x = torch.from_numpy(synth.cluster_tokens(9, B, 1024))
    xd = x.cuda()
    with torch.no_grad():
        monkeypatch.delenv("SQ_FWD_NO_FUSED_COMB", raising=False)
        fused = m(xd).float().cpu()
        fused2 = m(xd).float().cpu()
        monkeypatch.setenv("SQ_FWD_NO_FUSED_COMB", "1")
        plain = m(xd).float().cpu()
        monkeypatch.delenv("SQ_FWD_NO_FUSED_COMB")
        ref = vis_oracle.vis_forward(sd, x[:6]).numpy()
    assert torch.isfinite(fused).all() and torch.equal(fused, fused2)
    d = rel_err(fused.numpy(), plain.numpy())
    print(f"combiner in the epilogue vs two launches: rel diff {d:.2e}, bit-equal outputs {float((fused == plain).float().mean()):.4f}; vs oracle {rel_err(fused[:6].numpy(), ref):.2e}")
    assert d < 2e-3
    assert rel_err(fused[:6].numpy(), ref) < TOL["bf16"] and rel_err(plain[:6].numpy(), ref) < TOL["bf16"]


def test_no_cpu_fallback():
    m = ViS(8, 64, 1, 1, 64, 64, 64, device="cpu")
    with pytest.raises(_lib.SequoiaHipError):
        m(torch.zeros(1, 100, 64))
