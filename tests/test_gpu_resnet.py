"""HIP ResNet-50 forward_extract vs reference golden features (tests/golden/resnet50.npz) and the oracle.
Tolerances: fp32 (exact-fp32 MFMA) 1e-4 relative; bf16 activations/weights 5e-2 relative."""
import os

import numpy as np
import pytest
import torch

from gpu_util import rel_err

pytestmark = pytest.mark.gpu

from oracle import resnet_oracle as ro  # noqa: E402  (checker only)
from sequoia_pub_amd import _lib, synth  # noqa: E402
from sequoia_pub_amd.resnet import resnet50  # noqa: E402


def _model(mode):
    sd = ro.init_resnet50_state_dict(seed=99, perturb_bn=True)
    m = resnet50(pretrained=False, compute_dtype=mode)
    full = m.state_dict()
    full.update(sd)
    m.load_state_dict(full)
    return m.to("cuda:0").eval(), sd


@pytest.mark.parametrize("mode,tol", [("fp32", 1e-4), ("bf16", 5e-2)])
def test_features_match_reference_golden(golden_dir, mode, tol):
    _lib.require_gpu()
    z = np.load(os.path.join(golden_dir, "resnet50.npz"))
    m, sd = _model(mode)
    f224 = m.extract_patches_u8(synth.patches_u8(0, n_patches=2, size=224)).cpu().numpy()
    f256 = m.extract_patches_u8(synth.patches_u8(1, n_patches=1, size=256)).cpu().numpy()
    e224, e256 = rel_err(f224, z["feat224"]), rel_err(f256, z["feat256"])
    print(f"resnet50 {mode}: rel err 224px {e224:.3e}  256px {e256:.3e}")
    assert e224 < tol and e256 < tol


def test_forward_extract_float_input_and_batch_consistency():
    """reference call form (normalised fp32 NCHW, compute_features_hdf5.py:119-122) == fused uint8 path;
    a batch of 5 == five single-patch calls (the reference loop is batch 1)."""
    _lib.require_gpu()
    m, sd = _model("fp32")
    p = synth.patches_u8(3, n_patches=5, size=224)
    fused = m.extract_patches_u8(p).cpu().numpy()
    x = ro.transform_patch_u8(p)
    direct = m.forward_extract(x).cpu().numpy()
    assert rel_err(direct, fused) < 1e-6
    singles = np.concatenate([m.extract_patches_u8(p[i:i + 1]).cpu().numpy() for i in range(5)])
    assert np.array_equal(singles, fused)
    with torch.no_grad():
        ref = ro.forward_extract(sd, x).numpy()
    assert rel_err(fused, ref) < 1e-4


def test_two_streams_in_flight_equal_one(monkeypatch):
    """extract_patches_u8 keeps two sub-batches in flight on two streams: same bits as the sequential run."""
    _lib.require_gpu()
    from sequoia_pub_amd import synth
    torch.manual_seed(3)
    rn = resnet50(pretrained=False, compute_dtype="bf16").to("cuda:0").eval()
    patches = torch.from_numpy(synth.patches_u8(5, 24, 224)).cuda()
    a = rn.extract_patches_u8(patches, sub_batch=7)              # 4 chunks, ragged last one, 2 streams
    monkeypatch.setenv("SQ_RESNET_STREAMS", "1")
    b = rn.extract_patches_u8(patches, sub_batch=7)
    c = rn.extract_patches_u8(patches, sub_batch=24)
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    assert torch.equal(a, c)                                      # sub-batching itself does not change a patch's features


def test_fused_bottleneck_tail_is_bit_identical(monkeypatch):
    """bf16, 56 x 56 stage: the fused launch (3x3 -> expand + identity -> next block's reduce, csrc/bottleneck.hip)
    against the same network run as separate GEMM-engine launches (SQ_RESNET_NO_FUSE=1): identical bits, for a patch
    count whose pixel total is not a multiple of the 128-pixel tile and for single patches (image borders inside a tile)."""
    _lib.require_gpu()
    torch.manual_seed(5)
    rn = resnet50(pretrained=False, compute_dtype="bf16").to("cuda:0").eval()
    for m in rn.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
    for npatch in (1, 3, 7):
        patches = torch.from_numpy(synth.patches_u8(11 + npatch, npatch, 224)).cuda()
        monkeypatch.delenv("SQ_RESNET_NO_FUSE", raising=False)
        fused = rn.extract_patches_u8(patches)
        monkeypatch.setenv("SQ_RESNET_NO_FUSE", "1")
        plain = rn.extract_patches_u8(patches)
        torch.cuda.synchronize()
        assert torch.isfinite(fused).all()
        assert torch.equal(fused, plain), (npatch, float((fused - plain).abs().max()))
    monkeypatch.delenv("SQ_RESNET_NO_FUSE", raising=False)


def test_halo_staged_3x3_matches_implicit_gemm(monkeypatch):
    """conv_halo.hip (input tile resident in LDS, channel-block-major K order) against the tap-major implicit-GEMM
    kernels on the same network: same features up to the bf16 re-rounding the different fp32 summation order causes,
    and both within the bf16 tolerance of the fp32 mode."""
    _lib.require_gpu()
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for halo in ("1", "0"):                       # the switch is read once per process
        env = dict(os.environ, SQ_CONV_HALO=halo, SQ_CONV_HALO_MIN_TILES="1")
        p = os.path.join("/tmp", f"sq_halo_{halo}_{os.getpid()}.pt")
        r = subprocess.run([sys.executable, os.path.join(root, "tests", "halo_check_worker.py"), p], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-800:]
        outs.append(torch.load(p))
        os.remove(p)
    a, b = outs
    assert torch.isfinite(a).all() and not torch.equal(a, b)            # a different kernel really ran
    assert float((a - b).abs().max() / b.abs().max()) < 1e-2
