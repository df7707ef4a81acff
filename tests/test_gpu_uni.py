"""UNI patch embedder (timm vit_large_patch16_224) on the HIP path vs oracle/uni_oracle.py -- a
restatement of timm's published algorithm ("parity unpinned": timm and the gated weights are absent).
Tolerances: fp32 (exact-fp32 MFMA) 1e-4 relative; bf16 5e-2."""
import numpy as np
import pytest
import torch

from gpu_util import rel_err

pytestmark = pytest.mark.gpu

from oracle import uni_oracle  # noqa: E402  (checker only)
from sequoia_pub_amd import _lib  # noqa: E402
from sequoia_pub_amd.uni import UniViT, create_model  # noqa: E402


def _pair(mode, **cfg):
    sd = uni_oracle.init_state_dict(dim=cfg["embed_dim"], depth=cfg["depth"], heads=cfg["num_heads"],
                                    mlp_dim=int(cfg["embed_dim"] * cfg.get("mlp_ratio", 4.0)), img_size=cfg["img_size"], seed=3,
                                    scale_ls=cfg.pop("scale_ls", 0.5))
    m = UniViT(compute_dtype=mode, **cfg)
    m.load_state_dict(sd)
    return m.to("cuda:0").eval(), sd


@pytest.mark.parametrize("mode,tol", [("fp32", 1e-4), ("bf16", 5e-2)])
def test_tiny_config_matches_oracle(mode, tol):
    """Small instance of the same architecture (2 heads of 64, 3 blocks, 5 x 5 + 1 tokens), LayerScale gains of O(1)."""
    _lib.require_gpu()
    m, sd = _pair(mode, embed_dim=128, depth=3, num_heads=2, mlp_ratio=4.0, img_size=80)
    rs = np.random.RandomState(0)
    patches = rs.randint(0, 256, (5, 80, 80, 3), dtype=np.uint8)
    got = m.extract_patches_u8(patches).cpu().numpy()
    x = uni_oracle.transform_patch_u8(patches)
    with torch.no_grad():
        ref = uni_oracle.forward(sd, x, heads=2).numpy()
    assert got.shape == (5, 128)
    e = rel_err(got, ref)
    print(f"UNI tiny {mode}: rel err {e:.3e}")
    assert e < tol
    direct = m(x).cpu().numpy()                                   # the reference's call form: normalised NCHW tensor
    assert rel_err(direct, got) < (1e-6 if mode == "fp32" else 2e-2)
    single = m.extract_patches_u8(patches[2:3]).cpu().numpy()     # batch 1 (the reference loop) == row of the batch
    assert np.array_equal(single[0], got[2])


@pytest.mark.parametrize("mode,tol", [("fp32", 2e-4), ("bf16", 6e-2)])
def test_vit_large_patch16_224_matches_oracle(mode, tol):
    """The real architecture (1024 wide, 24 blocks, 16 heads, 197 tokens) on two 224 x 224 patches, UNI's LayerScale
    init (1e-5) replaced by O(0.3) gains so that all 24 blocks contribute."""
    _lib.require_gpu()
    m, sd = _pair(mode, embed_dim=1024, depth=24, num_heads=16, mlp_ratio=4.0, img_size=224, scale_ls=0.3)
    rs = np.random.RandomState(1)
    patches = rs.randint(0, 256, (2, 224, 224, 3), dtype=np.uint8)
    got = m.extract_patches_u8(patches).cpu().numpy()
    torch.set_num_threads(min(32, torch.get_num_threads() or 8))
    with torch.no_grad():
        ref = uni_oracle.forward(sd, uni_oracle.transform_patch_u8(patches), heads=16).numpy()
    e = rel_err(got, ref)
    print(f"UNI ViT-L/16 {mode}: rel err {e:.3e}")
    assert got.shape == (2, 1024) and np.isfinite(got).all() and e < tol


def test_create_model_signature_state_dict_keys_and_size_check():
    _lib.require_gpu()
    m = create_model("vit_large_patch16_224", img_size=224, patch_size=16, init_values=1e-5, num_classes=0, dynamic_img_size=True)
    keys = list(m.state_dict().keys())
    assert keys[:4] == ["cls_token", "pos_embed", "patch_embed.proj.weight", "patch_embed.proj.bias"]
    assert "blocks.23.ls2.gamma" in keys and "norm.bias" in keys and len(keys) == 4 + 24 * 14 + 2
    assert sum(v.numel() for v in m.state_dict().values()) == 303_350_784          # ViT-L/16 + LayerScale, no head
    assert float(m.state_dict()["blocks.0.ls1.gamma"][0]) == pytest.approx(1e-5)
    m.to("cuda:0")
    with pytest.raises(ValueError):
        m.extract_patches_u8(np.zeros((1, 256, 256, 3), dtype=np.uint8))           # needs Resize(224) first
