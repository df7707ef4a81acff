"""UNI oracle (oracle/uni_oracle.py, a restatement of timm's vit_large_patch16_224) against an INDEPENDENT third-party
implementation of the same published architecture: HuggingFace transformers' ``ViTModel`` (patch-embedding conv, class
token + learned position embedding, pre-norm blocks with softmax attention and exact-GELU MLP, final LayerNorm,
class-token read-out).  transformers has no LayerScale; gamma is folded into ``attn.proj`` / ``mlp.fc2`` (exact up to
fp32 rounding), which is also what the HIP path's execution copy does (sequoia_pub_amd/uni.py:_exec_params).

This does NOT pin the oracle to the reference (timm is absent from the image, compute_features_hdf5.py:63-64): it
shows that two independent restatements of the published algorithm agree, on seeded weights, at a reduced size and at
the full ViT-L/16 size."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import uni_oracle  # noqa: E402

transformers = pytest.importorskip("transformers")


def _hf_model(sd, dim, depth, heads, mlp_dim, img):
    cfg = transformers.ViTConfig(hidden_size=dim, num_hidden_layers=depth, num_attention_heads=heads, intermediate_size=mlp_dim,
                                 hidden_act="gelu", hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, layer_norm_eps=1e-6,
                                 image_size=img, patch_size=16, num_channels=3, qkv_bias=True)
    m = transformers.ViTModel(cfg, add_pooling_layer=False).eval()
    hf = {"embeddings.cls_token": sd["cls_token"], "embeddings.position_embeddings": sd["pos_embed"],
          "embeddings.patch_embeddings.projection.weight": sd["patch_embed.proj.weight"],
          "embeddings.patch_embeddings.projection.bias": sd["patch_embed.proj.bias"],
          "layernorm.weight": sd["norm.weight"], "layernorm.bias": sd["norm.bias"]}
    v5 = any(k.startswith("layers.") for k in m.state_dict())          # transformers >= 5 renamed the encoder's tensors
    for i in range(depth):
        t = f"blocks.{i}."
        h = f"layers.{i}." if v5 else f"encoder.layer.{i}."
        q, k, v, o = ((h + "attention." + n for n in ("q_proj", "k_proj", "v_proj", "o_proj")) if v5 else
                      (h + "attention.attention.query", h + "attention.attention.key", h + "attention.attention.value", h + "attention.output.dense"))
        fc1, fc2 = (h + "mlp.fc1", h + "mlp.fc2") if v5 else (h + "intermediate.dense", h + "output.dense")
        qw, kw, vw = sd[t + "attn.qkv.weight"].chunk(3, 0)
        qb, kb, vb = sd[t + "attn.qkv.bias"].chunk(3, 0)
        g1, g2 = sd[t + "ls1.gamma"], sd[t + "ls2.gamma"]
        hf.update({h + "layernorm_before.weight": sd[t + "norm1.weight"], h + "layernorm_before.bias": sd[t + "norm1.bias"],
                   q + ".weight": qw, q + ".bias": qb, k + ".weight": kw, k + ".bias": kb, v + ".weight": vw, v + ".bias": vb,
                   o + ".weight": g1[:, None] * sd[t + "attn.proj.weight"], o + ".bias": g1 * sd[t + "attn.proj.bias"],
                   h + "layernorm_after.weight": sd[t + "norm2.weight"], h + "layernorm_after.bias": sd[t + "norm2.bias"],
                   fc1 + ".weight": sd[t + "mlp.fc1.weight"], fc1 + ".bias": sd[t + "mlp.fc1.bias"],
                   fc2 + ".weight": g2[:, None] * sd[t + "mlp.fc2.weight"], fc2 + ".bias": g2 * sd[t + "mlp.fc2.bias"]})
    missing, unexpected = m.load_state_dict({k: v.clone() for k, v in hf.items()}, strict=False)
    assert not unexpected and not [k for k in missing if "pooler" not in k], (missing, unexpected)
    return m


@pytest.mark.parametrize("dim,depth,heads,mlp_dim,batch", [(128, 3, 2, 512, 3), (1024, 24, 16, 4096, 1)])
def test_oracle_matches_transformers_vit(dim, depth, heads, mlp_dim, batch):
    torch.manual_seed(0)
    sd = uni_oracle.init_state_dict(dim=dim, depth=depth, heads=heads, mlp_dim=mlp_dim, seed=7, scale_ls=0.5)
    patches = torch.randint(0, 256, (batch, 224, 224, 3), dtype=torch.uint8)
    x = uni_oracle.transform_patch_u8(patches)
    with torch.no_grad():
        ours = uni_oracle.forward(sd, x, heads)
        theirs = _hf_model(sd, dim, depth, heads, mlp_dim, 224)(pixel_values=x).last_hidden_state[:, 0]
    err = float((ours - theirs).abs().max() / theirs.abs().max())
    assert ours.shape == (batch, dim) and err < 2e-5, err        # fp32 on both sides: rounding of the folded gains only
