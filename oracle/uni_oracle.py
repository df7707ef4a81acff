"""Oracle (test infrastructure): UNI patch embedder = timm ``vit_large_patch16_224`` forward, CPU fp32.

PARITY UNPINNED.  The reference builds the model with a third-party library that is neither vendored nor installed
here -- ``timm.create_model("vit_large_patch16_224", img_size=224, patch_size=16, init_values=1e-5, num_classes=0,
dynamic_img_size=True)`` (/root/reference/pre_processing/compute_features_hdf5.py:63-64, requirements.txt pins no
timm version; UNI's model card asks for timm >= 0.9.8) -- and loads gated weights (``pytorch_model.bin``, :65-66).  No
golden vector can be produced in the build image, so this file restates the PUBLISHED timm algorithm
(timm/models/vision_transformer.py, 0.9.x): ``PatchEmbed`` (Conv2d(3, D, 16, stride 16), flatten to [B, N, D]),
``_pos_embed`` (class token concatenated in front, learned position embedding added to all 1 + N tokens),
``Block`` (pre-norm; ``Attention``: fused qkv Linear with bias, heads of 64, softmax(q k^T * 64^-0.5) v, proj Linear;
``LayerScale`` gamma per channel on both branches; ``Mlp``: fc1, exact GELU, fc2), final ``LayerNorm`` (eps 1e-6
everywhere), ``forward_head`` with global_pool='token' and num_classes=0 = the normalised class token.
Independent cross-check (not a pin): tests/test_uni_oracle_vs_hf.py runs this restatement against HuggingFace
transformers' ViTModel -- a third-party implementation of the same published architecture -- on seeded weights at a
reduced size and at the full ViT-L/16 size (LayerScale folded into proj / fc2): agreement < 2e-5 (fp32).
The call site it stands behind: ``features = feat_model(image)`` with ``image`` = Resize(224) + ToTensor +
Normalize(ImageNet) of an RGB patch (compute_features_hdf5.py:53-56,126-129).  State-dict keys are timm's."""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)
LN_EPS = 1e-6


def transform_patch_u8(img_u8_hwc):
    """compute_features_hdf5.py:53-56 for patches that already are 224 x 224 (Resize is then the identity):
    ToTensor (HWC uint8 -> CHW float / 255) and Normalize."""
    x = torch.as_tensor(img_u8_hwc).movedim(-1, -3).to(torch.float32) / 255.0
    mean = torch.tensor(IMAGENET_MEAN, dtype=torch.float32).view(3, 1, 1)
    std = torch.tensor(IMAGENET_STD, dtype=torch.float32).view(3, 1, 1)
    return (x - mean) / std


def init_state_dict(dim=1024, depth=24, heads=16, mlp_dim=4096, img_size=224, seed=0, init_values=1e-5, scale_ls=None):
    """Seeded synthetic weights with timm's tensor names and shapes (trunc-normal-like std 0.02 weights, zero-ish
    biases made non-trivial so every term is exercised).  scale_ls: LayerScale value (default init_values; tests use
    O(1) gains so the branches matter numerically)."""
    g = torch.Generator().manual_seed(seed)
    ls = init_values if scale_ls is None else scale_ls
    n_tok = (img_size // 16) ** 2 + 1
    sd = OrderedDict()

    def rn(*shape, std=0.02):
        return torch.randn(*shape, generator=g) * std

    sd["cls_token"] = rn(1, 1, dim)
    sd["pos_embed"] = rn(1, n_tok, dim)
    sd["patch_embed.proj.weight"] = rn(dim, 3, 16, 16, std=0.05)
    sd["patch_embed.proj.bias"] = rn(dim, std=0.05)
    for i in range(depth):
        p = f"blocks.{i}."
        for nrm in ("norm1", "norm2"):
            sd[p + nrm + ".weight"] = 1.0 + rn(dim, std=0.1)
            sd[p + nrm + ".bias"] = rn(dim, std=0.1)
        sd[p + "attn.qkv.weight"] = rn(3 * dim, dim, std=1.0 / math.sqrt(dim))
        sd[p + "attn.qkv.bias"] = rn(3 * dim, std=0.05)
        sd[p + "attn.proj.weight"] = rn(dim, dim, std=1.0 / math.sqrt(dim))
        sd[p + "attn.proj.bias"] = rn(dim, std=0.05)
        sd[p + "ls1.gamma"] = ls * (1.0 + rn(dim, std=0.2))
        sd[p + "mlp.fc1.weight"] = rn(mlp_dim, dim, std=1.0 / math.sqrt(dim))
        sd[p + "mlp.fc1.bias"] = rn(mlp_dim, std=0.05)
        sd[p + "mlp.fc2.weight"] = rn(dim, mlp_dim, std=1.0 / math.sqrt(mlp_dim))
        sd[p + "mlp.fc2.bias"] = rn(dim, std=0.05)
        sd[p + "ls2.gamma"] = ls * (1.0 + rn(dim, std=0.2))
    sd["norm.weight"] = 1.0 + rn(dim, std=0.1)
    sd["norm.bias"] = rn(dim, std=0.1)
    return sd


def forward(sd, x, heads):
    """x: f32 [B, 3, S, S] normalised -> f32 [B, D] (the normalised class token)."""
    B = x.shape[0]
    D = sd["cls_token"].shape[-1]
    t = F.conv2d(x, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=16)       # [B, D, G, G]
    t = t.flatten(2).transpose(1, 2)                                                             # [B, N, D]
    t = torch.cat([sd["cls_token"].expand(B, -1, -1), t], dim=1) + sd["pos_embed"]
    depth = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("blocks."))
    hd = D // heads
    for i in range(depth):
        p = f"blocks.{i}."
        y = F.layer_norm(t, (D,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], LN_EPS)
        qkv = F.linear(y, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"])
        N = qkv.shape[1]
        q, k, v = qkv.reshape(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
        attn = ((q * hd ** -0.5) @ k.transpose(-2, -1)).softmax(dim=-1)
        o = (attn @ v).transpose(1, 2).reshape(B, N, D)
        o = F.linear(o, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
        t = t + sd[p + "ls1.gamma"] * o
        y = F.layer_norm(t, (D,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], LN_EPS)
        h = F.gelu(F.linear(y, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]))
        h = F.linear(h, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
        t = t + sd[p + "ls2.gamma"] * h
    t = F.layer_norm(t, (D,), sd["norm.weight"], sd["norm.bias"], LN_EPS)
    return t[:, 0]


def embed_patches(sd, patches_u8, heads, batch=1):
    """The uni branch of compute_features_hdf5.py:116-129: per-patch (batch=1, literal) or batched forward."""
    outs = []
    with torch.no_grad():
        for i in range(0, len(patches_u8), batch):
            outs.append(forward(sd, transform_patch_u8(patches_u8[i:i + batch]), heads))
    return torch.cat(outs, 0)
