"""UNI patch embedder -- host-side mirror of what /root/reference/pre_processing/compute_features_hdf5.py:62-68 builds
with timm: ``feat_model = timm.create_model("vit_large_patch16_224", img_size=224, patch_size=16, init_values=1e-5,
num_classes=0, dynamic_img_size=True)``, ``load_state_dict(torch.load(.../pytorch_model.bin))``, ``.eval()``,
``.to(device)``, ``feat_model(image) -> [1, 1024]`` (:126-129; spatial_vis/visualize.py:220-232 likewise).

``UniViT`` keeps timm's state-dict keys (``cls_token``, ``pos_embed``, ``patch_embed.proj.*``,
``blocks.{i}.{norm1,attn.qkv,attn.proj,ls1.gamma,norm2,mlp.fc1,mlp.fc2,ls2.gamma}``, ``norm.*``) so the published
``pytorch_model.bin`` loads unchanged; the parameters live in one flat fp32 buffer (``sq_uni_layout``) and all
arithmetic runs in ``sq_uni_forward`` (csrc/uni.hip).  Forward only (the reference never trains the extractor).
timm is absent from the build image: parity is against oracle/uni_oracle.py, a restatement of timm's published
algorithm ("parity unpinned")."""
import ctypes
from collections import OrderedDict

import torch
import torch.nn as nn

from . import _lib

SQ_UNI_MAX_DEPTH = 32
_LAYER_FIELDS = ["ln1_g", "ln1_b", "qkv_w", "qkv_b", "proj_w", "proj_b", "ls1", "ln2_g", "ln2_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b", "ls2"]


class UniConfig(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("dim", "depth", "heads", "mlp_dim", "img_size")]


class UniLayerOffsets(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int64) for n in _LAYER_FIELDS]


class UniLayout(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int64) for n in ("patch_w", "patch_b", "cls", "pos", "norm_g", "norm_b", "total")] + \
               [("layer", UniLayerOffsets * SQ_UNI_MAX_DEPTH)]


vp, sz, i32 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
_lib.register_signature("sq_uni_layout_init", i32, [ctypes.POINTER(UniConfig), ctypes.POINTER(UniLayout)])
_lib.register_signature("sq_uni_workspace_bytes", sz, [ctypes.POINTER(UniConfig), i32, i32])
_lib.register_signature("sq_uni_forward", i32, [ctypes.POINTER(UniConfig), i32, vp, vp, vp, vp, vp, i32, vp, vp, sz, vp])


def tensor_map(cfg, lay):
    """timm state_dict key -> (offset, shape) in the flat buffer."""
    D, M, T = cfg.dim, cfg.mlp_dim, (cfg.img_size // 16) ** 2 + 1
    m = OrderedDict()
    m["cls_token"] = (lay.cls, (1, 1, D))
    m["pos_embed"] = (lay.pos, (1, T, D))
    m["patch_embed.proj.weight"] = (lay.patch_w, (D, 3, 16, 16))
    m["patch_embed.proj.bias"] = (lay.patch_b, (D,))
    for i in range(cfg.depth):
        L, p = lay.layer[i], f"blocks.{i}."
        m[p + "norm1.weight"] = (L.ln1_g, (D,)); m[p + "norm1.bias"] = (L.ln1_b, (D,))
        m[p + "attn.qkv.weight"] = (L.qkv_w, (3 * D, D)); m[p + "attn.qkv.bias"] = (L.qkv_b, (3 * D,))
        m[p + "attn.proj.weight"] = (L.proj_w, (D, D)); m[p + "attn.proj.bias"] = (L.proj_b, (D,))
        m[p + "ls1.gamma"] = (L.ls1, (D,))
        m[p + "norm2.weight"] = (L.ln2_g, (D,)); m[p + "norm2.bias"] = (L.ln2_b, (D,))
        m[p + "mlp.fc1.weight"] = (L.fc1_w, (M, D)); m[p + "mlp.fc1.bias"] = (L.fc1_b, (M,))
        m[p + "mlp.fc2.weight"] = (L.fc2_w, (D, M)); m[p + "mlp.fc2.bias"] = (L.fc2_b, (D,))
        m[p + "ls2.gamma"] = (L.ls2, (D,))
    m["norm.weight"] = (lay.norm_g, (D,)); m["norm.bias"] = (lay.norm_b, (D,))
    return m


def _numel(shape):
    n = 1
    for s in shape:
        n *= s
    return n


class UniViT(nn.Module):
    """timm VisionTransformer subset: ViT with class token, learned position embedding, LayerScale, token pooling,
    no classifier head.  ``forward(x f32 [B, 3, S, S]) -> f32 [B, dim]``."""

    def __init__(self, embed_dim=1024, depth=24, num_heads=16, mlp_ratio=4.0, img_size=224, patch_size=16, init_values=1e-5,
                 num_classes=0, compute_dtype="fp32", **_ignored):
        super().__init__()
        if patch_size != 16 or num_classes != 0 or embed_dim != num_heads * 64:
            raise ValueError("the HIP kernels cover patch_size=16, head dim 64, num_classes=0 (the reference's UNI configuration)")
        self.cfg = UniConfig(int(embed_dim), int(depth), int(num_heads), int(embed_dim * mlp_ratio), int(img_size))
        self.layout = UniLayout()
        _lib.check(_lib.lib().sq_uni_layout_init(ctypes.byref(self.cfg), ctypes.byref(self.layout)))
        self._tmap = tensor_map(self.cfg, self.layout)
        self.compute_dtype = _lib.DTYPES[compute_dtype]
        flat = torch.zeros(self.layout.total, dtype=torch.float32)
        # timm's init: trunc_normal(std .02) embeddings / Linear weights, zero biases, LayerNorm 1 / 0, LayerScale init_values
        g = torch.Generator().manual_seed(torch.initial_seed() % (2 ** 31))
        for k, (off, shape) in self._tmap.items():
            n = _numel(shape)
            if k.endswith("gamma"):
                flat[off:off + n] = init_values if init_values is not None else 1.0
            elif "norm" in k and k.endswith("weight"):
                flat[off:off + n] = 1.0
            elif k.endswith("bias"):
                pass
            else:
                flat[off:off + n] = torch.nn.init.trunc_normal_(torch.empty(n), std=0.02, generator=g)
        self.flat = nn.Parameter(flat, requires_grad=False)
        self._exec = None
        self._exec_key = None
        self._ws = {}
        self._register_state_dict_hook(UniViT._sd_hook)
        self._register_load_state_dict_pre_hook(self._load_hook)

    # ---- timm-keyed state_dict over the flat buffer -------------------------------------------------------------
    @staticmethod
    def _sd_hook(module, state_dict, prefix, local_metadata):
        flat = state_dict.pop(prefix + "flat")
        for k, (off, shape) in module._tmap.items():
            state_dict[prefix + k] = flat.detach()[off:off + _numel(shape)].reshape(shape).clone()
        return state_dict

    def _load_hook(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        if prefix + "flat" in state_dict or not any(prefix + k in state_dict for k in self._tmap):
            return
        flat = self.flat.detach().to("cpu", torch.float32).clone()
        for k, (off, shape) in self._tmap.items():
            full = prefix + k
            if full not in state_dict:
                if strict:
                    missing_keys.append(full)
                continue
            t = state_dict.pop(full).detach().to("cpu", torch.float32)
            if tuple(t.shape) != tuple(shape):
                error_msgs.append(f"size mismatch for {full}: {tuple(t.shape)} vs {tuple(shape)}")
                continue
            flat[off:off + t.numel()] = t.reshape(-1)
        state_dict[prefix + "flat"] = flat

    # ---- execution copies: LayerScale folded into attn.proj / mlp.fc2 -------------------------------------------------
    def _exec_params(self):
        dev = self.flat.device
        key = (dev, self.compute_dtype, self.flat._version)
        if self._exec_key == key:
            return self._exec
        w = self.flat.detach().clone()
        D, M = self.cfg.dim, self.cfg.mlp_dim
        for i in range(self.cfg.depth):
            L = self.layout.layer[i]
            for w_off, b_off, g_off, k in ((L.proj_w, L.proj_b, L.ls1, D), (L.fc2_w, L.fc2_b, L.ls2, M)):
                gam = w[g_off:g_off + D]
                w[w_off:w_off + D * k] = (w[w_off:w_off + D * k].view(D, k) * gam[:, None]).reshape(-1)
                w[b_off:b_off + D] = w[b_off:b_off + D] * gam
        bias_exec = w                                                   # fp32, folded biases (weights in it are unused)
        weights_exec = w.to(torch.bfloat16) if self.compute_dtype == _lib.SQ_BF16 else w
        self._exec, self._exec_key = (weights_exec, bias_exec), key
        return self._exec

    def _run(self, patches_u8=None, x_f32=None, slot=0):
        _lib.require_gpu()
        if not self.flat.is_cuda:
            raise _lib.SequoiaHipError("UniViT parameters are on the CPU: call .to('cuda') first (no CPU fallback)")
        dev = self.flat.device
        src = patches_u8 if patches_u8 is not None else x_f32
        n = src.shape[0]
        S = src.shape[1] if patches_u8 is not None else src.shape[2]
        if S != self.cfg.img_size:
            raise ValueError(f"patches are {S} x {S}, the model's position embedding is for {self.cfg.img_size} x {self.cfg.img_size}: "
                             "resize first (compute_features_hdf5.py:54 Resize(224))")
        wx, bx = self._exec_params()
        out = torch.empty(n, self.cfg.dim, dtype=torch.float32, device=dev)
        need = _lib.lib().sq_uni_workspace_bytes(ctypes.byref(self.cfg), self.compute_dtype, n)
        ws = self._ws.get(slot)
        if ws is None or ws.numel() < need or ws.device != dev:
            ws = self._ws[slot] = torch.empty(need, dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().sq_uni_forward(ctypes.byref(self.cfg), self.compute_dtype, _lib.ptr(self.flat), _lib.ptr(wx), _lib.ptr(bx),
                                                 _lib.ptr(patches_u8), _lib.ptr(x_f32), n, _lib.ptr(out), _lib.ptr(ws), ws.numel(),
                                                 _lib.stream_ptr(dev)))
        return out

    @torch.no_grad()
    def forward(self, x):
        """timm's ``model(image)`` with num_classes=0: f32 [B, 3, S, S] (normalised) -> f32 [B, dim]."""
        return self._run(x_f32=x.to(self.flat.device, torch.float32).contiguous())

    def max_sub_batch(self, S=None):
        """Largest launch group the 2 GiB buffer-descriptor limit allows (sq_uni_forward's check): the widest activation is
        [n * tokens, max(3 dim, mlp_dim)] in the compute dtype."""
        S = S or self.cfg.img_size
        tokens = (S // 16) ** 2 + 1
        es = 2 if self.compute_dtype == _lib.SQ_BF16 else 4
        return max(1, ((1 << 31) - 1) // (tokens * max(3 * self.cfg.dim, self.cfg.mlp_dim) * es))

    @torch.no_grad()
    def extract_patches_u8(self, patches, sub_batch=128):
        """uint8 HWC patches [n, S, S, 3] -> f32 [n, dim]; fuses the ToTensor + Normalize of compute_features_hdf5.py:53-56.
        Launch groups of <= sub_batch patches (clamped to the descriptor limit); larger groups are faster -- the 256 x 256 GEMM
        tiles then fill more rounds (bench: 6.09 / 6.38 / 6.58 slides/s at 256 / 500 / 1000)."""
        dev = self.flat.device
        patches = torch.as_tensor(patches)
        if patches.shape[0] == 0:
            return torch.empty(0, self.cfg.dim, dtype=torch.float32, device=dev)
        sub_batch = max(1, min(int(sub_batch), self.max_sub_batch(patches.shape[1])))
        return torch.cat([self._run(patches_u8=patches[i:i + sub_batch].to(dev).contiguous())
                          for i in range(0, patches.shape[0], sub_batch)], 0)


def resize_u8(patches_u8, size=224):
    """transforms.Resize(224) of compute_features_hdf5.py:54 for a batch of uint8 HWC patches on the device: bilinear with
    anti-aliasing (what PIL's BILINEAR resampler does when shrinking), rounded back to uint8.  PIL works in fixed point
    per image; this float version can differ by one grey level."""
    x = patches_u8.permute(0, 3, 1, 2).to(torch.float32)
    y = torch.nn.functional.interpolate(x, size=(size, size), mode="bilinear", antialias=True, align_corners=False)
    return y.round_().clamp_(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()


def create_model(name="vit_large_patch16_224", img_size=224, patch_size=16, init_values=1e-5, num_classes=0, dynamic_img_size=True,
                 compute_dtype="fp32", **kw):
    """The ``timm.create_model`` call of compute_features_hdf5.py:63-64 for the one architecture the reference uses."""
    if name != "vit_large_patch16_224":
        raise ValueError(f"only vit_large_patch16_224 (UNI) is provided, not {name!r}")
    return UniViT(embed_dim=1024, depth=24, num_heads=16, mlp_ratio=4.0, img_size=img_size, patch_size=patch_size,
                  init_values=init_values, num_classes=num_classes, compute_dtype=compute_dtype)
