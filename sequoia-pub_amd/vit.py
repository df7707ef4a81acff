"""ViT -- host-side mirror of /root/reference/src/vit.py:91-115 (softmax-attention baseline, ``--model_type vit``):
``ViT(*, num_outputs, dim, depth, heads, mlp_dim, dim_head=64, num_clusters=100, device='cuda')`` with the
reference's ``state_dict`` keys; arithmetic in ``sq_vit_forward`` / ``sq_vit_backward`` (csrc/vit.hip).
Shares the flat-parameter plumbing of :class:`sequoia_pub_amd.vis.ViS`."""
import ctypes
from collections import OrderedDict

import torch
import torch.nn as nn

from . import _lib
from ._lib import VitConfig, VitLayout
from .vis import ViS


def vit_layout(cfg):
    lay = VitLayout()
    _lib.check(_lib.lib().sq_vit_layout_init(ctypes.byref(cfg), ctypes.byref(lay)))
    return lay


def vit_tensor_map(cfg, lay):
    D, I, F, G, N = cfg.dim, cfg.heads * 64, cfg.mlp_dim, cfg.num_outputs, cfg.num_clusters
    m = OrderedDict()
    m["pos_emb1D"] = (lay.pos, (N, D))
    for l in range(cfg.depth):
        L = lay.layer[l]
        p = f"transformer.layers.{l}.0."
        m[p + "norm.weight"] = (L.ln1_g, (D,))
        m[p + "norm.bias"] = (L.ln1_b, (D,))
        m[p + "to_qkv.weight"] = (L.qkv_w, (3 * I, D))
        m[p + "to_out.weight"] = (L.out_w, (D, I))
        p = f"transformer.layers.{l}.1.net."
        m[p + "0.weight"] = (L.ln2_g, (D,))
        m[p + "0.bias"] = (L.ln2_b, (D,))
        m[p + "1.weight"] = (L.ff1_w, (F, D))
        m[p + "1.bias"] = (L.ff1_b, (F,))
        m[p + "3.weight"] = (L.ff2_w, (D, F))
        m[p + "3.bias"] = (L.ff2_b, (D,))
    m["linear_head.0.weight"] = (lay.head_ln_g, (D,))
    m["linear_head.0.bias"] = (lay.head_ln_b, (D,))
    m["linear_head.1.weight"] = (lay.head_w, (G, D))
    m["linear_head.1.bias"] = (lay.head_b, (G,))
    return m


class ViT(ViS):
    _C_WS, _C_FWD, _C_BWS, _C_BWD = "sq_vit_workspace_bytes", "sq_vit_forward", "sq_vit_backward_workspace_bytes", "sq_vit_backward"

    def _dim(self):
        return self.cfg.dim

    def __init__(self, *, num_outputs, dim, depth, heads, mlp_dim, dim_head=64, num_clusters=100, device='cuda',
                 compute_dtype='fp32'):
        nn.Module.__init__(self)
        if dim_head != 64:
            raise ValueError("the HIP attention kernels are specialised for dim_head = 64 (src/main.py:143,161-163)")
        self.cfg = VitConfig(int(dim), int(depth), int(heads), int(mlp_dim), int(num_outputs), int(num_clusters))
        self.layout = vit_layout(self.cfg)
        self._tmap = vit_tensor_map(self.cfg, self.layout)
        self.compute_dtype = _lib.DTYPES[compute_dtype]
        self.device = device
        flat = torch.zeros(self.layout.total, dtype=torch.float32)

        def put(key, t):
            off, _ = self._tmap[key]
            flat[off:off + t.numel()] = t.detach().reshape(-1)

        inner = heads * 64
        put("pos_emb1D", torch.randn(num_clusters, dim))              # same draw order as vit.py:96-104
        for l in range(depth):
            p = f"transformer.layers.{l}.0."
            put(p + "norm.weight", torch.ones(dim))
            put(p + "to_qkv.weight", nn.Linear(dim, inner * 3, bias=False).weight)
            put(p + "to_out.weight", nn.Linear(inner, dim, bias=False).weight)
            p = f"transformer.layers.{l}.1.net."
            put(p + "0.weight", torch.ones(dim))
            lin = nn.Linear(dim, mlp_dim)
            put(p + "1.weight", lin.weight); put(p + "1.bias", lin.bias)
            lin = nn.Linear(mlp_dim, dim)
            put(p + "3.weight", lin.weight); put(p + "3.bias", lin.bias)
        put("linear_head.0.weight", torch.ones(dim))
        lin = nn.Linear(dim, num_outputs)
        put("linear_head.1.weight", lin.weight); put("linear_head.1.bias", lin.bias)
        self.flat = nn.Parameter(flat)
        self._lp, self._lp_version, self._ws, self._ws_key = None, -1, None, None
        self._register_state_dict_hook(ViS._sd_hook)
        self._register_load_state_dict_pre_hook(self._load_hook)

    def replace_head(self, head):
        ln, lin = head[0], head[1]
        G = lin.out_features
        old_lay, c = self.layout, self.cfg
        cfg = VitConfig(c.dim, c.depth, c.heads, c.mlp_dim, int(G), c.num_clusters)
        lay = vit_layout(cfg)
        flat = torch.zeros(lay.total, dtype=torch.float32, device=self.flat.device)
        flat[:old_lay.head_ln_g] = self.flat.detach()[:old_lay.head_ln_g]
        D = cfg.dim
        flat[lay.head_ln_g:lay.head_ln_g + D] = ln.weight.detach().to(flat.device)
        flat[lay.head_ln_b:lay.head_ln_b + D] = ln.bias.detach().to(flat.device)
        flat[lay.head_w:lay.head_w + G * D] = lin.weight.detach().reshape(-1).to(flat.device)
        flat[lay.head_b:lay.head_b + G] = lin.bias.detach().to(flat.device)
        self.cfg, self.layout = cfg, lay
        self._tmap = vit_tensor_map(cfg, lay)
        self.flat = nn.Parameter(flat)
        self._lp, self._lp_version, self._ws, self._ws_key = None, -1, None, None
