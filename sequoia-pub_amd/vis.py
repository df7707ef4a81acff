"""ViS -- host-side mirror of /root/reference/src/tformer_lin.py:80-106.

Same constructor signature, ``forward([B,100,D]) -> [B,G]``, ``.device`` attribute,
``state_dict`` keys and HuggingFace ``PyTorchModelHubMixin`` behaviour as the reference
class, so ``load_state_dict`` / ``from_pretrained`` / ``torch.save(model.state_dict())``
round-trip with reference checkpoints.  All arithmetic runs in libsequoia_hip
(``sq_vis_forward`` / ``sq_vis_backward``); the parameters live in ONE flat fp32 buffer
(layout: ``sq_vis_layout_init``) in which every reference tensor is a contiguous slice.
"""
import ctypes
import math
from collections import OrderedDict

import torch
import torch.nn as nn

from . import _lib
from ._lib import HEAD_DIM, VisConfig, VisLayout

try:  # the reference class mixes this in (tformer_lin.py:4,80)
    from huggingface_hub import PyTorchModelHubMixin
except Exception:  # pragma: no cover - hub is optional
    class PyTorchModelHubMixin:  # type: ignore
        pass


def vis_layout(cfg: VisConfig) -> VisLayout:
    lay = VisLayout()
    _lib.check(_lib.lib().sq_vis_layout_init(ctypes.byref(cfg), ctypes.byref(lay)))
    return lay


def tensor_map(cfg: VisConfig, lay: VisLayout):
    """reference state_dict key -> (offset, shape) in the flat buffer, in the reference's key order."""
    D, H, G, N = cfg.input_dim, cfg.nheads, cfg.num_outputs, cfg.num_clusters
    hd = HEAD_DIM
    m = OrderedDict()
    m["pos_emb1D"] = (lay.pos, (N, D))
    for l in range(cfg.depth):
        L = lay.layer[l]
        p = f"transformer.layers.{l}.0."
        for h in range(H):
            q = p + f"mixers.{h}."
            m[q + "local_norm.weight"] = (L.lnf_g + h * hd, (hd,))
            m[q + "local_norm.bias"] = (L.lnf_b + h * hd, (hd,))
            m[q + "summary_norm.weight"] = (L.lns_g + h * hd, (hd,))
            m[q + "summary_norm.bias"] = (L.lns_b + h * hd, (hd,))
            m[q + "s.weight"] = (L.s_w + h * hd * D, (hd, D))
            m[q + "s.bias"] = (L.s_b + h * hd, (hd,))
            m[q + "f.weight"] = (L.f_w + h * hd * D, (hd, D))
            m[q + "f.bias"] = (L.f_b + h * hd, (hd,))
            m[q + "c.weight"] = (L.c_w + h * hd * 2 * hd, (hd, 2 * hd))
            m[q + "c.bias"] = (L.c_b + h * hd, (hd,))
        m[p + "projection.weight"] = (L.proj_w, (D, H * hd))
        m[p + "projection.bias"] = (L.proj_b, (D,))
        p = f"transformer.layers.{l}.1.net."
        m[p + "0.weight"] = (L.ffln_g, (D,))
        m[p + "0.bias"] = (L.ffln_b, (D,))
        m[p + "1.weight"] = (L.ff1_w, (D, D))
        m[p + "1.bias"] = (L.ff1_b, (D,))
        m[p + "3.weight"] = (L.ff2_w, (D, D))
        m[p + "3.bias"] = (L.ff2_b, (D,))
    m["linear_head.0.weight"] = (lay.head_ln_g, (D,))
    m["linear_head.0.bias"] = (lay.head_ln_b, (D,))
    m["linear_head.1.weight"] = (lay.head_w, (G, D))
    m["linear_head.1.bias"] = (lay.head_b, (G,))
    return m


def pack_state_dict(sd, cfg: VisConfig, lay: VisLayout = None, tmap=None):
    """reference-keyed state_dict -> flat fp32 CPU tensor."""
    lay = lay or vis_layout(cfg)
    tmap = tmap or tensor_map(cfg, lay)
    flat = torch.zeros(lay.total, dtype=torch.float32)
    for k, (off, shape) in tmap.items():
        t = torch.as_tensor(sd[k]).detach().to("cpu", torch.float32)
        if tuple(t.shape) != tuple(shape):
            raise ValueError(f"{k}: shape {tuple(t.shape)} != expected {tuple(shape)}")
        flat[off:off + t.numel()] = t.reshape(-1)
    return flat


class _VisFunction(torch.autograd.Function):
    """forward = sq_vis_forward; backward = sq_vis_backward (grads w.r.t. the flat buffer and x)."""

    @staticmethod
    def forward(ctx, flat, x, module, need_grad):
        out = module._run_forward(x, save=need_grad)
        ctx.saved_gen = module._saved_gen if need_grad else None     # which saving forward this node belongs to
        ctx.module = module
        ctx.need_x_grad = x.requires_grad
        ctx.batch = x.shape[0]
        ctx.x_shape = x.shape
        return out

    @staticmethod
    def backward(ctx, grad_out):
        module = ctx.module
        if ctx.saved_gen is None or ctx.saved_gen != module._saved_gen:
            # the activations live in ONE per-module workspace: a later forward (another micro-batch, an eval pass with
            # a different batch size, ...) has overwritten what this node saved -- refuse instead of returning garbage
            raise RuntimeError("ViS.backward: the saved activations of this forward were overwritten by a later forward "
                               "of the same module; call backward() before the next grad-enabled forward "
                               "(one forward in flight per module)")
        gflat, gx = module._run_backward(grad_out.contiguous(), ctx.batch, ctx.need_x_grad)
        if gx is not None:
            gx = gx.reshape(ctx.x_shape)
        return gflat, gx, None, None


class ViS(nn.Module, PyTorchModelHubMixin):
    # C entry points and config accessors (the ViT baseline subclasses this plumbing, vit.py)
    _C_WS, _C_FWD, _C_BWS, _C_BWD = "sq_vis_workspace_bytes", "sq_vis_forward", "sq_vis_backward_workspace_bytes", "sq_vis_backward"

    def _dim(self):
        return self.cfg.input_dim

    """Drop-in for the reference ``ViS`` (tformer_lin.py:80-106); see module docstring.

    Extra keyword ``compute_dtype``: ``"fp32"`` (exact-fp32 MFMA, parity mode, default) or
    ``"bf16"`` (bf16 MFMA with fp32 accumulation and fp32 residual stream, perf mode).
    """

    def __init__(self, num_outputs, input_dim, depth, nheads,
                 dimensions_f, dimensions_s, dimensions_c,
                 num_clusters=100, device='cuda:0', compute_dtype='fp32'):
        super().__init__()
        if not (dimensions_f == dimensions_s == dimensions_c == HEAD_DIM):
            raise ValueError("the HIP kernels are specialised for dimensions_f = dimensions_s = dimensions_c = 64 "
                             "(the value every reference call site uses: src/main.py:147,167,202)")
        self.cfg = VisConfig(int(input_dim), int(depth), int(nheads), int(num_outputs), int(num_clusters))
        self.layout = vis_layout(self.cfg)
        self._tmap = tensor_map(self.cfg, self.layout)
        self.compute_dtype = _lib.DTYPES[compute_dtype]
        self.device = device
        # Same RNG draw order as the reference constructor (tformer_lin.py:86-94: pos-emb randn, then
        # per layer per head s, f, c Linear, projection, FF Linears, head Linear) so that
        # torch.manual_seed(s); ViS(...) gives the reference's initial weights.
        flat = torch.zeros(self.layout.total, dtype=torch.float32)

        def put(key, t):
            off, shape = self._tmap[key]
            flat[off:off + t.numel()] = t.detach().reshape(-1)

        put("pos_emb1D", torch.randn(num_clusters, input_dim))
        hd = HEAD_DIM
        for l in range(depth):
            p = f"transformer.layers.{l}.0."
            for h in range(nheads):
                q = p + f"mixers.{h}."
                for name, (o, i) in (("s", (hd, input_dim)), ("f", (hd, input_dim)), ("c", (hd, 2 * hd))):
                    lin = nn.Linear(i, o)
                    put(q + name + ".weight", lin.weight)
                    put(q + name + ".bias", lin.bias)
                put(q + "local_norm.weight", torch.ones(hd))
                put(q + "summary_norm.weight", torch.ones(hd))
            lin = nn.Linear(nheads * hd, input_dim)
            put(p + "projection.weight", lin.weight)
            put(p + "projection.bias", lin.bias)
            p = f"transformer.layers.{l}.1.net."
            put(p + "0.weight", torch.ones(input_dim))
            for idx in ("1", "3"):
                lin = nn.Linear(input_dim, input_dim)
                put(p + idx + ".weight", lin.weight)
                put(p + idx + ".bias", lin.bias)
        put("linear_head.0.weight", torch.ones(input_dim))
        lin = nn.Linear(input_dim, num_outputs)
        put("linear_head.1.weight", lin.weight)
        put("linear_head.1.bias", lin.bias)
        self.flat = nn.Parameter(flat)
        self._lp = None            # bf16 shadow of `flat`
        self._lp_version = -1
        self._ws = None            # workspace (uint8) and what it was sized for
        self._ws_key = None
        self._register_state_dict_hook(ViS._sd_hook)
        self._register_load_state_dict_pre_hook(self._load_hook)

    # ---- reference-compatible state_dict -------------------------------------------------
    @staticmethod
    def _sd_hook(module, state_dict, prefix, local_metadata):
        flat = state_dict.pop(prefix + "flat")
        for k, (off, shape) in module._tmap.items():
            n = int(torch.tensor(shape).prod())
            state_dict[prefix + k] = flat.detach()[off:off + n].reshape(shape).clone()
        return state_dict

    def _load_hook(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        if prefix + "flat" in state_dict:
            return
        keys = [prefix + k for k in self._tmap]
        if not any(k in state_dict for k in keys):
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

    def named_reference_tensors(self):
        """Views (not copies) of the flat buffer under the reference's parameter names."""
        for k, (off, shape) in self._tmap.items():
            n = 1
            for s in shape:
                n *= s
            yield k, self.flat.detach()[off:off + n].view(shape)

    def grad_views(self, gflat):
        out = OrderedDict()
        for k, (off, shape) in self._tmap.items():
            n = 1
            for s in shape:
                n *= s
            out[k] = gflat[off:off + n].view(shape)
        return out

    # fine-tuning replaces the head (src/main.py:155-157: model.linear_head = nn.Sequential(LN, Linear))
    def __setattr__(self, name, value):
        if name == "linear_head" and isinstance(value, nn.Module):
            self.replace_head(value)
            return
        super().__setattr__(name, value)

    def replace_head(self, head):
        ln, lin = head[0], head[1]
        G = lin.out_features
        old_cfg, old_lay = self.cfg, self.layout
        cfg = VisConfig(old_cfg.input_dim, old_cfg.depth, old_cfg.nheads, int(G), old_cfg.num_clusters)
        lay = vis_layout(cfg)
        flat = torch.zeros(lay.total, dtype=torch.float32, device=self.flat.device)
        flat[:old_lay.head_ln_g] = self.flat.detach()[:old_lay.head_ln_g]
        D = cfg.input_dim
        flat[lay.head_ln_g:lay.head_ln_g + D] = ln.weight.detach().to(flat.device)
        flat[lay.head_ln_b:lay.head_ln_b + D] = ln.bias.detach().to(flat.device)
        flat[lay.head_w:lay.head_w + G * D] = lin.weight.detach().reshape(-1).to(flat.device)
        flat[lay.head_b:lay.head_b + G] = lin.bias.detach().to(flat.device)
        self.cfg, self.layout = cfg, lay
        self._tmap = tensor_map(cfg, lay)
        self.flat = nn.Parameter(flat)
        self._lp, self._lp_version, self._ws, self._ws_key = None, -1, None, None

    # ---- device plumbing --------------------------------------------------------------------
    def _params_lp(self):
        if self.compute_dtype != _lib.SQ_BF16:
            return None
        if self._lp is None or self._lp.device != self.flat.device or self._lp.numel() != self.flat.numel():
            self._lp = torch.empty(self.flat.numel(), dtype=torch.bfloat16, device=self.flat.device)
            self._lp_version = -1
        if self._lp_version != self.flat._version:
            _lib.check(_lib.lib().sq_cast_f32_to_bf16(_lib.ptr(self.flat), _lib.ptr(self._lp), self.flat.numel(),
                                                      _lib.stream_ptr(self.flat.device)))
            self._lp_version = self.flat._version
        return self._lp

    def _workspace(self, batch, save, slot=0):
        key = (batch, bool(save), self.compute_dtype, self.flat.device)
        need = getattr(_lib.lib(), self._C_WS)(ctypes.byref(self.cfg), self.compute_dtype, batch, int(save))
        if need == 0:
            _lib.check(-1)
        if slot:                     # extra inference workspaces: forwards in flight on several streams
            extra = self.__dict__.setdefault("_ws_extra", {})
            ws, k = extra.get(slot, (None, None))
            if ws is None or k != key or ws.numel() < need:
                ws = torch.empty(need, dtype=torch.uint8, device=self.flat.device)
                extra[slot] = (ws, key)
            return ws
        if self._ws is None or self._ws_key != key or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.flat.device)
            self._ws_key = key
        return self._ws

    def _run_forward(self, x, save, slot=0):
        _lib.require_gpu()
        if not self.flat.is_cuda:
            raise _lib.SequoiaHipError("ViS parameters are on the CPU: call .to('cuda') first (no CPU fallback)")
        x = x.detach().to(self.flat.device, torch.float32)
        x = x.reshape(x.shape[0], -1, x.shape[-1]).contiguous()      # rearrange 'b ... d -> b (...) d'
        B, N, D = x.shape
        if N != self.cfg.num_clusters or D != self._dim():
            raise ValueError(f"expected [B, {self.cfg.num_clusters}, {self._dim()}] tokens, got {tuple(x.shape)}")
        out = torch.empty(B, self.cfg.num_outputs, dtype=torch.float32, device=x.device)
        ws = self._workspace(B, save, slot)
        lp = self._params_lp()
        with torch.cuda.device(x.device):
            _lib.check(getattr(_lib.lib(), self._C_FWD)(ctypes.byref(self.cfg), self.compute_dtype, _lib.ptr(self.flat),
                                                         _lib.ptr(lp), _lib.ptr(x), _lib.ptr(out), B, int(save),
                                                         _lib.ptr(ws), ws.numel(), _lib.stream_ptr(x.device)))
        self._saved_x = x if save else None
        if slot == 0:           # slot-0 workspace rewritten (or re-allocated): earlier saved activations are gone
            self._saved_gen = self.__dict__.get("_saved_gen", 0) + 1
        return out

    def tile_projections(self, cache):
        """Layer 0's local projection f (tformer_lin.py:20, all heads) per TILE of a feature cache f32 [rows, D]: returns
        (f_tile f32 [rows, heads*64] = cache . Wf^T, f_pos f32 [num_clusters, heads*64] = pos_emb1D . Wf^T + bf).  f is linear in
        tile feature + position, so a window token's f(x) is f_tile[tile] + f_pos[slot] -- what sq_vis_forward_tiles gathers
        instead of running the projection over every window token (bf16 mode; the exact fp32 mode keeps the per-token product)."""
        _lib.require_gpu()
        if self._C_FWD != "sq_vis_forward" or self.compute_dtype != _lib.SQ_BF16:
            raise NotImplementedError("tile projections are the bf16 ViS sliding-window path's")
        lay, dev = self.layout, cache.device
        D, HD, N = self._dim(), self.cfg.nheads * 64, self.cfg.num_clusters
        lp = self._params_lp()
        L0 = lay.layer[0]
        w_ptr = ctypes.c_void_p(lp.data_ptr() + 2 * L0.f_w)
        a = cache.to(torch.bfloat16).contiguous()
        pos = self.flat.detach()[lay.pos:lay.pos + N * D].view(N, D).to(torch.bfloat16).contiguous()
        f_tile = torch.empty(a.shape[0], HD, dtype=torch.float32, device=dev)
        f_pos = torch.empty(N, HD, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            for r0 in range(0, a.shape[0], 32768):         # operand extents stay below the 2 GiB buffer-descriptor limit
                r1 = min(a.shape[0], r0 + 32768)
                _lib.check(_lib.lib().sq_linear(self.compute_dtype, _lib.ptr(a[r0:r1]), D, w_ptr, D, None, None, 0, 0, 0,
                                                _lib.ptr(f_tile[r0:r1]), _lib.SQ_F32, HD, r1 - r0, HD, D, None, 0, _lib.stream_ptr(dev)))
            b_ptr = ctypes.c_void_p(self.flat.data_ptr() + 4 * L0.f_b)
            _lib.check(_lib.lib().sq_linear(self.compute_dtype, _lib.ptr(pos), D, w_ptr, D, b_ptr, None, 0, 0, 0,
                                            _lib.ptr(f_pos), _lib.SQ_F32, HD, N, HD, D, None, 0, _lib.stream_ptr(dev)))
        return f_tile, f_pos

    def _run_head_inputs(self, cache, members, slot=0, tile_proj=None):
        """Sliding-window form (sq_vis_forward_ex): cache f32 [n_rows, D] on the device, members int32 [B, 100] rows of
        the cache per window (-1 = zero padding).  Returns the linear head's input LayerNorm(mean_tokens X) f32 [B, D]
        -- the window batch [B, 100, D] is gathered inside the first kernel and the head is left to the caller.
        ``tile_proj`` = tile_projections(cache): the first layer's local projection is gathered per tile (sq_vis_forward_tiles)."""
        _lib.require_gpu()
        if self._C_FWD != "sq_vis_forward":
            raise NotImplementedError("head inputs are implemented for ViS (the linear-attention aggregator)")
        B, N = members.shape
        if N != self.cfg.num_clusters or cache.shape[1] != self._dim():
            raise ValueError(f"expected members [B, {self.cfg.num_clusters}] and a [rows, {self._dim()}] cache")
        out = torch.empty(B, self._dim(), dtype=torch.float32, device=cache.device)
        ws = self._workspace(B, False, slot)
        lp = self._params_lp()
        with torch.cuda.device(cache.device):
            if tile_proj is not None:
                f_tile, f_pos = tile_proj
                if f_tile.shape != (cache.shape[0], self.cfg.nheads * 64) or f_pos.shape != (N, self.cfg.nheads * 64):
                    raise ValueError("tile_proj does not belong to this cache / model")
                _lib.check(_lib.lib().sq_vis_forward_tiles(ctypes.byref(self.cfg), self.compute_dtype, _lib.ptr(self.flat), _lib.ptr(lp),
                                                           _lib.ptr(cache), _lib.ptr(members), cache.shape[0], _lib.ptr(f_tile), _lib.ptr(f_pos),
                                                           _lib.ptr(out), B, _lib.ptr(ws), ws.numel(), _lib.stream_ptr(cache.device)))
            else:
                _lib.check(_lib.lib().sq_vis_forward_ex(ctypes.byref(self.cfg), self.compute_dtype, _lib.ptr(self.flat), _lib.ptr(lp), None,
                                                        _lib.ptr(cache), _lib.ptr(members), cache.shape[0], None, _lib.ptr(out), B, 0,
                                                        _lib.ptr(ws), ws.numel(), _lib.stream_ptr(cache.device)))
        return out

    def apply_head(self, head_in, chunk=32768):
        """linear_head[1] of the reference (tformer_lin.py:91-94,106) on already normalised inputs f32 [R, D] -> f32 [R, G],
        `chunk` rows per product (callers that shard rows over ranks pass the grid they share with the one-rank run)."""
        if not 0 < chunk <= 32768:
            raise ValueError("apply_head: chunk must be in 1..32768 (operand extents stay below the 2 GiB buffer-descriptor limit)")
        R, D = head_in.shape
        G = self.cfg.num_outputs
        lay, dev = self.layout, head_in.device
        lp = self._params_lp()
        out = torch.empty(R, G, dtype=torch.float32, device=dev)
        if lp is not None:
            a = head_in.to(torch.bfloat16).contiguous()
            w_ptr = ctypes.c_void_p(lp.data_ptr() + 2 * lay.head_w)
        else:
            a = head_in.contiguous()
            w_ptr = ctypes.c_void_p(self.flat.data_ptr() + 4 * lay.head_w)
        b_ptr = ctypes.c_void_p(self.flat.data_ptr() + 4 * lay.head_b)
        with torch.cuda.device(dev):
            for r0 in range(0, R, chunk):
                r1 = min(R, r0 + chunk)
                _lib.check(_lib.lib().sq_linear(self.compute_dtype, _lib.ptr(a[r0:r1]), D, w_ptr, D, b_ptr, None, 0, 0, 0,
                                                _lib.ptr(out[r0:r1]), _lib.SQ_F32, G, r1 - r0, G, D, None, 0, _lib.stream_ptr(dev)))
        return out

    def _run_backward(self, grad_out, batch, need_x_grad):
        from . import train as _train           # sq_vis_backward binding lives with the training step
        return _train.vis_backward(self, grad_out, batch, need_x_grad)

    def forward(self, x):
        if x.dim() == 2:
            # spatial_vis/visualize.py:82 feeds [100, D]; the reference's rearrange turns that into
            # [100, 1, D] (+ pos-emb broadcast).  We implement the intended one-window = one sample.
            x = x.unsqueeze(0)
        need_grad = torch.is_grad_enabled() and (self.flat.requires_grad or x.requires_grad)
        return _VisFunction.apply(self.flat, x, self, need_grad)

    def forward_literal_2d(self, x2d):
        """Bit-for-bit behaviour of the reference on a 2-D [100, D] input:
        output row i depends only on tile i, replicated over the 100 positions."""
        n = x2d.shape[0]
        return self.forward(x2d[:, None, :].expand(n, self.cfg.num_clusters, x2d.shape[-1]))
