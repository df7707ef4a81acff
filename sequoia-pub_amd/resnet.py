"""resnet50 -- host-side mirror of the interface /root/reference/pre_processing/compute_features_hdf5.py
uses: ``model = resnet50(pretrained=True).to(device); model.eval(); model.forward_extract(x)``
(src/resnet.py:370-379, :155-170), with torchvision's ``resnet50`` state_dict key names.

The module keeps the reference's parameters/buffers (so ``load_state_dict`` of
``resnet50-19c8e357.pth`` works unchanged); ``eval()``/first use folds BatchNorm into the
convolutions in fp64 and packs the weights into the layout of ``sq_resnet50_layout``; all
arithmetic then runs in ``sq_resnet50_extract``.  Training mode is not implemented (the reference
only ever runs this network in eval mode, compute_features_hdf5.py:60)."""
import ctypes
import math
from collections import OrderedDict

import os

import torch
import torch.nn as nn

from . import _lib

LAYERS = (3, 4, 6, 3)
PLANES = (64, 128, 256, 512)
BN_EPS = 1e-5


def conv_names():
    """State-dict prefixes (conv, bn) in the kernel's conv order."""
    names = [("conv1", "bn1")]
    for li, nblocks in enumerate(LAYERS, start=1):
        for b in range(nblocks):
            p = f"layer{li}.{b}"
            names += [(p + ".conv1", p + ".bn1"), (p + ".conv2", p + ".bn2"), (p + ".conv3", p + ".bn3")]
            if b == 0:
                names.append((p + ".downsample.0", p + ".downsample.1"))
    return names


def resnet50_layout():
    lay = _lib.ResNet50Layout()
    _lib.check(_lib.lib().sq_resnet50_layout_init(ctypes.byref(lay)))
    return lay


def pack_weights(sd, lay=None):
    """state_dict -> (weights fp32 flat [w_total], bias fp32 flat [b_total]) with eval-mode BN folded (fp64)."""
    lay = lay or resnet50_layout()
    w = torch.zeros(lay.w_total, dtype=torch.float32)
    b = torch.zeros(lay.b_total, dtype=torch.float32)
    for i, (cn, bn) in enumerate(conv_names()):
        d = lay.conv[i]
        cw = sd[cn + ".weight"].detach().double().cpu()                  # [cout, cin, k, k]
        assert tuple(cw.shape) == (d.cout, d.cin, d.k, d.k), (cn, tuple(cw.shape))
        g = sd[bn + ".weight"].detach().double().cpu()
        beta = sd[bn + ".bias"].detach().double().cpu()
        mean = sd[bn + ".running_mean"].detach().double().cpu()
        var = sd[bn + ".running_var"].detach().double().cpu()
        scale = g / torch.sqrt(var + BN_EPS)
        wf = (cw * scale.view(-1, 1, 1, 1)).permute(0, 2, 3, 1).reshape(d.cout, d.k * d.k * d.cin)   # [cout][kh][kw][cin]
        if d.k_padded > wf.shape[1]:
            wf = torch.cat([wf, torch.zeros(d.cout, d.k_padded - wf.shape[1], dtype=torch.float64)], 1)
        w[d.w_off:d.w_off + d.cout * d.k_padded] = wf.reshape(-1).float()
        b[d.b_off:d.b_off + d.cout] = (beta - mean * scale).float()
    return w, b


def split_planes(w, b, dtype, lay=None):
    """Packed fp32 weights / biases -> what sq_resnet50_extract takes in the split modes: 16-bit hi plane followed by the
    lo plane (hi = cvt(w'), lo = cvt(w' - hi)), and biases followed by the per-output-channel factors that undo w' = w * s.
    bf16 planes: s = 1.  fp16 planes: s = the power of two that lifts the row's max |w| into (128, 256], so the lo plane
    (<= 2^-12 of the value) stays in fp16's normal range for every weight that matters in its row."""
    lay = lay or resnet50_layout()
    scale = torch.ones_like(b)
    if dtype == _lib.SQ_F16X3:
        w = w.clone()
        for i in range(len(lay.conv)):
            d = lay.conv[i]
            rows = w[d.w_off:d.w_off + d.cout * d.k_padded].view(d.cout, d.k_padded)
            amax = rows.abs().amax(dim=1).double()
            s = torch.where(amax > 0, torch.exp2(torch.floor(torch.log2(256.0 / amax.clamp_min(1e-30)))), torch.ones_like(amax))
            s = s.clamp(2.0 ** -20, 2.0 ** 20).float()
            rows.mul_(s[:, None])                      # exact: powers of two
            scale[d.b_off:d.b_off + d.cout] = 1.0 / s
        hdt = torch.float16
    else:
        hdt = torch.bfloat16
    # every convolution behind the stem in K-tile-major order: element (n, k) at ((k // 32) * cout + n) * 32 + k % 32 of its
    # block, so that a 32-deep K-tile of consecutive output channels is one contiguous run (the kernels' weight requests
    # then cover whole cache lines; csrc/gemm.h b_tiled).  The stem's [64][152] rows are read straight into registers.
    if w is not None:
        w = w.clone()
        for i in range(1, len(lay.conv)):
            d = lay.conv[i]
            blk = w[d.w_off:d.w_off + d.cout * d.k_padded]
            blk.copy_(blk.view(d.cout, d.k_padded // 32, 32).permute(1, 0, 2).reshape(-1))
    hi = w.to(hdt)
    lo = (w - hi.float()).to(hdt)
    return torch.cat([hi, lo]).view(torch.int16), torch.cat([b, scale])


class ResNet50(nn.Module):
    """Reference-shaped container (same parameter / buffer names as src/resnet.py ResNet(Bottleneck,[3,4,6,3]))."""

    def __init__(self, num_classes=1000, compute_dtype="fp32"):
        super().__init__()
        self.compute_dtype = _lib.DTYPES[compute_dtype]

        def conv(cout, cin, k):
            c = nn.Conv2d(cin, cout, k, bias=False)
            n = k * k * cout
            c.weight.data.normal_(0, math.sqrt(2.0 / n))                  # resnet.py:113-116
            return c

        self.conv1 = conv(64, 3, 7)
        self.bn1 = nn.BatchNorm2d(64)
        inplanes = 64
        for li, (nblocks, planes) in enumerate(zip(LAYERS, PLANES), start=1):
            blocks = []
            for b in range(nblocks):
                blk = nn.Module()
                blk.conv1, blk.bn1 = conv(planes, inplanes, 1), nn.BatchNorm2d(planes)
                blk.conv2, blk.bn2 = conv(planes, planes, 3), nn.BatchNorm2d(planes)
                blk.conv3, blk.bn3 = conv(planes * 4, planes, 1), nn.BatchNorm2d(planes * 4)
                if b == 0:
                    blk.downsample = nn.Sequential(conv(planes * 4, inplanes, 1), nn.BatchNorm2d(planes * 4))
                blocks.append(blk)
                inplanes = planes * 4
            setattr(self, f"layer{li}", nn.Sequential(*blocks))
        self.fc = nn.Linear(2048, num_classes)                            # unused by forward_extract, kept for checkpoints
        self._packed = None
        self._ws = None

    # ---- packing -------------------------------------------------------------------------------
    def _pack(self):
        dev = self.conv1.weight.device
        # BatchNorm running statistics are folded into the packed weights too: buffer updates must repack
        key = (dev, self.compute_dtype, tuple(p._version for p in self.parameters()), tuple(b._version for b in self.buffers()))
        if self._packed is not None and self._packed[0] == key:
            return self._packed[1], self._packed[2]
        w, b = pack_weights(self.state_dict())
        if self.compute_dtype in (_lib.SQ_BF16X3, _lib.SQ_F16X3):
            w, b = split_planes(w, b, self.compute_dtype)
        w = w.to(dev)
        if self.compute_dtype == _lib.SQ_BF16:
            w = w.to(torch.bfloat16)
        self._packed = (key, w, b.to(dev))
        return self._packed[1], self._packed[2]

    def mark_dirty(self):
        """Drop the packed weights and the fp32 twin's sync mark.  The caches are keyed on the tensors' autograd version counters,
        which ``load_state_dict`` / in-place tensor ops bump but ``p.data.copy_()``, ``.data.normal_()`` and raw-pointer writes do
        NOT: call this after such an update (load_state_dict, .to() / .cuda() and friends call it themselves)."""
        self.__dict__["_packed"] = None
        self.__dict__["_twin_key"] = None

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self.mark_dirty()
        return out

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self.mark_dirty()
        return out

    def train(self, mode=True):
        if mode and getattr(self, "_built", False):
            raise NotImplementedError("sequoia-pub_amd ResNet50 runs in eval mode only (as the reference does)")
        return super().train(mode)

    # ---- the reduced range of the split-fp16 mode ----------------------------------------------------------------
    def exact_twin(self):
        """The same network in the exact fp32 mode (a copy of the parameters, refreshed only when they changed): what a
        launch group is re-run in when the split-fp16 mode overflowed (an activation >= 65504; sq_resnet50_extract_checked)."""
        key = (self.conv1.weight.device, tuple(p._version for p in self.parameters()), tuple(b._version for b in self.buffers()))
        tw = self.__dict__.get("_twin")
        if tw is None:
            tw = ResNet50(num_classes=self.fc.out_features, compute_dtype="fp32")
            tw._built = True
            self.__dict__["_twin"] = tw                  # not a registered submodule: state_dict() keeps the reference's keys
            self.__dict__["_twin_key"] = None
        if self.__dict__.get("_twin_key") != key:        # (load_state_dict bumps every version: synced once per change, so the twin's pack cache hits)
            tw.load_state_dict(self.state_dict(), strict=True)
            tw.to(self.conv1.weight.device).eval()
            self.__dict__["_twin_key"] = key
        return tw

    def new_flag(self):
        """A zeroed device word for sq_resnet50_extract_checked's non-finite flag."""
        return torch.zeros(1, dtype=torch.int32, device=self.conv1.weight.device)

    def _resolve_nonfinite(self, feats, flag, on_nonfinite, rerun):
        """Read the flag (host sync); overflowed -> raise, or re-run in fp32 (`rerun()` returns the exact features)."""
        if flag is None or int(flag.item()) == 0:
            return feats
        msg = ("ResNet50 in split-fp16 mode (f16x3): an activation left fp16's range (>= 65504) and the features are not finite"
               " -- unusual weights or BatchNorm statistics")
        if on_nonfinite == "raise":
            raise _lib.SequoiaHipError(msg + "; use compute_dtype='fp32' or 'bf16x3' for this checkpoint")
        import warnings
        warnings.warn(msg + "; this launch group is re-run in exact fp32", RuntimeWarning, stacklevel=3)
        self.last_nonfinite_reruns = getattr(self, "last_nonfinite_reruns", 0) + 1
        return rerun()

    def _run(self, patches_u8=None, x_f32=None, slot=0, flag=None):
        _lib.require_gpu()
        w, b = self._pack()
        if not w.is_cuda:
            raise _lib.SequoiaHipError("ResNet50 weights are on the CPU: call .to('cuda') first (no CPU fallback)")
        src = patches_u8 if patches_u8 is not None else x_f32
        n = src.shape[0]
        S = src.shape[1] if patches_u8 is not None else src.shape[2]
        feats = torch.empty(n, 2048, dtype=torch.float32, device=w.device)
        need = _lib.lib().sq_resnet50_workspace_bytes(self.compute_dtype, n, S)
        if need == 0:
            raise ValueError(f"unsupported patch size {S} (need a multiple of 32, >= 224)")
        if not isinstance(self._ws, dict):
            self._ws = {}
        ws = self._ws.get(slot)
        if ws is None or ws.numel() < need or ws.device != w.device:
            ws = self._ws[slot] = torch.empty(need, dtype=torch.uint8, device=w.device)
        with torch.cuda.device(w.device):
            _lib.check(_lib.lib().sq_resnet50_extract_checked(self.compute_dtype, _lib.ptr(w), _lib.ptr(b), _lib.ptr(patches_u8),
                                                              _lib.ptr(x_f32), n, S, _lib.ptr(feats), _lib.ptr(ws), ws.numel(),
                                                              _lib.ptr(flag) if flag is not None else None,
                                                              _lib.stream_ptr(w.device)))
        return feats

    @torch.no_grad()
    def forward_extract(self, x, on_nonfinite="rerun"):
        """src/resnet.py:155-170: x f32 [n, 3, H, W] (normalised) -> f32 [n, 2048]."""
        dev = self.conv1.weight.device
        x = x.to(dev, torch.float32).contiguous()
        flag = self.new_flag() if self.compute_dtype == _lib.SQ_F16X3 else None
        feats = self._run(x_f32=x, flag=flag)

        def rerun():
            tw = self.exact_twin()                      # built and synced once, not per chunk
            step = max(1, min(128, tw.max_sub_batch(x.shape[2])))
            return torch.cat([tw._run(x_f32=x[i:i + step]) for i in range(0, x.shape[0], step)])
        return self._resolve_nonfinite(feats, flag, on_nonfinite, rerun)

    def max_sub_batch(self, S):
        """Largest launch group the 2 GiB buffer-descriptor limit allows at patch size S (sq_resnet50_extract's check): the
        [n, S/2, S/2, 64] activation planes, and in fp32 mode the [n (S/2)^2, 152] im2col matrix of the stem."""
        es = 4 if self.compute_dtype == _lib.SQ_F32 else 2
        per = (S // 2) ** 2 * (152 if self.compute_dtype == _lib.SQ_F32 else 64) * es
        return max(1, ((1 << 31) - 1) // per)

    @torch.no_grad()
    def extract_patches_u8(self, patches, sub_batch=500, on_nonfinite="rerun", flag=None):
        """uint8 HWC patches [n, S, S, 3] -> f32 [n, 2048]; fuses compute_features_hdf5.py:119-120's transform.
        Patches go through in launch groups of <= sub_batch (clamped to what the 2 GiB buffer-descriptor limit allows for
        this mode and patch size); larger groups measured faster -- longer grids, fewer tails.
        Split-fp16 mode: a group whose features came out non-finite (an activation beyond fp16's range) is, per
        `on_nonfinite`, re-run in exact fp32 with a warning ("rerun", one host sync per call), reported by an exception
        ("raise"), or left to the caller ("defer": no sync; pass `flag`, a zeroed int32 device word from new_flag(), and
        check it later -- SlidePipeline does)."""
        dev = self.conv1.weight.device
        patches = torch.as_tensor(patches)
        if patches.shape[0] == 0:                        # a slide whose patch store is empty: no features, no launch
            return torch.empty(0, 2048, dtype=torch.float32, device=dev)
        sub_batch = max(1, min(int(sub_batch), self.max_sub_batch(patches.shape[1])))
        if self.compute_dtype == _lib.SQ_F16X3:
            if on_nonfinite != "defer":
                flag = self.new_flag()
                feats = self.extract_patches_u8(patches, sub_batch, "defer", flag)
                return self._resolve_nonfinite(feats, flag, on_nonfinite, lambda: self.exact_twin().extract_patches_u8(patches, 128))
        else:
            flag = None
        chunks = [patches[i:i + sub_batch] for i in range(0, patches.shape[0], sub_batch)]
        if len(chunks) == 1 or not patches.is_cuda:
            return torch.cat([self._run(patches_u8=c.to(dev).contiguous(), flag=flag) for c in chunks], 0)
        # Two sub-batches in flight on two streams (own workspaces): every convolution of the chain is one launch
        # that depends on the previous one, so a single chain leaves the chip idle in each launch's ramp-up, tail and
        # store-drain phase; a second, independent chain fills those.
        main = torch.cuda.current_stream(dev)
        self._pack()            # (re)pack on the main stream, ahead of `start`: the chains only wait on that event
        ns = max(1, int(os.environ.get("SQ_RESNET_STREAMS", "2")))
        if getattr(self, "_streams", None) is None or self._streams[0].device != dev or len(self._streams) != ns:
            self._streams = [torch.cuda.Stream(device=dev) for _ in range(ns)]
        start = torch.cuda.Event()
        start.record(main)
        outs = [None] * len(chunks)
        for i, c in enumerate(chunks):
            st = self._streams[i % ns]
            st.wait_event(start)
            with torch.cuda.stream(st):
                outs[i] = self._run(patches_u8=c.contiguous(), slot=i % ns, flag=flag)
        for st in self._streams:
            main.wait_stream(st)
        for o in outs:
            o.record_stream(main)
        return torch.cat(outs, 0)

    def forward(self, x):
        raise NotImplementedError("only forward_extract is on the SEQUOIA path (fc is never used by the reference scripts)")


def resnet50(pretrained=False, compute_dtype="fp32", **kwargs):
    """src/resnet.py:370-379.  ``pretrained=True`` needs the torchvision checkpoint on disk or network."""
    model = ResNet50(compute_dtype=compute_dtype, **kwargs)
    if pretrained:
        import torch.utils.model_zoo as model_zoo
        model.load_state_dict(model_zoo.load_url("https://download.pytorch.org/models/resnet50-19c8e357.pth"))
    model._built = True
    return model
