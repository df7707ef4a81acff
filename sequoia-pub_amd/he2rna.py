"""HE2RNA -- the benchmark comparator of the reference (``/root/reference/src/he2rna.py:42-106``, built by
``src/pretrain_gtex.py:102-105`` as ``HE2RNA(input_dim, layers=[256, 256], ks=[1, 2, 5, 10, 20, 50, 100], output_dim)``)
on the HIP engine: same constructor, ``state_dict`` keys (``conv{i}.weight [out, in, 1]``, ``conv{i}.bias``),
``forward`` / ``forward_fixed_k`` / ``conv`` semantics and HuggingFace mixin, plus the loops ``training_epoch``,
``evaluate``, ``he2rna_predict`` and ``fit`` (he2rna.py:108-318).

Device path (fp32): the 1x1 Conv1d layers run as ``sq_linear`` on the token-major tensor [B * N, C] (the layout the
cluster features have BEFORE the reference's ``rearrange('b c f -> b f c')``), bias + ReLU fused; tile mask, masking,
top-k over the tiles and the position-weighted mean of he2rna.py:93-99 -- for all ``ks`` of the eval-mode mean in ONE
launch -- are ``sq_he2rna_tile_mask`` / ``sq_he2rna_topk_mean``; the backward pass is ``sq_he2rna_topk_mean_bwd``,
``sq_linear`` on transposed weights and ``sq_linear_weight_grad``.  Dropout masks are drawn with torch's device RNG
(no stream of the reference's CPU generator can be reproduced on another device anyway); everything else is C calls.
There is no CPU fallback: tensors must live on the MI355X."""
import os
import time

import numpy as np
import torch
from torch import nn

from . import _lib

try:
    from huggingface_hub import PyTorchModelHubMixin
except Exception:                                     # pragma: no cover
    class PyTorchModelHubMixin:  # type: ignore
        pass


def _up(n, m):
    return (n + m - 1) // m * m


class _He2rnaFn(torch.autograd.Function):
    """x [B, C, N] f32 -> [B, G]: forward_fixed_k for one k (scale 1) or the eval mean over ks (scale 1 / len(ks))."""

    @staticmethod
    def forward(ctx, x, ks, scale, p_drop, input_dim, ws, *params):
        lib = _lib.lib()
        dev = x.device
        B, C, N = x.shape
        M = B * N
        st = _lib.stream_ptr(dev)
        xt = x.detach().to(torch.float32).transpose(1, 2).contiguous().view(M, C)        # free when x came from 'b c f -> b f c'
        mask = torch.empty(M, dtype=torch.float32, device=dev)
        _lib.check(lib.sq_he2rna_tile_mask(_lib.ptr(xt), M, C, _lib.ptr(mask), st))
        n_layers = len(params) // 2
        # every activation / weight is zero-padded to a multiple of 8 columns: the GEMM engine wants 16-byte rows, and
        # the reference's default layers=[1] has a hidden width of ONE
        kin = _up(input_dim, 8)
        a = torch.zeros(M, kin, dtype=torch.float32, device=dev)
        a[:, :input_dim] = xt[:, C - input_dim:]
        acts, wpads, drops = [a], [], []
        for i in range(n_layers):
            w, b = params[2 * i], params[2 * i + 1]
            n_out, k_in = w.shape[0], w.shape[1]
            wp = torch.zeros(n_out, acts[-1].shape[1], dtype=torch.float32, device=dev)
            wp[:, :k_in] = w.detach().view(n_out, k_in)
            ld = _up(n_out, 8)
            out = torch.zeros(M, ld, dtype=torch.float32, device=dev)
            last = i + 1 == n_layers
            bias = b.detach().float().contiguous()
            _lib.check(lib.sq_linear(_lib.SQ_F32, _lib.ptr(acts[-1]), acts[-1].shape[1], _lib.ptr(wp), wp.shape[1], _lib.ptr(bias),
                                     None, 0, _lib.SQ_F32, 0 if last else 2, _lib.ptr(out), _lib.SQ_F32, ld, M, n_out, wp.shape[1],
                                     _lib.ptr(ws), ws.numel(), st))
            dm = None
            if not last and p_drop > 0.0:
                dm = (torch.rand(M, ld, device=dev) >= p_drop).to(torch.float32) / (1.0 - p_drop)
                out.mul_(dm)
            wpads.append(wp)
            drops.append(dm)
            acts.append(out)
        G = params[-2].shape[0]
        ks_arr = np.asarray(ks, dtype=np.int32)
        pred = torch.empty(B, G, dtype=torch.float32, device=dev)
        _lib.check(lib.sq_he2rna_topk_mean(_lib.ptr(acts[-1]), acts[-1].shape[1], _lib.ptr(mask), ks_arr.ctypes.data, len(ks_arr), float(scale),
                                           _lib.ptr(pred), B, N, G, st))
        ctx.saved = (acts, wpads, drops, mask, ks_arr, float(scale), (B, C, N, input_dim), ws, [params[2 * i].shape[1] for i in range(n_layers)])
        return pred

    @staticmethod
    def backward(ctx, gout):
        lib = _lib.lib()
        acts, wpads, drops, mask, ks_arr, scale, (B, C, N, input_dim), ws, kins = ctx.saved
        dev = gout.device
        M = B * N
        st = _lib.stream_ptr(dev)
        n_layers = len(wpads)
        G = wpads[-1].shape[0]
        dy = torch.zeros_like(acts[-1])
        _lib.check(lib.sq_he2rna_topk_mean_bwd(_lib.ptr(acts[-1]), acts[-1].shape[1], _lib.ptr(mask), ks_arr.ctypes.data, len(ks_arr), scale,
                                               _lib.ptr(gout.detach().float().contiguous()), _lib.ptr(dy), dy.shape[1], B, N, G, st))
        grads = [None] * (2 * n_layers)
        for i in range(n_layers - 1, -1, -1):
            wp, a_in = wpads[i], acts[i]
            n_out, k_pad = wp.shape
            dw = torch.empty(n_out, k_pad, dtype=torch.float32, device=dev)
            db = torch.empty(n_out, dtype=torch.float32, device=dev)
            _lib.check(lib.sq_linear_weight_grad(_lib.SQ_F32, _lib.ptr(dy), dy.shape[1], _lib.ptr(a_in), k_pad, _lib.ptr(dw), k_pad, _lib.ptr(db),
                                                 n_out, k_pad, M, _lib.ptr(ws), ws.numel(), st))
            grads[2 * i] = dw[:, :kins[i]].unsqueeze(-1)
            grads[2 * i + 1] = db
            if i > 0 or ctx.needs_input_grad[0]:
                wt = torch.zeros(k_pad, dy.shape[1], dtype=torch.float32, device=dev)      # W^T, contraction over the padded outputs
                wt[:, :n_out] = wp.t()
                dx = torch.empty(M, k_pad, dtype=torch.float32, device=dev)
                _lib.check(lib.sq_linear(_lib.SQ_F32, _lib.ptr(dy), dy.shape[1], _lib.ptr(wt), wt.shape[1], None, None, 0, _lib.SQ_F32, 0,
                                         _lib.ptr(dx), _lib.SQ_F32, k_pad, M, k_pad, dy.shape[1], _lib.ptr(ws), ws.numel(), st))
                if i > 0:
                    dx.mul_((a_in > 0).to(torch.float32))          # ReLU (a dropped unit is 0 here and gets no gradient either way)
                    if drops[i - 1] is not None:
                        dx.mul_(drops[i - 1])
                dy = dx
        gx = None
        if ctx.needs_input_grad[0]:
            gx = torch.zeros(B, N, C, dtype=torch.float32, device=dev)
            gx[:, :, C - input_dim:] = dy[:, :input_dim].view(B, N, input_dim)
            gx = gx.transpose(1, 2)
        return (gx, None, None, None, None, None) + tuple(grads)


class HE2RNA(nn.Module, PyTorchModelHubMixin):
    """he2rna.py:42-106.  ``nonlin`` other than ReLU is not supported on the device path (the reference never passes one)."""

    def __init__(self, input_dim, output_dim, layers=[1], nonlin=None, ks=[10], dropout=0.5, device="cpu", bias_init=None, **kwargs):
        super().__init__()
        if nonlin is not None and not isinstance(nonlin, nn.ReLU):
            raise NotImplementedError("HE2RNA on the HIP engine fuses ReLU into the layer launches; other activations are not implemented")
        self.input_dim = input_dim
        self.output_dim = output_dim
        dims = [input_dim] + list(layers) + [output_dim]
        self.n_layers = len(dims) - 1
        for i in range(self.n_layers):                      # parameter containers with the reference's names and shapes
            setattr(self, "conv" + str(i), nn.Conv1d(dims[i], dims[i + 1], kernel_size=1, stride=1, bias=True))
        if bias_init is not None:
            getattr(self, "conv" + str(self.n_layers - 1)).bias = bias_init
        self.ks = np.array(ks)
        self.p_drop = float(dropout)
        self.device = device
        self.to(device)
        self._ws = None

    def _params(self):
        out = []
        for i in range(self.n_layers):
            c = getattr(self, "conv" + str(i))
            out += [c.weight, c.bias]
        return out

    def _workspace(self, dev):
        if self._ws is None or self._ws.device != dev:
            self._ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
        return self._ws

    def __getstate__(self):
        """``torch.save(model)`` (he2rna.py:261 pickles the whole module): the 64 MiB device workspace is scratch, not state."""
        state = self.__dict__.copy()
        state["_ws"] = None
        return state

    def _run(self, x, ks, scale, training):
        _lib.require_gpu()
        if not x.is_cuda:
            raise _lib.SequoiaHipError("HE2RNA needs CUDA tensors (no CPU fallback)")
        if x.dim() != 3 or x.shape[1] < self.input_dim:
            raise ValueError(f"HE2RNA: x must be [batch, channels >= {self.input_dim}, tiles], got {tuple(x.shape)}")
        return _He2rnaFn.apply(x, tuple(int(k) for k in ks), scale, self.p_drop if training else 0.0, self.input_dim,
                               self._workspace(x.device), *self._params())

    def forward(self, x):
        if self.training:
            k = int(np.random.choice(self.ks))              # he2rna.py:85
            return self._run(x, [k], 1.0, True)
        return self._run(x, self.ks, 1.0 / len(self.ks), False)        # the mean over ks (:88-91) in one launch

    def forward_fixed_k(self, x, k):
        return self._run(x, [int(k)], 1.0, self.training)

    def conv(self, x):
        """Per-tile scores [B, G, N] (he2rna.py:101-106); the fused forward never materialises this layout."""
        eye = _ScoresFn.apply(x, self.p_drop if self.training else 0.0, self.input_dim, self._workspace(x.device), *self._params())
        return eye


class _ScoresFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p_drop, input_dim, ws, *params):
        lib = _lib.lib()
        dev = x.device
        B, C, N = x.shape
        M = B * N
        st = _lib.stream_ptr(dev)
        xt = x.detach().to(torch.float32).transpose(1, 2).contiguous().view(M, C)
        a = torch.zeros(M, _up(input_dim, 8), dtype=torch.float32, device=dev)
        a[:, :input_dim] = xt[:, C - input_dim:]
        n_layers = len(params) // 2
        for i in range(n_layers):
            w, b = params[2 * i], params[2 * i + 1]
            n_out, k_in = w.shape[0], w.shape[1]
            wp = torch.zeros(n_out, a.shape[1], dtype=torch.float32, device=dev)
            wp[:, :k_in] = w.detach().view(n_out, k_in)
            out = torch.zeros(M, _up(n_out, 8), dtype=torch.float32, device=dev)
            last = i + 1 == n_layers
            bias = b.detach().float().contiguous()
            _lib.check(lib.sq_linear(_lib.SQ_F32, _lib.ptr(a), a.shape[1], _lib.ptr(wp), wp.shape[1], _lib.ptr(bias),
                                     None, 0, _lib.SQ_F32, 0 if last else 2, _lib.ptr(out), _lib.SQ_F32, out.shape[1], M, n_out, wp.shape[1],
                                     _lib.ptr(ws), ws.numel(), st))
            if not last and p_drop > 0.0:
                out.mul_((torch.rand_like(out) >= p_drop).to(torch.float32) / (1.0 - p_drop))
            a = out
        G = params[-2].shape[0]
        return a[:, :G].reshape(B, N, G).transpose(1, 2)

    @staticmethod
    def backward(ctx, g):
        raise NotImplementedError("HE2RNA.conv is an inspection helper; train through forward / forward_fixed_k")


# ---------------------------------------------------------------------------------------------
# loops (he2rna.py:108-318)
# ---------------------------------------------------------------------------------------------
def compute_correlations(labels, preds):
    """he2rna.py:140-149: mean Pearson r over the genes whose target varies, NaN r dropped."""
    rs = []
    for i in range(labels.shape[1]):
        y = labels[:, i]
        if len(np.unique(y)) > 1:
            rs.append(np.corrcoef(y, preds[:, i])[0, 1])
    rs = np.asarray(rs)
    return np.mean(rs[~np.isnan(rs)])


def _batches(model, loader):
    for x, y, wsi, proj in loader:
        x = x.float().to(model.device).transpose(1, 2)          # 'b c f -> b f c' (he2rna.py:118): channels x tiles
        yield x, y, wsi, proj


def training_epoch(model, dataloader, optimizer):
    """he2rna.py:108-127."""
    model.train()
    loss_fn = nn.MSELoss()
    losses = []
    for x, y, _, _ in _batches(model, dataloader):
        pred = model(x)
        loss = loss_fn(pred, y.float().to(model.device))
        losses.append(float(loss.detach()))
        optimizer.zero_grad()
        loss.backward()
        optimizer.step()
    return float(np.mean(losses))


def evaluate(model, dataloader):
    """he2rna.py:151-173: validation loss on the raw predictions, correlations on ReLU(pred)."""
    model.eval()
    loss_fn = nn.MSELoss()
    losses, preds, labels = [], [], []
    with torch.no_grad():
        for x, y, _, _ in _batches(model, dataloader):
            pred = model(x)
            labels.append(np.asarray(y))
            losses.append(float(loss_fn(pred, torch.as_tensor(y).float().to(model.device))))
            preds.append(torch.relu(pred).cpu().numpy())
    preds, labels = np.concatenate(preds), np.concatenate(labels)
    return float(np.mean(losses)), compute_correlations(labels, preds)


def he2rna_predict(model, dataloader):
    """he2rna.py:175-197."""
    model.eval()
    preds, wsis, projs, labels = [], [], [], []
    with torch.no_grad():
        for x, y, wsi, proj in _batches(model, dataloader):
            preds.append(torch.relu(model(x)).cpu().numpy())
            wsis.append(wsi)
            projs.append(proj)
            labels.append(np.asarray(y))
    return np.concatenate(preds, 0), np.concatenate(labels, 0), np.concatenate(wsis, 0), np.concatenate(projs, 0)


def fit(model, lr, train_loader, valid_loader, test_loader, params={}, fold=None, optimizer=None, path=None, verbose=True):
    """he2rna.py:217-318: Adam(lr) unless an optimizer is given, up to ``max_epochs`` (200) epochs; with a validation
    loader the model with the best mean correlation is kept (``model[_fold].pt``) and training stops after ``patience``
    (100) epochs without improvement; returns the test predictions when a test loader is given, else the model.
    (The reference's wandb calls inside the validation branch reference an undefined ``args``; they are not mirrored.)"""
    if path is not None and not os.path.exists(path):
        os.mkdir(path)
    cfg = {"max_epochs": 200, "patience": 100}
    cfg.update(params)
    if optimizer is None:
        optimizer = torch.optim.Adam(list(model.parameters()), lr=lr, weight_decay=0.0)
    name = "model" if fold is None else "model_" + str(fold)
    best = 0
    if valid_loader is not None:
        _, best = evaluate(model, valid_loader)
        if np.isnan(best):
            best = 0
    since_best, t0 = 0, time.time()
    try:
        for e in range(cfg["max_epochs"]):
            since_best += 1
            train_loss = training_epoch(model, train_loader, optimizer)
            if verbose:
                print("Epoch {}/{} - {:.2f}s".format(e + 1, cfg["max_epochs"], time.time() - t0))
            t0 = time.time()
            if valid_loader is not None:
                valid_loss, score = evaluate(model, valid_loader)
                if verbose:
                    print("loss: {:.4f}, val loss: {:.4f}".format(train_loss, valid_loss))
                    print("correlations: {:.3f}".format(score))
                if score > best:
                    since_best, best = 0, score
                    if path is not None:
                        torch.save(model, os.path.join(path, name + ".pt"))
                if since_best == cfg["patience"]:
                    if verbose:
                        print("Early stopping at epoch {}".format(e + 1))
                    break
    except KeyboardInterrupt:
        pass
    if path is not None and os.path.exists(os.path.join(path, name + ".pt")):
        model = torch.load(os.path.join(path, name + ".pt"), weights_only=False)
    elif path is not None:
        torch.save(model, os.path.join(path, name + ".pt"))
    if test_loader is not None:
        return he2rna_predict(model, test_loader)
    return model
