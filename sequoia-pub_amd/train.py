"""Training / evaluation loops -- host-side mirror of /root/reference/src/vit.py:117-311
(``train``, ``evaluate``, ``predict``, ``smape``) and of the metrics it calls per batch
(``src/he2rna.py:140-149 compute_correlations``, ``sklearn mean_absolute_error``).

What changes versus the reference is where the work runs, not what is computed:
  * forward / backward / MSE / AdamW are libsequoia_hip calls on one flat parameter buffer
    (``FusedTrainStep``); ``loss.backward(); optimizer.step()`` with a torch optimizer also works
    (``ViS`` is an autograd Function over the same C calls).
  * MAE and the mean per-gene Pearson score are reduced on the device (``sq_batch_metrics``)
    instead of three D2H copies + 20 820 ``np.corrcoef`` calls per batch (1.2 s/batch on CPU).
  * with ``torch.distributed`` initialised, the flat gradient is all-reduced over RCCL once per
    step and epoch scalars / stop decisions are reduced so every rank takes the same branch.
Checkpoint naming, save/stop rules and return values follow the reference line by line.
"""
import ctypes
import math
import os

import numpy as np
import torch
import torch.distributed as dist

from . import _lib


def smape(A, F):
    """vit.py:32-33."""
    return 100 / len(A) * np.sum(2 * np.abs(F - A) / (np.abs(A) + np.abs(F)))


# ---------------------------------------------------------------------------------------------
# device-side pieces
# ---------------------------------------------------------------------------------------------
def grad_buckets(model):
    """[(lo, hi)] element ranges of the flat gradient in the order the backward pass completes them
    (sq_vis_grad_buckets); models without bucket support (the ViT baseline) are one bucket."""
    if model._C_BWD != "sq_vis_backward":
        return [(0, model.flat.numel())]
    cap = model.cfg.depth + 1
    lo, hi = (ctypes.c_int64 * cap)(), (ctypes.c_int64 * cap)()
    n = _lib.lib().sq_vis_grad_buckets(ctypes.byref(model.cfg), lo, hi, cap)
    if n < 0:
        _lib.check(n)
    return [(int(lo[i]), int(hi[i])) for i in range(n)]


def vis_backward(model, grad_out, batch, need_x_grad, bucket_events=None):
    """sq_vis_backward on the workspace the matching forward saved.  Returns (grad_flat, grad_x).
    bucket_events: torch.cuda.Event per grad_buckets() entry, recorded mid-pass as each bucket becomes final."""
    dev = model.flat.device
    grad_out = grad_out.to(dev, torch.float32).contiguous()
    need = getattr(_lib.lib(), model._C_BWS)(ctypes.byref(model.cfg), model.compute_dtype, batch)
    if getattr(model, "_bws", None) is None or model._bws.numel() < need or model._bws.device != dev:
        model._bws = torch.empty(need, dtype=torch.uint8, device=dev)
    gflat = getattr(model, "_gflat", None)
    if gflat is None or gflat.shape != model.flat.shape or gflat.device != dev:
        gflat = model._gflat = torch.zeros_like(model.flat.detach())
    gx = torch.empty(batch, model.cfg.num_clusters, model._dim(), device=dev) if need_x_grad else None
    ws = model._ws
    with torch.cuda.device(dev):
        args = (ctypes.byref(model.cfg), model.compute_dtype, _lib.ptr(model.flat), _lib.ptr(model._params_lp()),
                _lib.ptr(grad_out), _lib.ptr(gflat), _lib.ptr(gx), batch, _lib.ptr(ws), ws.numel(),
                _lib.ptr(model._bws), model._bws.numel(), _lib.stream_ptr(dev))
        if bucket_events is not None and model._C_BWD == "sq_vis_backward":
            handles = (ctypes.c_void_p * len(bucket_events))(*[e.cuda_event for e in bucket_events])
            _lib.check(_lib.lib().sq_vis_backward_buckets(*args, handles, len(bucket_events)))
        else:
            _lib.check(getattr(_lib.lib(), model._C_BWD)(*args))
            if bucket_events is not None:
                for e in bucket_events:
                    e.record(torch.cuda.current_stream(dev))
    return gflat, gx


def _scratch(model):
    s = getattr(model, "_tscratch", None)
    need = _lib.lib().sq_train_scratch_bytes(model.cfg.num_outputs)
    if s is None or s.numel() < need or s.device != model.flat.device:
        s = model._tscratch = torch.empty(need, dtype=torch.uint8, device=model.flat.device)
    return s


def mse_loss_grad(model, pred, target, grad_scale=None, want_grad=True):
    """nn.MSELoss() value (device scalar tensor) and d loss / d pred."""
    n = pred.numel()
    grad = torch.empty_like(pred) if want_grad else None
    loss = torch.empty(1, dtype=torch.float32, device=pred.device)
    scale = (2.0 / n) if grad_scale is None else grad_scale
    _lib.check(_lib.lib().sq_mse_loss_grad(_lib.ptr(pred), _lib.ptr(target), n, scale, _lib.ptr(grad), _lib.ptr(loss),
                                           _lib.ptr(_scratch(model)), _lib.stream_ptr(pred.device)))
    return loss, grad


def batch_metrics(model, pred, target):
    """(MAE, mean per-gene Pearson, n_genes) of one batch as a device tensor [3] (vit.py:167-168)."""
    out = torch.empty(3, dtype=torch.float32, device=pred.device)
    B, G = pred.shape
    _lib.check(_lib.lib().sq_batch_metrics(_lib.ptr(pred), _lib.ptr(target), B, G, _lib.ptr(out),
                                           _lib.ptr(_scratch(model)), _lib.stream_ptr(pred.device)))
    return out


class FusedTrainStep:
    """forward + MSE + backward + (RCCL all-reduce) + AdamW as C calls on the flat buffers.

    Equivalent to vit.py:163-180 with ``torch.optim.AdamW(lr, amsgrad=False, weight_decay=0.)``
    (main.py:180-183).  With world_size > 1 the per-rank gradient of the local batch is summed over
    ranks and the loss normalised by the GLOBAL element count, so the update equals the
    single-device update on the concatenated batch.
    """

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, world_size=1, metrics=False):
        self.model = model
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.world = world_size
        self.step_count = 0
        self.metrics = metrics
        self.exp_avg = torch.zeros_like(model.flat.detach())
        self.exp_avg_sq = torch.zeros_like(model.flat.detach())
        # gradient all-reduce overlapped with the backward pass: one bucket per layer (+ head), each reduced on
        # a side stream as soon as the backward pass has recorded its event.  SQ_FORCE_BUCKETS=1 exercises the
        # same path on a single rank.
        self.overlap = world_size > 1 or os.environ.get("SQ_FORCE_BUCKETS") == "1"
        if self.overlap:
            dev = model.flat.device
            self.buckets = grad_buckets(model)
            self.comm_stream = torch.cuda.Stream(device=dev)
            self.events = [torch.cuda.Event() for _ in self.buckets]
            with torch.cuda.device(dev):
                for e in self.events:
                    e.record()              # instantiates the hipEvent_t the C side records into

    def step(self, x, target):
        m = self.model
        dev = m.flat.device
        B = x.shape[0]
        pred = m._run_forward(x, save=True)
        n_global = pred.numel() * self.world
        loss, gpred = mse_loss_grad(m, pred, target, grad_scale=2.0 / n_global)
        mets = batch_metrics(m, pred, target) if self.metrics else None
        if not self.overlap:
            gflat, _ = vis_backward(m, gpred, B, False)
        else:
            main = torch.cuda.current_stream(dev)
            gflat, _ = vis_backward(m, gpred, B, False, bucket_events=self.events)
            reduce_ = dist.is_available() and dist.is_initialized()
            for (lo, hi), ev in zip(self.buckets, self.events):
                self.comm_stream.wait_event(ev)
                if reduce_:
                    with torch.cuda.stream(self.comm_stream):
                        dist.all_reduce(gflat[lo:hi], op=dist.ReduceOp.SUM)
            main.wait_stream(self.comm_stream)
        self.step_count += 1
        lp = m._params_lp()
        with torch.no_grad():
            _lib.check(_lib.lib().sq_adamw_step(_lib.ptr(m.flat), _lib.ptr(gflat), _lib.ptr(self.exp_avg),
                                                _lib.ptr(self.exp_avg_sq), _lib.ptr(lp), m.flat.numel(), self.lr,
                                                self.betas[0], self.betas[1], self.eps, self.wd, self.step_count, 1.0,
                                                _lib.stream_ptr(dev)))
        # the kernel refreshed the bf16 shadow in the same pass (flat._version is unchanged by C-side writes)
        return loss, pred, mets


# ---------------------------------------------------------------------------------------------
# loops (vit.py:117-311)
# ---------------------------------------------------------------------------------------------
def _is_empty(image):
    return isinstance(image, list) and len(image) == 0


def _all_mean(vals, device):
    """Mean over all batches of all ranks (each rank contributes its own list)."""
    t = torch.tensor([float(np.sum(vals)), float(len(vals))], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t)
    return float(t[0] / t[1]) if float(t[1]) > 0 else float("nan")


def train(model, dataloaders, optimizer=None, accelerator=None,
          num_epochs=200, save_dir='exp/', patience=20,
          run=None, verbose=True, phases=['train', 'val'], split=None,
          save_on='loss', stop_on='loss', delta=0.5, lr=1e-3):
    """Same contract as vit.py:117-243.  ``optimizer`` may be a torch optimizer over
    ``model.parameters()`` (used through autograd) or None (fused AdamW step, the fast path)."""
    rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    if save_dir is not None and not os.path.exists(save_dir) and rank == 0:
        os.mkdir(save_dir)
    if split:                                   # NB falsy for split 0 -> no suffix (vit.py:124)
        save_path = os.path.join(save_dir, f'model_best_{split}.pt')
    else:
        save_path = os.path.join(save_dir, 'model_best.pt')

    fused = FusedTrainStep(model, lr=lr, world_size=world, metrics=True) if optimizer is None else None
    dev = model.flat.device
    epoch_since_best = 0
    best_loss = np.inf
    early_stop_on_loss_triggered = 0
    epoch_since_best_score = 0
    best_score = 0
    epoch_since_ok_loss = 0

    def save():
        if rank == 0:
            torch.save(model.state_dict(), save_path)

    for epoch in range(num_epochs):
        for phase in phases:
            model.train() if phase == 'train' else model.eval()
            losses, maes, scores = [], [], []
            for s, (image, rna_data, _, _) in enumerate(dataloaders[phase]):
                if _is_empty(image):
                    continue
                image = image.to(dev)
                rna_data = rna_data.to(dev)
                if phase == 'train' and fused is not None:
                    loss, pred, mets = fused.step(image, rna_data)
                else:
                    with torch.set_grad_enabled(phase == 'train'):
                        pred = model(image)
                    loss, gpred = mse_loss_grad(model, pred.detach(), rna_data, want_grad=phase == 'train')
                    mets = batch_metrics(model, pred.detach(), rna_data)
                    if phase == 'train':
                        optimizer.zero_grad()
                        pred.backward(gpred)
                        optimizer.step()
                vals = torch.cat([loss, mets[:2]]).cpu().numpy()          # one small D2H per batch
                losses.append(vals[0]); maes.append(vals[1]); scores.append(vals[2])
            L, A, S = _all_mean(losses, dev), _all_mean(maes, dev), _all_mean(scores, dev)
            suffix = 'id' if phase == 'val' else ''
            if run and rank == 0:
                run.log({'epoch': epoch, f'score {phase}{suffix} {split}': S})
                run.log({'epoch': epoch, f'{phase}{suffix} loss fold {split}': L})
                run.log({'epoch': epoch, f'{phase}{suffix} mae fold {split}': A})
            if verbose and rank == 0:
                print(f'Epoch {epoch}: {phase} loss {L} mae {A}')

            if (phase == 'val') or (len(phases) == 1):
                if early_stop_on_loss_triggered == 1:
                    if L < (best_loss + delta):
                        epoch_since_ok_loss = 0
                    else:
                        epoch_since_ok_loss += 1
                if L < best_loss:
                    best_loss = L
                    epoch_since_best = 0
                    if save_on == 'loss':
                        save()
                    elif (save_on == 'loss+corr') and (early_stop_on_loss_triggered == 0):
                        save()
                else:
                    epoch_since_best += 1
                if S > best_score:
                    best_score = S
                    epoch_since_best_score = 0
                    if (save_on == 'loss+corr') and (early_stop_on_loss_triggered == 1):
                        save()
                        if rank == 0:
                            print(f'Saved model on loss+corr at epoch {epoch} of better score and loss within {delta} of optimal loss')
                else:
                    epoch_since_best_score += 1

        if epoch_since_best == patience:
            early_stop_on_loss_triggered = 1
            if stop_on == 'loss':
                if rank == 0:
                    print(f'Early stopping at epoch {epoch}!')
                break
        if stop_on == 'loss+corr':
            if (early_stop_on_loss_triggered == 1) and (epoch_since_best_score == patience):
                if rank == 0:
                    print(f'Early stopping at epoch {epoch} because neither loss nor score is improving anymore!')
                break
            if (early_stop_on_loss_triggered == 1) and (epoch_since_ok_loss == patience):
                if rank == 0:
                    print(f'Early stopping at epoch {epoch} because loss is not within {delta} of best loss anymore!')
                break
    return model          # last-epoch model, not the best checkpoint (vit.py:243)


def evaluate(model, dataloader, run=None, verbose=True, suff=''):
    """vit.py:245-291: returns (preds, real, wsis, projs) as numpy arrays."""
    model.eval()
    dev = model.flat.device
    losses, maes, smapes, preds, real, wsis, projs = [], [], [], [], [], [], []
    for image, rna_data, wsi_file_name, tcga_project in dataloader:
        if _is_empty(image):
            continue
        image = image.to(dev)
        rna_data = rna_data.to(dev)
        wsis.append(wsi_file_name)
        projs.append(tcga_project)
        with torch.no_grad():
            pred = model(image)
        loss, _ = mse_loss_grad(model, pred, rna_data, want_grad=False)
        mets = batch_metrics(model, pred, rna_data)
        p_np, r_np = pred.cpu().numpy(), rna_data.cpu().numpy()
        preds.append(p_np)
        real.append(r_np)
        losses.append(float(loss))
        maes.append(float(mets[0]))
        smapes.append(smape(r_np, p_np))
    losses, maes_m, smapes = np.mean(losses), np.mean(maes), np.mean(smapes)
    if run:
        run.log({'test_loss' + suff: losses})
        run.log({'test_MAE' + suff: maes_m})
        run.log({'test_MAPE' + suff: smapes})
    if verbose:
        print(f'Test loss: {losses}')
        print(f'Test MAE: {maes[-1]}')          # the reference prints the LAST batch's mae (vit.py:283)
        print(f'Test MAPE: {smapes}')
    return (np.concatenate(preds, axis=0), np.concatenate(real, axis=0),
            np.concatenate(wsis, axis=0), np.concatenate(projs, axis=0))


def predict(model, dataloader, run=None, verbose=True):
    """vit.py:293-311: returns (preds, wsis, projs)."""
    model.eval()
    dev = model.flat.device
    preds, wsis, projs = [], [], []
    for image, rna_data, wsi_file_name, tcga_project in dataloader:
        if _is_empty(image):
            continue
        wsis.append(wsi_file_name)
        projs.append(tcga_project)
        with torch.no_grad():
            preds.append(model(image.to(dev)).cpu().numpy())
    return np.concatenate(preds, axis=0), np.concatenate(wsis, axis=0), np.concatenate(projs, axis=0)


def smoke_check():
    """Tiny fwd+bwd+AdamW step on cuda:0 against the oracle (called from __graft_entry__.smoke)."""
    from oracle import vis_oracle
    from .vis import ViS
    cfg = dict(num_outputs=200, input_dim=256, depth=2, nheads=4, dimensions_f=64, dimensions_s=64, dimensions_c=64)
    sd = vis_oracle.perturb_norm_params(vis_oracle.init_vis_state_dict(**cfg, seed=1), seed=2)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(3, 100, 256, generator=g)
    y = torch.rand(3, 200, generator=g) * 8
    loss_ref, _, grads_ref = vis_oracle.vis_loss_and_grads(sd, x, y)
    m = ViS(**cfg, device="cuda:0")
    m.load_state_dict(sd)
    m.to("cuda:0")
    pred = m(x.cuda())
    loss, gpred = mse_loss_grad(m, pred.detach(), y.cuda())
    pred.backward(gpred)
    gv = m.grad_views(m.flat.grad)
    worst = 0.0
    for k, gr in grads_ref.items():
        e = float((gv[k].cpu() - gr).abs().max() / max(float(gr.abs().max()), 1e-12))
        worst = max(worst, e)
    print(f"smoke: ViS backward fp32 worst per-tensor rel err vs oracle {worst:.3e}; loss {float(loss):.6f} vs {float(loss_ref):.6f}")
    assert worst < 1e-3 and abs(float(loss) - float(loss_ref)) < 1e-4 * abs(float(loss_ref))
