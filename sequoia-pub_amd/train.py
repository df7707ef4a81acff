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
Checkpoint naming, save/stop rules (``CheckpointPolicy``) and return values are the reference's.
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


def _pair_f32(pred, target):
    """The C entry points read packed f32 [B, G] arrays through raw pointers: bring both operands to that form on
    pred's device (the reference's loss would up-cast / raise; garbage or out-of-bounds reads are not an option)."""
    if not pred.is_cuda:
        raise _lib.SequoiaHipError("loss / metrics run on the GPU: pred is on the CPU (no CPU fallback)")
    pred = pred.detach().to(torch.float32).contiguous()
    target = torch.as_tensor(target).detach().to(pred.device, torch.float32).contiguous()
    if target.shape != pred.shape:
        raise ValueError(f"target shape {tuple(target.shape)} != prediction shape {tuple(pred.shape)}")
    return pred, target


def mse_loss_grad(model, pred, target, grad_scale=None, want_grad=True):
    """nn.MSELoss() value (device scalar tensor) and d loss / d pred."""
    pred, target = _pair_f32(pred, target)
    n = pred.numel()
    grad = torch.empty_like(pred) if want_grad else None
    loss = torch.empty(1, dtype=torch.float32, device=pred.device)
    scale = (2.0 / n) if grad_scale is None else grad_scale
    _lib.check(_lib.lib().sq_mse_loss_grad(_lib.ptr(pred), _lib.ptr(target), n, scale, _lib.ptr(grad), _lib.ptr(loss),
                                           _lib.ptr(_scratch(model)), _lib.stream_ptr(pred.device)))
    return loss, grad


def batch_metrics(model, pred, target):
    """(MAE, mean per-gene Pearson, n_genes) of one batch as a device tensor [3] (vit.py:167-168)."""
    pred, target = _pair_f32(pred, target)
    if pred.dim() != 2:
        raise ValueError(f"batch_metrics wants [B, G] predictions, got {tuple(pred.shape)}")
    out = torch.empty(3, dtype=torch.float32, device=pred.device)
    B, G = pred.shape
    _lib.check(_lib.lib().sq_batch_metrics(_lib.ptr(pred), _lib.ptr(target), B, G, _lib.ptr(out),
                                           _lib.ptr(_scratch(model)), _lib.stream_ptr(pred.device)))
    return out


class FusedTrainStep:
    """forward + MSE + backward + (RCCL all-reduce) + AdamW as C calls on the flat buffers.

    Equivalent to vit.py:163-180 with ``torch.optim.AdamW(lr, amsgrad=False, weight_decay=0.)``
    (main.py:180-183).  With world_size > 1 the per-rank gradient of the local batch is summed over
    ranks and the loss normalised by the GLOBAL element count.  With the default fp32 wire format the update
    equals the single-device update on the concatenated batch up to the summation order (1e-5 of a tensor's
    maximum, tests/test_gpu_ddp.py); with ``grad_exchange='bf16'`` (opt-in: BASELINE config 4's "bf16"
    exchange, half the bytes) every bucket is rounded to bf16 once before the sum and the ranks' partial sums
    are rounded again on the way round the ring -- the exchanged gradient then agrees with the single-device
    one to ~2^-8 of each tensor's maximum (world_size - 1 roundings at worst), all ranks still bit-identical.
    """

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, world_size=1, metrics=False,
                 grad_exchange=None):
        """grad_exchange: 'fp32' | 'bf16' -- the wire format of the gradient all-reduce (world_size > 1).  Default 'fp32'
        (multi-rank numerics == single-rank numerics); 'bf16' is opt-in (BASELINE config 4: "RCCL grad all-reduce over xGMI,
        bf16": half the ring traffic; bench.py's train_kfold workload and ``cli.main --grad_exchange bf16`` ask for it): every
        bucket is cast to bf16 behind its completion event, summed over the ranks in bf16, cast back into the fp32 flat
        gradient AdamW reads; master weights, moments and the local accumulation stay fp32.  Held to the fp32 exchange over 12
        steps on 8 ranks by tests/test_gpu_ddp.py."""
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
        # AdamW per bucket on the communication stream, as each bucket becomes final (SQ_ADAMW_BUCKETS=1; also switches the bucket
        # path on for one rank).  Opt-in: measured SLOWER on one MI355X -- the 1.6 GB HBM-bound update running beside the backward
        # pass's products costs them more than the 0.28 ms pass it removes (config 2: 3.26 ms without, 3.33 ms with)
        self.adamw_buckets = os.environ.get("SQ_ADAMW_BUCKETS", "0") == "1" and model._C_BWD == "sq_vis_backward"
        self.overlap = self.overlap or self.adamw_buckets
        if grad_exchange is None:
            grad_exchange = "fp32"
        if grad_exchange not in ("bf16", "fp32"):
            raise ValueError(f"grad_exchange {grad_exchange!r}: 'bf16' or 'fp32'")
        self.grad_exchange = grad_exchange
        self.exchange_bytes_per_step = 0              # payload handed to all_reduce per step (all buckets), set below
        self._wire = None
        if self.overlap:
            dev = model.flat.device
            self.buckets = grad_buckets(model)
            self.comm_stream = torch.cuda.Stream(device=dev)
            self.events = [torch.cuda.Event() for _ in self.buckets]
            n_el = sum(hi - lo for lo, hi in self.buckets)
            self.exchange_bytes_per_step = (n_el * (2 if grad_exchange == "bf16" else 4)) if world_size > 1 else 0
            if grad_exchange == "bf16" and world_size > 1:      # (one rank exchanges nothing: its gradient is never rounded)
                self._wire = torch.empty(model.flat.numel(), dtype=torch.bfloat16, device=dev)     # bucket i travels as _wire[lo:hi]
            with torch.cuda.device(dev):
                for e in self.events:
                    e.record()              # instantiates the hipEvent_t the C side records into
        # exchange-step timing (every 16th step, read back one sample later so that nothing waits): how long the backward
        # pass ran, how long the bucketed all-reduce was busy behind it, and how much of it stuck out past the backward
        # pass -- what a first multi-GPU run needs to explain its own scaling (timing_report(); train() prints it per epoch)
        self._tm = dict(n=0, backward_ms=0.0, allreduce_span_ms=0.0, exposed_ms=0.0, pending=None)

    def step(self, x, target, n_global=None):
        """One optimizer step.  x [B, 100, D], target [B, G].  n_global: number of target elements of this step over
        ALL ranks (default: every rank holds a batch like this one); a rank whose batch is empty passes x=None and
        still joins the gradient exchange with zeros, so the collectives of the ranks stay matched.
        Returns (loss, pred, metrics) of the local batch, or None for an empty local batch."""
        m = self.model
        dev = m.flat.device
        empty = x is None
        if empty and not self.overlap:
            raise ValueError("FusedTrainStep.step: empty batch on a single rank (the loop skips those, vit.py:159)")
        if not empty:
            B = x.shape[0]
            pred = m._run_forward(x, save=True)
            if n_global is None:
                n_global = pred.numel() * self.world
            loss, gpred = mse_loss_grad(m, pred, target, grad_scale=2.0 / n_global)
            mets = batch_metrics(m, pred, target) if self.metrics else None
        if not self.overlap:
            gflat, _ = vis_backward(m, gpred, B, False)
        else:
            main = torch.cuda.current_stream(dev)
            if empty:
                gflat = getattr(m, "_gflat", None)
                if gflat is None or gflat.shape != m.flat.shape or gflat.device != dev:
                    gflat = m._gflat = torch.zeros_like(m.flat.detach())
                gflat.zero_()
                for e in self.events:
                    e.record(main)
            else:
                sample = self.step_count % 16 == 0
                if sample:
                    self._collect_timing()
                    tb0, tb1, tc0, tc1 = (torch.cuda.Event(enable_timing=True) for _ in range(4))
                    tb0.record(main)
                gflat, _ = vis_backward(m, gpred, B, False, bucket_events=self.events)
                if sample:
                    tb1.record(main)
            reduce_ = dist.is_available() and dist.is_initialized() and self.world > 1
            if self.adamw_buckets:
                self.step_count += 1
                lp = m._params_lp()
            for i, ((lo, hi), ev) in enumerate(zip(self.buckets, self.events)):
                self.comm_stream.wait_event(ev)
                if not empty and sample and i == 0:
                    tc0.record(self.comm_stream)
                if reduce_:                   # (no process group: nothing is exchanged and nothing is rounded)
                    with torch.cuda.stream(self.comm_stream):
                        if self._wire is None:
                            dist.all_reduce(gflat[lo:hi], op=dist.ReduceOp.SUM)
                        else:                 # bf16 on the wire: pack -> sum over ranks in bf16 -> unpack into the fp32 gradient
                            cs = _lib.stream_ptr(dev)                   # inside the block: the communication stream
                            _lib.check(_lib.lib().sq_cast_f32_to_bf16(_lib.ptr(gflat[lo:hi]), _lib.ptr(self._wire[lo:hi]), hi - lo, cs))
                            dist.all_reduce(self._wire[lo:hi], op=dist.ReduceOp.SUM)
                            _lib.check(_lib.lib().sq_cast_bf16_to_f32(_lib.ptr(self._wire[lo:hi]), _lib.ptr(gflat[lo:hi]), hi - lo, cs))
                if not self.adamw_buckets:
                    continue
                # the bucket's AdamW update (element-wise: identical to one pass over the flat buffer) behind its exchange
                with torch.cuda.stream(self.comm_stream), torch.no_grad():
                    _lib.check(_lib.lib().sq_adamw_step(_lib.ptr(m.flat[lo:hi]), _lib.ptr(gflat[lo:hi]), _lib.ptr(self.exp_avg[lo:hi]),
                                                        _lib.ptr(self.exp_avg_sq[lo:hi]), _lib.ptr(lp[lo:hi]) if lp is not None else None,
                                                        hi - lo, self.lr, self.betas[0], self.betas[1], self.eps, self.wd, self.step_count, 1.0,
                                                        _lib.stream_ptr(dev)))
            if not empty and sample:
                tc1.record(self.comm_stream)
                self._tm["pending"] = (tb0, tb1, tc0, tc1)
            main.wait_stream(self.comm_stream)
            if self.adamw_buckets:
                return None if empty else (loss, pred, mets)
        self.step_count += 1
        lp = m._params_lp()
        with torch.no_grad():
            _lib.check(_lib.lib().sq_adamw_step(_lib.ptr(m.flat), _lib.ptr(gflat), _lib.ptr(self.exp_avg),
                                                _lib.ptr(self.exp_avg_sq), _lib.ptr(lp), m.flat.numel(), self.lr,
                                                self.betas[0], self.betas[1], self.eps, self.wd, self.step_count, 1.0,
                                                _lib.stream_ptr(dev)))
        # the kernel refreshed the bf16 shadow in the same pass (flat._version is unchanged by C-side writes)
        return None if empty else (loss, pred, mets)

    def _collect_timing(self):
        pend = self._tm.get("pending") if hasattr(self, "_tm") else None
        if pend is None:
            return
        tb0, tb1, tc0, tc1 = pend
        tc1.synchronize()                         # a sample from >= 16 steps ago: long complete
        tb1.synchronize()
        self._tm["pending"] = None
        self._tm["n"] += 1
        self._tm["backward_ms"] += tb0.elapsed_time(tb1)
        self._tm["allreduce_span_ms"] += tc0.elapsed_time(tc1)
        self._tm["exposed_ms"] += max(0.0, tb1.elapsed_time(tc1))

    def timing_report(self, reset=True):
        """Mean over the sampled steps since the last report: backward pass, first-bucket-ready -> last all-reduce done on the
        communication stream, and the part of that span behind the end of the backward pass (the exposed exchange time).
        None when the exchange path is off (one rank) or nothing was sampled."""
        if not self.overlap:
            return None
        self._collect_timing()
        n = self._tm["n"]
        if n == 0:
            return None
        out = {k: round(self._tm[k] / n, 4) for k in ("backward_ms", "allreduce_span_ms", "exposed_ms")}
        out["sampled_steps"] = n
        out["wire_format"] = self.grad_exchange
        out["bytes_exchanged_per_step"] = self.exchange_bytes_per_step
        if reset:
            self._tm.update(n=0, backward_ms=0.0, allreduce_span_ms=0.0, exposed_ms=0.0)
        return out


# ---------------------------------------------------------------------------------------------
# loops (vit.py:117-311)
# ---------------------------------------------------------------------------------------------
def _is_empty(image):
    return isinstance(image, list) and len(image) == 0


def _dist_on():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def _all_mean(vals, device):
    """Mean over all batches of all ranks (each rank contributes its own list)."""
    t = torch.tensor([float(np.sum(vals)), float(len(vals))], dtype=torch.float64, device=device)
    if _dist_on():
        dist.all_reduce(t)
    return float(t[0] / t[1]) if float(t[1]) > 0 else float("nan")


class CheckpointPolicy:
    """When to write ``model_best*.pt`` and when to end training -- the decision rules of vit.py:199-242 as a small
    state machine fed one (loss, score) pair per epoch.  Pinned by tests/golden/early_stop.json (events recorded
    from the reference loop on scripted validation curves).

    Two regimes.  Before the loss has gone `patience` epochs without a new minimum ("plateau"), a new minimum is
    a checkpoint.  After the plateau, ``save_on='loss+corr'`` keeps checkpointing on new best SCORES instead, and
    ``stop_on='loss+corr'`` keeps training until the score has been stale for `patience` epochs or the loss has
    stayed `delta` above its minimum for `patience` epochs."""

    def __init__(self, save_on="loss", stop_on="loss", patience=20, delta=0.5):
        self.save_on, self.stop_on, self.patience, self.delta = save_on, stop_on, patience, delta
        self.min_loss, self.max_score = np.inf, 0
        self.age = {"loss": 0, "score": 0, "drift": 0}     # epochs since: loss minimum, score maximum, loss within delta
        self.plateau = False

    def observe(self, loss, score):
        """One validation result.  Returns the list of reasons to checkpoint now ('loss' and / or 'score')."""
        reasons = []
        if self.plateau:
            self.age["drift"] = 0 if loss < self.min_loss + self.delta else self.age["drift"] + 1
        if loss < self.min_loss:
            self.min_loss, self.age["loss"] = loss, 0
            if self.save_on == "loss" or (self.save_on == "loss+corr" and not self.plateau):
                reasons.append("loss")
        else:
            self.age["loss"] += 1
        if score > self.max_score:
            self.max_score, self.age["score"] = score, 0
            if self.save_on == "loss+corr" and self.plateau:
                reasons.append("score")
        else:
            self.age["score"] += 1
        return reasons

    def end_of_epoch(self, epoch):
        """Returns the message of the stop rule that fired, or None to go on."""
        if self.age["loss"] == self.patience:
            self.plateau = True
            if self.stop_on == "loss":
                return f"Early stopping at epoch {epoch}!"
        if self.stop_on == "loss+corr" and self.plateau:
            if self.age["score"] == self.patience:
                return f"Early stopping at epoch {epoch} because neither loss nor score is improving anymore!"
            if self.age["drift"] == self.patience:
                return f"Early stopping at epoch {epoch} because loss is not within {self.delta} of best loss anymore!"
        return None


def _paired_batches(loader, device):
    """Iterate a rank's loader while all ranks agree on the number of steps, yielding (item, n_global): a rank whose loader
    is exhausted keeps yielding None until every rank is done, so collectives stay matched.  ONE small all-reduce per step
    carries both facts the step needs: how many ranks still have items and how many target elements the step's batch has
    over all ranks (ranks may hold ragged or empty batches: collate drops unreadable slides)."""
    it = iter(loader)
    while True:
        item = next(it, None)
        n_local = 0
        if item is not None and not _is_empty(item[0]):
            n_local = item[1].numel()
        if _dist_on():
            t = torch.tensor([0.0 if item is None else 1.0, float(n_local)], dtype=torch.float64, device=device)
            dist.all_reduce(t)
            alive, n_global = (int(v) for v in t.tolist())
            if alive == 0:
                return
        elif item is None:
            return
        else:
            n_global = n_local
        yield item, n_global


def train(model, dataloaders, optimizer=None, accelerator=None,
          num_epochs=200, save_dir='exp/', patience=20,
          run=None, verbose=True, phases=['train', 'val'], split=None,
          save_on='loss', stop_on='loss', delta=0.5, lr=1e-3, grad_exchange=None):
    """Same contract as vit.py:117-243.  ``optimizer`` may be a torch optimizer over
    ``model.parameters()`` (used through autograd) or None (fused AdamW step, the fast path).

    Under torch.distributed every rank feeds its own shard of the training batches; the gradient of the step is
    the gradient of the MSE over the union of the ranks' batches (summed over ranks, normalised by the global
    element count), in both the fused and the torch-optimizer path.  A rank whose batch collated to nothing still
    takes part in the exchange with a zero gradient.  ``grad_exchange`` ('fp32' default | 'bf16'): FusedTrainStep's wire format."""
    rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    if save_dir is not None and not os.path.exists(save_dir) and rank == 0:
        os.mkdir(save_dir)
    # vit.py:124: `if split:` -- fold 0 (and None) get no suffix.  save_dir=None (not a reference mode): the policy runs
    # but no checkpoint file is written (benchmarks)
    save_path = None if save_dir is None else os.path.join(save_dir, f'model_best_{split}.pt' if split else 'model_best.pt')

    fused = FusedTrainStep(model, lr=lr, world_size=world, metrics=True, grad_exchange=grad_exchange) if optimizer is None else None
    dev = model.flat.device
    policy = CheckpointPolicy(save_on, stop_on, patience, delta)
    observing = [ph for ph in phases if ph == 'val'] or (list(phases) if len(phases) == 1 else [])

    for epoch in range(num_epochs):
        for phase in phases:
            training = phase == 'train'
            model.train() if training else model.eval()
            loader = dataloaders[phase]
            reshuffle = getattr(getattr(loader, "sampler", None), "set_epoch", None)
            if reshuffle is not None:
                reshuffle(epoch)                                  # DistributedSampler: a new permutation every epoch
            stats = []                                            # per batch: (loss, mae, score)
            paired = training and world > 1
            for batch in (_paired_batches(loader, dev) if paired else loader):
                n_global = None
                if paired:
                    batch, n_global = batch
                image, rna_data = (batch[0], batch[1]) if batch is not None else ([], None)
                empty = _is_empty(image)
                if (empty and not paired) or n_global == 0:
                    continue                                      # vit.py:159 (under DDP: the batch is empty on EVERY rank -- no optimizer step)
                if not empty:
                    image, rna_data = image.to(dev), rna_data.to(dev)
                if training:
                    if n_global is None:
                        n_global = rna_data.numel()
                    if fused is not None:
                        res = fused.step(None if empty else image, rna_data, n_global=n_global)
                        res = None if res is None else (res[0], res[2])
                    else:
                        res = _autograd_step(model, optimizer, None if empty else image, rna_data, n_global, world)
                    if res is None:
                        continue
                    loss, mets = res
                else:
                    with torch.no_grad():
                        pred = model(image)
                    loss, _ = mse_loss_grad(model, pred, rna_data, want_grad=False)
                    mets = batch_metrics(model, pred, rna_data)
                stats.append(torch.cat([loss, mets[:2]]).cpu().numpy())           # one small D2H per batch
            cols = list(zip(*stats)) if stats else ([], [], [])
            L, A, S = (_all_mean(c, dev) for c in cols)
            tag = 'id' if phase == 'val' else ''
            if run and rank == 0:
                run.log({'epoch': epoch, f'score {phase}{tag} {split}': S})
                run.log({'epoch': epoch, f'{phase}{tag} loss fold {split}': L})
                run.log({'epoch': epoch, f'{phase}{tag} mae fold {split}': A})
            if verbose and rank == 0:
                print(f'Epoch {epoch}: {phase} loss {L} mae {A}')
            if training and fused is not None and world > 1:
                tr = fused.timing_report()
                model.last_exchange_timing = tr                    # bench.py train_kfold prints it in its `check` block
                if tr is not None and verbose and rank == 0:
                    print(f"Epoch {epoch}: gradient exchange per step -- backward {tr['backward_ms']:.3f} ms, bucketed all-reduce busy "
                          f"{tr['allreduce_span_ms']:.3f} ms, of which {tr['exposed_ms']:.3f} ms behind the end of the backward pass "
                          f"({tr['sampled_steps']} sampled steps, {world} ranks)")
            if phase in observing:
                for why in policy.observe(L, S):
                    if rank == 0 and save_path is not None:
                        torch.save(model.state_dict(), save_path)
                        if why == "score":
                            print(f'Saved model on loss+corr at epoch {epoch} of better score and loss within {delta} of optimal loss')
        verdict = policy.end_of_epoch(epoch)
        if verdict is not None:
            if rank == 0:
                print(verdict)
            break
    return model          # last-epoch model, not the best checkpoint (vit.py:243)


def _autograd_step(model, optimizer, image, rna_data, n_global, world):
    """``loss.backward(); optimizer.step()`` of vit.py:175-180 with a torch optimizer over model.parameters().
    Returns (loss, metrics) of the local batch, or None when this rank had nothing to show."""
    optimizer.zero_grad()
    out = None
    if image is not None:
        pred = model(image)
        loss, gpred = mse_loss_grad(model, pred.detach(), rna_data, grad_scale=2.0 / max(n_global, 1))
        mets = batch_metrics(model, pred.detach(), rna_data)
        pred.backward(gpred)
        out = (loss, mets)                                     # the loss VALUE is the local batch's mean, as on one rank
    if world > 1:
        if model.flat.grad is None:
            model.flat.grad = torch.zeros_like(model.flat)
        dist.all_reduce(model.flat.grad)
    optimizer.step()
    return out


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
    if not preds:            # a rank whose shard of the loader is empty (slide-sharded evaluation): nothing to report
        g = int(getattr(getattr(model, "cfg", None), "num_outputs", 0))
        return np.zeros((0, g), np.float32), np.zeros((0, g), np.float32), np.zeros((0,), dtype=str), np.zeros((0,), dtype=str)
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
    if not preds:
        return np.zeros((0, int(getattr(getattr(model, "cfg", None), "num_outputs", 0))), np.float32), np.zeros((0,), dtype=str), np.zeros((0,), dtype=str)
    return np.concatenate(preds, axis=0), np.concatenate(wsis, axis=0), np.concatenate(projs, axis=0)
