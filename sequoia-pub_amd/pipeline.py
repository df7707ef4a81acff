"""End-to-end slide pipeline on one GPU: uint8 patches -> ResNet-50 features -> per-slide k-Means(100)
-> cluster means -> ViS -> gene-expression vector.

This is the composition the reference performs through files:
  pre_processing/compute_features_hdf5.py:116-136  ->  "<feat_type>_features" [n, 2048]
  pre_processing/kmean_features.py:96-108          ->  "cluster_features"     [100, 2048]
  evaluation/predict_independent_dataset.py:54-91  ->  predictions [n_slides, G]
kept on the device between stages (BASELINE config 3)."""
import torch

from . import _lib
from .kmeans import kmeans_fit_batch


class SlidePipeline:
    def __init__(self, resnet, vis, n_clusters=100, sub_batch=500):
        self.resnet = resnet
        self.vis = vis
        self.n_clusters = n_clusters
        self.sub_batch = sub_batch
        self._side = None

    @torch.no_grad()
    def embed(self, patches_u8):
        """[n, S, S, 3] uint8 (device or host) -> f32 [n, 2048] on the device."""
        return self.resnet.extract_patches_u8(patches_u8, sub_batch=self.sub_batch)

    @torch.no_grad()
    def cluster(self, features):
        """[S, n, D] f32 -> (cluster_features [S, 100, D], labels [S, n])."""
        r = kmeans_fit_batch(features, self.n_clusters, random_state=0)
        return r["cluster_features"], r["labels"]

    # ---- one slide: embed on the main stream, cluster on the side stream ------------------------------------------
    def _side_stream(self, dev):
        if self._side is None or self._side.device != dev:
            self._side = torch.cuda.Stream(device=dev)
        return self._side

    def _embed_deferred(self, p, main):
        """Enqueue the embedding of one slide on the main stream without any host sync.  Split-fp16 embedder: the launch
        groups OR their non-finite flag (resnet.extract_patches_u8 'defer') into a per-slide device word whose copy to
        pinned host memory is enqueued behind them; _cluster_on_side reads it where the host waits for k-Means anyway."""
        if isinstance(p, tuple):                   # (tensor, event): an upload still in flight on a copy stream
            p, uploaded = p
            main.wait_event(uploaded)
        host_flag = None
        if getattr(self.resnet, "compute_dtype", None) == _lib.SQ_F16X3:
            flag = self.resnet.new_flag()
            f = self.resnet.extract_patches_u8(p, sub_batch=self.sub_batch, on_nonfinite="defer", flag=flag)
            host_flag = torch.empty(1, dtype=torch.int32, pin_memory=True)
            host_flag.copy_(flag, non_blocking=True)
        else:
            f = self.embed(p)
        ev = torch.cuda.Event()
        ev.record(main)
        return f, ev, p, host_flag

    def _cluster_on_side(self, item, main, side):
        """k-Means + cluster means of one embedded slide on the side stream (the host polls convergence there while the
        main stream already holds the next slide's embedding).  Returns (features, cluster_features, labels, event)."""
        f, ev, p, host_flag = item
        side.wait_event(ev)
        f.record_stream(side)
        with torch.cuda.stream(side):
            if host_flag is not None:
                # read the flag BEFORE clustering: k-Means never sees non-finite features.  The host would block on this
                # slide's embedding inside k-Means' first convergence poll anyway, and the main stream already holds the
                # next slide's embedding, so the chip is not idle while it waits here
                ev.synchronize()
                if int(host_flag[0]) != 0:         # an fp16 plane overflowed in this slide: embed it again in exact fp32
                    import warnings
                    warnings.warn("split-fp16 embedder: an activation left fp16's range in one slide; the slide is re-embedded in exact fp32",
                                  RuntimeWarning, stacklevel=2)
                    self.nonfinite_reruns = getattr(self, "nonfinite_reruns", 0) + 1
                    f = self.resnet.exact_twin().extract_patches_u8(p.to(f.device), 128)
                    f.record_stream(main)
            cf, lab = self.cluster(f.unsqueeze(0))
            fin = torch.cuda.Event()
            fin.record(side)
        cf.record_stream(main)
        lab.record_stream(main)
        return f, cf, lab[0], fin

    @torch.no_grad()
    def __call__(self, slides_u8):
        """slides_u8: list of [n_i, S, S, 3] uint8 tensors (or (tensor, torch.cuda.Event) pairs whose upload is
        still in flight on another stream), or one [S, n, S, S, 3] tensor.
        Returns dict(pred [S, G], cluster_features [S, 100, D], labels list).

        The k-Means of slide i (small grids, a host check of the convergence flags every few Lloyd
        iterations) runs on a side stream while the ResNet of slide i+1 -- already enqueued on the main
        stream -- keeps the chip busy, so only the last slide's clustering is exposed."""
        dev = self.vis.flat.device
        main = torch.cuda.current_stream(dev)
        side = self._side_stream(dev)
        if len(slides_u8) == 0:
            z = torch.empty(0, self.vis.cfg.num_outputs, device=dev)
            return dict(pred=z, cluster_features=torch.empty(0, self.n_clusters, self.vis._dim(), device=dev), labels=[], features=[])
        done = []
        pending = None
        for p in slides_u8:
            item = self._embed_deferred(p, main)   # enqueued first: the GPU has this to chew on ...
            if pending is not None:
                done.append(self._cluster_on_side(pending, main, side))      # ... while the host drives the previous slide's k-Means
            pending = item
        done.append(self._cluster_on_side(pending, main, side))
        main.wait_stream(side)
        cf = torch.cat([d[1] for d in done])
        pred = self.vis(cf)
        return dict(pred=pred, cluster_features=cf, labels=[d[2] for d in done], features=[d[0] for d in done])

    # ---- streaming form: throughput over latency -------------------------------------------------------------
    @torch.no_grad()
    def submit(self, slides_u8):
        """Streaming form of __call__ for a long list of slides fed in groups: the clustering and the aggregator
        forward of the LAST slide of a group are not waited for -- they run beside the first ResNet of the next
        group -- so no group has an exposed tail.  Returns the results that became complete (a dict like __call__'s,
        for the slides finished so far, possibly of the previous group), or None.  Call flush() after the last group."""
        dev = self.vis.flat.device
        main = torch.cuda.current_stream(dev)
        side = self._side_stream(dev)
        st = self.__dict__.setdefault("_stream_state", dict(pending=None, done=[]))
        for p in slides_u8:
            item = self._embed_deferred(p, main)
            if st["pending"] is not None:
                st["done"].append(self._cluster_on_side(st["pending"], main, side))
            st["pending"] = item
        return self._collect(main)

    def _collect(self, main):
        st = self._stream_state
        if not st["done"]:
            return None
        done, st["done"] = st["done"], []
        main.wait_event(done[-1][3])               # the side stream is in order: the last event covers all of them
        cf = torch.cat([d[1] for d in done])
        pred = self.vis(cf)
        return dict(pred=pred, cluster_features=cf, labels=[d[2] for d in done], features=[d[0] for d in done])

    @torch.no_grad()
    def flush(self):
        """Finish the slide still in flight after the last submit(); returns its results (or None)."""
        st = self.__dict__.get("_stream_state")
        if not st:
            return None
        dev = self.vis.flat.device
        main = torch.cuda.current_stream(dev)
        if st["pending"] is not None:
            item, st["pending"] = st["pending"], None
            st["done"].append(self._cluster_on_side(item, main, self._side_stream(dev)))
        return self._collect(main)
