"""KMeans -- host-side mirror of the interface /root/reference/pre_processing/kmean_features.py:96-97
uses (``sklearn.cluster.KMeans(n_clusters, random_state=0).fit(X).labels_``) plus the cluster-mean
step of :99-108, on the HIP k-means kernels (``sq_kmeans_fit``).

The only host arithmetic is the data-independent MT19937 draw sequence of scikit-learn's
``_kmeans_plusplus`` (``RandomState(random_state).choice(n, p=uniform)`` then ``uniform(size=2+log k)``
per further centre), produced here with numpy's ``RandomState`` and handed to the kernel."""
import numpy as np
import torch

from . import _lib


def seeding_draws(n_samples, n_clusters, random_state=0):
    """sklearn/cluster/_kmeans.py:222-243: (first_center, uniforms[k-1, 2+int(log k)])."""
    rs = np.random.RandomState(random_state)
    sw = np.ones(n_samples, dtype=np.float32)
    first = int(rs.choice(n_samples, p=sw / sw.sum()))
    trials = 2 + int(np.log(n_clusters))
    u = np.empty((max(n_clusters - 1, 1), trials), dtype=np.float64)
    for c in range(n_clusters - 1):
        u[c] = rs.uniform(size=trials)
    return first, u


_draws_cache = {}


def _device_draws(n_samples, n_clusters, random_state, dev):
    """The draw sequence depends only on (n, k, seed): the reference clusters every slide with the same
    ``random_state`` (kmean_features.py:96), so the host RNG walk and the (synchronous, pageable) upload are paid once
    per shape instead of once per slide -- 60-80 us of a 1.6 ms call."""
    key = (int(n_samples), int(n_clusters), int(random_state), str(dev))
    hit = _draws_cache.get(key)
    if hit is None:
        first, u = seeding_draws(n_samples, n_clusters, random_state)
        if len(_draws_cache) >= 64:
            _draws_cache.clear()
        hit = _draws_cache[key] = (first, torch.from_numpy(u).to(dev), u.shape[1])      # blocking copy: visible to every stream
    return hit


def kmeans_fit_batch(X, n_clusters=100, random_state=0, max_iter=300, tol=1e-4, want_means=True):
    """X: f32 [S, n, D] CUDA tensor (S slides with the same patch count).  Returns dict of CUDA tensors:
    labels i32 [S, n], cluster_features f32 [S, k, D], indices i32 [S, k], n_iter i32 [S]."""
    _lib.require_gpu()
    if X.dim() == 2:
        X = X.unsqueeze(0)
    if not X.is_cuda:
        raise _lib.SequoiaHipError("kmeans_fit_batch needs a CUDA tensor (no CPU fallback)")
    X = X.to(torch.float32).contiguous()
    S, n, D = X.shape
    dev = X.device
    first, u_dev, trials = _device_draws(n, n_clusters, random_state, dev)
    labels = torch.empty(S, n, dtype=torch.int32, device=dev)
    means = torch.empty(S, n_clusters, D, dtype=torch.float32, device=dev) if want_means else None
    seeds = torch.empty(S, n_clusters, dtype=torch.int32, device=dev)
    n_iter = torch.empty(S, dtype=torch.int32, device=dev)
    need = _lib.lib().sq_kmeans_workspace_bytes(S, n, D, n_clusters)
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().sq_kmeans_fit(_lib.ptr(X), S, n, D, n_clusters, first, _lib.ptr(u_dev), trials, max_iter,
                                            float(tol), _lib.ptr(labels), _lib.ptr(means), _lib.ptr(seeds), _lib.ptr(n_iter),
                                            _lib.ptr(ws), need, _lib.stream_ptr(dev)))
    return dict(labels=labels, cluster_features=means, indices=seeds, n_iter=n_iter)


class KMeans:
    """The subset of sklearn.cluster.KMeans the reference touches: constructor (n_clusters,
    random_state), ``fit(X)``, ``labels_``; also ``cluster_features_`` (kmean_features.py:99-105)."""

    def __init__(self, n_clusters=8, random_state=None, max_iter=300, tol=1e-4, device="cuda:0"):
        self.n_clusters = n_clusters
        self.random_state = 0 if random_state is None else random_state
        self.max_iter = max_iter
        self.tol = tol
        self.device = device

    def fit(self, X):
        X = np.ascontiguousarray(np.asarray(X), dtype=np.float32)
        if X.shape[0] < self.n_clusters:
            raise ValueError(f"n_samples={X.shape[0]} should be >= n_clusters={self.n_clusters}.")
        r = kmeans_fit_batch(torch.from_numpy(X).to(self.device), self.n_clusters, self.random_state, self.max_iter, self.tol)
        self.labels_ = r["labels"][0].cpu().numpy()
        self.cluster_features_ = r["cluster_features"][0].cpu().numpy()
        self.seed_indices_ = r["indices"][0].cpu().numpy()
        self.n_iter_ = int(r["n_iter"][0])
        return self
