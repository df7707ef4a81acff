"""Oracle (test infrastructure): per-slide k-Means(100) + cluster means, CPU numpy.

Call site restated: /root/reference/pre_processing/kmean_features.py:96-108

    kmeans = KMeans(n_clusters=args.num_clusters, random_state=0).fit(features)
    clusters = kmeans.labels_
    mean_features[pos] = np.mean(features[np.where(clusters == pos)], axis=0)

The arithmetic lives in a third-party dependency that is NOT vendored in the
reference: scikit-learn, pinned ``scikit-learn==1.4.2`` (requirements.txt:69).
This file restates its published algorithm (``sklearn/cluster/_kmeans.py``:
``KMeans.fit`` / ``_kmeans_plusplus`` / ``_kmeans_single_lloyd``;
``_k_means_lloyd.pyx``: ``lloyd_iter_chunked_dense``; ``_k_means_common.pyx``:
``_relocate_empty_clusters_dense`` / ``_average_centers`` / ``_center_shift``;
``metrics/pairwise.py``: ``_euclidean_distances_upcast``) for the defaults the
call site uses: init='k-means++', n_init='auto' -> 1, algorithm='lloyd',
max_iter=300, tol=1e-4, dense float32 input, unit sample weights.

Pinning: the reference holds no test or golden vector for this step, so the
restatement is pinned against scikit-learn 1.7.2 (the version in the build
image; same algorithm and defaults as 1.4.x) run in the build container:
``tests/golden/kmeans_*.npz`` hold sklearn's ``labels_``, seeding indices and
``n_iter_`` for seeded inputs, and ``tests/test_oracle_kmeans.py`` checks this
file reproduces them bit-for-bit.

Where scikit-learn's own result depends on the BLAS it is linked against (fp32
``sdot``/``sgemm`` reduction order -- not reproducible across machines), this
restatement fixes a deterministic definition that lies inside that variability:

  * seeding distances  d = fl32(max(0, (-2 x.c + |x|^2) + |c|^2))   with the dot
    product and norms in fp64 (exactly sklearn's upcast path, pairwise.py:582-653)
  * potentials         pot = fl32( sum_fp64(d) )        (sklearn: fp32 BLAS dot)
  * Lloyd distances    |c|^2 - 2 x.c in fp64 on the fp32 centres
                                                        (sklearn: fp32 sgemm)
  * centre update      fl32( sum_fp64(x) / count )      (sklearn: fp32 running sum)

Labels differ from sklearn's only when one of those roundings crosses a
comparison (measured rate in DESIGN.md).  The HIP kernels implement exactly the
definitions above, so HIP == oracle is expected bit-for-bit on labels.
"""
import numpy as np


def seeding_draws(n_samples, n_clusters, random_state=0):
    """The data-independent MT19937 draw sequence of ``_kmeans_plusplus``
    (_kmeans.py:225,243): first centre ``choice(n, p=uniform)``, then for every
    further centre ``uniform(size=2+int(log(k)))``.  Returns (first_id, u[k-1, T])."""
    rs = np.random.RandomState(random_state)
    sw = np.ones(n_samples, dtype=np.float32)
    first = int(rs.choice(n_samples, p=sw / sw.sum()))
    n_local_trials = 2 + int(np.log(n_clusters))
    u = np.empty((n_clusters - 1, n_local_trials), dtype=np.float64)
    for c in range(n_clusters - 1):
        u[c] = rs.uniform(size=n_local_trials)
    return first, u


def center_data(X):
    """_kmeans.py:1479-1481: ``X_mean = X.mean(axis=0); X -= X_mean`` in fp32.
    numpy reduces axis 0 of a C-contiguous fp32 matrix by adding rows in order
    into an fp32 accumulator, then divides by n in fp32; restated explicitly so
    the HIP kernel can follow the same order."""
    X = np.ascontiguousarray(X, dtype=np.float32)
    acc = np.zeros(X.shape[1], dtype=np.float32)
    for i in range(X.shape[0]):
        acc += X[i]
    mean = acc / np.float32(X.shape[0])
    return X - mean, mean


def _sq_dists_upcast(Xc64, norms64, cand_ids):
    """pairwise.py:582-653 with X=candidates (fp32->fp64), Y=data (fp32->fp64):
    d = -2 * X.Y^T; d += XX; d += YY; cast fp32; then maximum(d, 0) (:428-431)."""
    d = -2.0 * (Xc64[cand_ids] @ Xc64.T)
    d += norms64[cand_ids][:, None]
    d += norms64[None, :]
    d = d.astype(np.float32)
    np.maximum(d, 0, out=d)
    return d


def _pot(d32):
    return np.float32(np.sum(d32.astype(np.float64)))


def kmeans_plusplus(Xc, n_clusters, first_id, u):
    """_kmeans.py:174-272 on centred fp32 data; returns indices[k] into Xc."""
    n = Xc.shape[0]
    X64 = Xc.astype(np.float64)
    norms64 = np.einsum("ij,ij->i", X64, X64)
    indices = np.full(n_clusters, -1, dtype=np.int64)
    indices[0] = first_id
    closest = _sq_dists_upcast(X64, norms64, np.array([first_id]))[0]
    pot = _pot(closest)
    for c in range(1, n_clusters):
        rand_vals = u[c - 1] * np.float64(pot)
        cum = np.cumsum(closest, dtype=np.float64)          # stable_cumsum
        cand = np.searchsorted(cum, rand_vals)               # side='left'
        np.clip(cand, None, n - 1, out=cand)
        dc = _sq_dists_upcast(X64, norms64, cand)
        np.minimum(closest, dc, out=dc)
        pots = np.array([_pot(row) for row in dc], dtype=np.float32)
        best = int(np.argmin(pots))
        pot = pots[best]
        closest = dc[best]
        indices[c] = cand[best]
    return indices


def _assign(X64, centers32):
    """_k_means_lloyd.pyx:196-214: argmin_j (|c_j|^2 - 2 x.c_j), strict '<' so
    the first minimum wins."""
    C64 = centers32.astype(np.float64)
    cn = np.einsum("ij,ij->i", C64, C64)
    pd = cn[None, :] - 2.0 * (X64 @ C64.T)
    return np.argmin(pd, axis=1).astype(np.int32)


def lloyd(Xc, centers_init, tol, max_iter=300):
    """_kmeans.py:624-760 (_kmeans_single_lloyd) + lloyd_iter_chunked_dense."""
    n, k = Xc.shape[0], centers_init.shape[0]
    X64 = Xc.astype(np.float64)
    centers = centers_init.astype(np.float32).copy()
    labels_old = np.full(n, -1, dtype=np.int32)
    strict = False
    n_iter = 0
    for it in range(max_iter):
        n_iter = it + 1
        labels = _assign(X64, centers)
        sums = np.zeros((k, Xc.shape[1]), dtype=np.float64)
        np.add.at(sums, labels, X64)
        weight = np.bincount(labels, minlength=k).astype(np.float64)
        # _relocate_empty_clusters_dense (_k_means_common.pyx:167-211)
        empty = np.where(weight == 0)[0]
        if len(empty) > 0:
            dist = ((Xc - centers[labels]) ** 2).sum(axis=1, dtype=np.float64)
            if dist.max() > 0:
                far = np.lexsort((np.arange(n), -dist))[:len(empty)]   # descending, ties by index
                for new_id, far_idx in zip(empty, far):
                    old_id = labels[far_idx]
                    sums[old_id] -= X64[far_idx]
                    sums[new_id] = X64[far_idx]
                    weight[new_id] = 1
                    weight[old_id] -= 1
        # _average_centers (:274-295)
        new = np.empty_like(centers)
        amax = int(np.argmax(weight))
        for j in range(k):
            if weight[j] > 0:
                new[j] = (sums[j] / weight[j]).astype(np.float32)
        for j in range(k):
            if not weight[j] > 0:
                new[j] = new[amax]
        shift_tot = float(((new.astype(np.float64) - centers.astype(np.float64)) ** 2).sum())
        centers = new
        if np.array_equal(labels, labels_old):
            strict = True
            break
        if shift_tot <= tol:
            break
        labels_old = labels
    if not strict:
        labels = _assign(X64, centers)
    return labels, centers, n_iter


def tolerance(X, tol=1e-4):
    """_kmeans.py:279-287: mean(var(X, axis=0)) * tol (fp64 here)."""
    return float(np.mean(np.var(X.astype(np.float64), axis=0)) * tol)


def kmeans_fit(X, n_clusters=100, random_state=0, max_iter=300, tol=1e-4):
    """KMeans(n_clusters, random_state=0).fit(X) -> dict(labels, indices, n_iter, centers)."""
    X = np.ascontiguousarray(X, dtype=np.float32)
    tol_ = tolerance(X, tol)
    first, u = seeding_draws(X.shape[0], n_clusters, random_state)
    Xc, mean = center_data(X)
    idx = kmeans_plusplus(Xc, n_clusters, first, u)
    labels, centers, n_iter = lloyd(Xc, Xc[idx], tol_, max_iter)
    return dict(labels=labels, indices=idx, n_iter=n_iter, centers=centers + mean)


def cluster_means(features, labels, n_clusters=100):
    """kmean_features.py:99-105: row j = np.mean(features[labels == j], axis=0)
    on the ORIGINAL (uncentred) fp32 features; numpy adds member rows in index
    order into an fp32 accumulator and divides by the count in fp32.  An empty
    cluster gives a NaN row, as numpy does."""
    features = np.ascontiguousarray(features, dtype=np.float32)
    out = np.empty((n_clusters, features.shape[1]), dtype=np.float32)
    for j in range(n_clusters):
        rows = features[np.where(labels == j)]
        acc = np.zeros(features.shape[1], dtype=np.float32)
        for r in rows:
            acc += r
        with np.errstate(invalid="ignore", divide="ignore"):
            out[j] = acc / np.float32(len(rows))
    return out
