"""How often does the deterministic k-Means definition (oracle/kmeans_oracle.py == the HIP kernels) part from scikit-learn on
slides whose features are distributed like the four reference-made pipeline goldens (ResNet-50 pooled features: non-negative,
heavy-tailed columns, D = 2048)?

The goldens keep 16 / 63 probe rows of the reference ResNet's features each (``feat_probe``).  A scan slide is 1000 rows built from
one golden's probe rows: |a P[j] + (1 - a) P[k]| * (1 + 0.05 N(0, 1)) with seeded j, k, a -- mixtures of real feature rows with 5 %
multiplicative jitter, so column scales, sparsity and tails are the goldens'.  4 goldens x 64 seeds = 256 slides through
``sklearn.cluster.KMeans(100, random_state=0)`` (pre_processing/kmean_features.py:96-97) and through the oracle; the output
records the count and every slide on which the labels differ (with the first differing seeding centre).

    python tests/golden/make_kmeans_scan_goldenlike.py        (needs scikit-learn; run in the build container)
Output: tests/golden/kmeans_sklearn_scan_goldenlike.json (read by tests/test_oracle_kmeans.py)."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import sequoia_pub_amd  # noqa: E402,F401
from oracle import kmeans_oracle as ko  # noqa: E402

GOLDENS = ("pipeline_slide", "pipeline_slide_struct224", "pipeline_slide_struct256", "pipeline_slide_wide224")
SEEDS = range(7000, 7064)


def slide(golden, seed):
    P = np.load(os.path.join(HERE, golden + ".npz"))["feat_probe"].astype(np.float32)
    rs = np.random.RandomState(seed)
    j, k = rs.randint(0, P.shape[0], 1000), rs.randint(0, P.shape[0], 1000)
    a = rs.rand(1000, 1).astype(np.float32)
    return np.abs((a * P[j] + (1 - a) * P[k]) * (1 + 0.05 * rs.randn(1000, P.shape[1]))).astype(np.float32)


def main():
    import sklearn
    from sklearn.cluster import KMeans
    from sklearn.cluster._kmeans import _kmeans_plusplus
    from sklearn.utils.extmath import row_norms
    scanned, cases = [], []
    for g in GOLDENS:
        for seed in SEEDS:
            X = slide(g, seed)
            tag = f"{g}_{seed}"
            scanned.append(tag)
            km = KMeans(n_clusters=100, random_state=0).fit(X)
            r = ko.kmeans_fit(X)
            if np.array_equal(r["labels"], km.labels_):
                continue
            Xc = X - X.mean(axis=0)
            _, idx = _kmeans_plusplus(Xc, 100, row_norms(Xc, squared=True), np.ones(1000, np.float32), np.random.RandomState(0))
            first = int(np.argmax(idx != r["indices"])) if not np.array_equal(idx, r["indices"]) else -1
            cases.append(dict(tag=tag, golden=g, seed=seed, xsum=float(X.astype(np.float64).sum()), first_differing_centre=first,
                              labels_differing=int((r["labels"] != km.labels_).sum()), n_iter_sklearn=int(km.n_iter_), n_iter_oracle=int(r["n_iter"])))
            print(tag, "labels differ:", cases[-1]["labels_differing"], "first differing centre", first, flush=True)
    out = dict(sklearn_version=sklearn.__version__, numpy_version=np.__version__, n_scanned=len(scanned), n_mismatch=len(cases),
               goldens=list(GOLDENS), seeds=[SEEDS.start, SEEDS.stop], mismatches=cases,
               note="slides built from the goldens' probe rows (see the script's header); oracle == scikit-learn on n_scanned - n_mismatch of them")
    json.dump(out, open(os.path.join(HERE, "kmeans_sklearn_scan_goldenlike.json"), "w"), indent=1)
    print(len(scanned), "scanned,", len(cases), "mismatches")


if __name__ == "__main__":
    main()
