"""HIP k-Means vs scikit-learn golden labels (tests/golden/kmeans.npz) and the CPU oracle.
Bar: labels, seeding indices and iteration counts bit-equal; cluster means bit-equal (same fp32 add order)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import kmeans_oracle as ko  # noqa: E402  (checker only)
from sequoia_pub_amd import _lib, synth  # noqa: E402
from sequoia_pub_amd.kmeans import KMeans, kmeans_fit_batch, seeding_draws  # noqa: E402

CASES = [("gmm", 0, 1024), ("gmm", 1, 2048), ("lowrank", 2, 1024), ("lowrank", 3, 2048),
         ("normal", 4, 1024), ("lowrank", 5, 256), ("gmm", 6, 1024)]


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "kmeans.npz"))


def test_draw_sequence_matches_oracle():
    f, u = seeding_draws(1000, 100)
    fo, uo = ko.seeding_draws(1000, 100)
    assert f == fo and np.array_equal(u, uo)


@pytest.mark.parametrize("kind,seed,dim", CASES)
def test_labels_bit_equal_to_sklearn_golden(gold, kind, seed, dim):
    _lib.require_gpu()
    X = getattr(synth, "features_" + kind)(seed, 1000, dim)
    km = KMeans(n_clusters=100, random_state=0).fit(X)          # kmean_features.py:96
    tag = f"{kind}_{seed}_{dim}"
    assert np.array_equal(km.seed_indices_, gold[tag + "::indices"]), "k-means++ seeding order"
    assert np.array_equal(km.labels_, gold[tag + "::labels"]), int((km.labels_ != gold[tag + "::labels"]).sum())
    assert km.n_iter_ == int(gold[tag + "::n_iter"])
    assert np.array_equal(km.cluster_features_, gold[tag + "::cluster_features"])      # bitwise


def test_ragged_slide_and_batch(gold):
    _lib.require_gpu()
    X = synth.features_gmm(8, 257, 512)
    km = KMeans(n_clusters=100, random_state=0).fit(X)
    assert np.array_equal(km.labels_, gold["gmm_8_512_n257::labels"])
    # batched: 3 different slides in one call == 3 single calls == oracle
    Xs = np.stack([synth.features_lowrank(20 + i, 600, 256) for i in range(3)])
    r = kmeans_fit_batch(torch.from_numpy(Xs).cuda(), 100)
    for i in range(3):
        o = ko.kmeans_fit(Xs[i])
        assert np.array_equal(r["labels"][i].cpu().numpy(), o["labels"])
        assert np.array_equal(r["indices"][i].cpu().numpy(), o["indices"])
        assert int(r["n_iter"][i]) == o["n_iter"]
        assert np.array_equal(r["cluster_features"][i].cpu().numpy(), ko.cluster_means(Xs[i], o["labels"]))


def test_many_slides_vs_oracle():
    """statistical check at BASELINE size (1000 x 1024): 16 slides, every label equal to the oracle's."""
    _lib.require_gpu()
    Xs = np.stack([synth.features_gmm(100 + i, 1000, 1024) if i % 2 else synth.features_lowrank(100 + i, 1000, 1024)
                   for i in range(16)])
    r = kmeans_fit_batch(torch.from_numpy(Xs).cuda(), 100)
    bad = 0
    for i in range(16):
        o = ko.kmeans_fit(Xs[i])
        bad += int(not np.array_equal(r["labels"][i].cpu().numpy(), o["labels"]))
    assert bad == 0, bad


def test_duplicate_points_and_empty_clusters():
    """More clusters than distinct points (40 distinct rows, 180 samples, k=100): many centres are exact
    duplicates, so the argmin has exact ties.  The kernel breaks them towards the lower index (sklearn's
    rule, _k_means_lloyd.pyx:205-213); a BLAS-based CPU run breaks them by sgemm/dgemm rounding, so the
    comparison is on what is well defined: every point sitting on a centre
    identical to itself, empty clusters giving NaN rows (np.mean of an empty selection)."""
    _lib.require_gpu()
    rs = np.random.RandomState(0)
    base = rs.randn(40, 64).astype(np.float32)
    X = np.concatenate([base] * 5)[:180]
    km = KMeans(n_clusters=100, random_state=0).fit(X)
    seeds = km.seed_indices_
    # once every distinct row is a centre all remaining distances are exactly 0 on the GPU (duplicates
    # share one Gram row), where BLAS rounding leaves ~1e-15 residues on the CPU: only the well-defined
    # part is compared -- every distinct row gets seeded, the run stops after the first repeat of labels
    assert len({X[i].tobytes() for i in seeds}) == 40 and km.n_iter_ == 2
    for j in range(len(X)):
        assert np.array_equal(X[seeds[km.labels_[j]]], X[j])          # assigned to a centre equal to the point
        lab = km.labels_[j]
        dup = [c for c in range(100) if np.array_equal(X[seeds[c]], X[j])]
        assert lab == min(dup)                                         # exact tie -> first index
    used = np.unique(km.labels_)
    assert len(used) == 40
    empty = np.setdiff1d(np.arange(100), used)
    assert np.isnan(km.cluster_features_[empty]).all() and not np.isnan(km.cluster_features_[used]).any()


def test_too_few_samples_raises():
    with pytest.raises(ValueError):
        KMeans(n_clusters=100).fit(np.zeros((50, 16), np.float32))


def test_maximum_patch_count_and_minimum():
    """The reference caps slides at max_patch_number = 4000 patches (compute_features_hdf5.py:33): the largest slide
    the kernels accept (n = 4096) and the smallest legal one (n = n_clusters) against the oracle; n = 4097 is refused."""
    _lib.require_gpu()
    X = synth.features_gmm(77, 4096, 2048)
    r = kmeans_fit_batch(torch.from_numpy(X).cuda()[None], 100)
    o = ko.kmeans_fit(X)
    assert np.array_equal(r["labels"][0].cpu().numpy(), o["labels"]) and int(r["n_iter"][0]) == o["n_iter"]
    assert np.array_equal(r["cluster_features"][0].cpu().numpy(), ko.cluster_means(X, o["labels"]))
    Xs = synth.features_gmm(78, 100, 64)
    rs = kmeans_fit_batch(torch.from_numpy(Xs).cuda()[None], 100)
    assert sorted(rs["labels"][0].cpu().tolist()) == list(range(100))          # every point its own cluster
    with pytest.raises(_lib.SequoiaHipError):
        kmeans_fit_batch(torch.zeros(1, 4097, 64).cuda(), 100)


def test_slides_where_oracle_and_sklearn_part_follow_the_oracle(golden_dir):
    """The four slides of tests/golden/kmeans_sklearn_mismatch.json (a one-ulp tie of two candidates' potentials in one
    k-means++ step; scikit-learn's fp32 BLAS and the fl32(fp64 sum) definition pick different candidates): the HIP path
    implements the oracle's definition, so it must take the ORACLE's centre at that step -- and scikit-learn's before it."""
    import json
    _lib.require_gpu()
    fx = json.load(open(os.path.join(golden_dir, "kmeans_sklearn_mismatch.json")))
    for m in fx["mismatches"]:
        X = getattr(synth, "features_" + m["kind"])(m["seed"], 1000, m["dim"])
        km = KMeans(n_clusters=100, random_state=0).fit(X)
        first = m["first_differing_centre"]
        assert np.array_equal(km.seed_indices_, np.array(m["oracle_indices"])), m["tag"]
        assert np.array_equal(km.seed_indices_[:first], np.array(m["sklearn_indices"])[:first])
        assert np.array_equal(km.labels_, ko.kmeans_fit(X)["labels"])
