"""Seeded synthetic inputs shared by bench.py, the tests and the golden-vector
script.  numpy ``RandomState`` streams only, so the same
arrays are regenerated bit-for-bit on any machine."""
import numpy as np


def patches_u8(slide_idx, n_patches=1000, size=224, seed=99):
    """uint8 HWC patches of one synthetic slide."""
    rs = np.random.RandomState(seed + slide_idx)
    return rs.randint(0, 256, (n_patches, size, size, 3), dtype=np.uint8)


def features_gmm(seed, n=1000, dim=1024, comps=30):
    """Non-negative Gaussian-mixture patch features (post-ReLU/avg-pool like)."""
    rs = np.random.RandomState(seed)
    mu = np.abs(rs.randn(comps, dim)).astype(np.float32)
    z = rs.randint(0, comps, n)
    return np.abs(mu[z] + 0.5 * rs.randn(n, dim)).astype(np.float32)


def features_lowrank(seed, n=1000, dim=1024, rank=6):
    """Low-rank + noise features: a hard k-Means instance (many Lloyd iterations)."""
    rs = np.random.RandomState(seed)
    return (rs.randn(n, rank) @ rs.randn(rank, dim) + 0.05 * rs.randn(n, dim)).astype(np.float32)


def features_normal(seed, n=1000, dim=1024):
    return np.random.RandomState(seed).randn(n, dim).astype(np.float32)


def cluster_tokens(seed, n_slides, dim=1024, n_clusters=100):
    """Pre-computed cluster features [n_slides, 100, dim] f32 (BASELINE configs 2/4)."""
    rs = np.random.RandomState(seed)
    return rs.randn(n_slides, n_clusters, dim).astype(np.float32)


def rna_targets(seed, n_slides, n_genes=20820):
    """Targets U(0, 8): examples/ref_file.csv values lie in about 0..6."""
    rs = np.random.RandomState(seed)
    return (rs.rand(n_slides, n_genes) * 8.0).astype(np.float32)
