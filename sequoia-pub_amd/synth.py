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


# H&E-like colours (RGB): eosin pinks, haematoxylin purples, a dark fold and a pale stroma tone
_HE_PALETTE = np.array([[232, 160, 196], [205, 110, 170], [120, 60, 150], [70, 30, 110], [24, 12, 40], [244, 214, 228]], dtype=np.int64)


def structured_patches_u8(slide_idx, n_patches=1000, size=224, seed=4099):
    """uint8 HWC patches that look like what a slide gives rather than uniform noise: stain blobs on a 255-white
    background, saturated (0 / 255) and near-black regions, almost flat tiles, smooth gradients with a few grey levels of
    noise, dense nucleus-like texture -- five kinds in turn, every patch unique.  Integer arithmetic only (no float
    rounding between machines)."""
    rs = np.random.RandomState(seed + slide_idx)
    yy, xx = np.mgrid[0:size, 0:size].astype(np.int64)
    out = np.empty((n_patches, size, size, 3), dtype=np.uint8)

    def blobs(img, count, rmin, rmax, palette):
        for _ in range(count):
            cx, cy = rs.randint(0, size, 2)
            R = int(rs.randint(rmin, rmax))
            col = palette[rs.randint(len(palette))] + rs.randint(-12, 13, 3)
            y0, y1, x0, x1 = max(cy - R, 0), min(cy + R + 1, size), max(cx - R, 0), min(cx + R + 1, size)
            wgt = np.clip(R * R - ((xx[y0:y1, x0:x1] - cx) ** 2 + (yy[y0:y1, x0:x1] - cy) ** 2), 0, None)     # quadratic fall-off, 0 outside the disc
            img[y0:y1, x0:x1] -= ((img[y0:y1, x0:x1] - col) * wgt[..., None]) // (R * R)
        return img

    for i in range(n_patches):
        kind = i % 5
        if kind == 0:                    # tissue on glass: stain blobs on a white background
            img = blobs(np.full((size, size, 3), 255, np.int64), int(rs.randint(2, 9)), size // 10, size // 2, _HE_PALETTE)
        elif kind == 1:                  # over-stained / folded tissue: channels driven into 0 and 255, a near-black band
            img = blobs(np.full((size, size, 3), 255, np.int64), 6, size // 6, size // 2, _HE_PALETTE[2:5])
            img = (img - 128) * 3 + 128                                                   # contrast stretch: saturates both ends
            a, b, c = rs.randint(-3, 4), rs.randint(-3, 4), rs.randint(0, size)
            band = np.abs(a * xx + b * yy - (a + b) * c) < 6 * size // 10
            img[band] = rs.randint(0, 9, 3)
        elif kind == 2:                  # almost flat tile: one colour, a few dozen speckle pixels
            img = np.empty((size, size, 3), np.int64)
            img[:] = _HE_PALETTE[rs.randint(len(_HE_PALETTE))] + rs.randint(-20, 21, 3) if i % 10 == 2 else rs.choice([255, 250, 238, 16])
            py, px = rs.randint(0, size, (2, 40))
            img[py, px] = rs.randint(0, 256, (40, 3))
        elif kind == 3:                  # smooth illumination gradient with a few grey levels of noise
            c0, c1 = rs.randint(60, 256, 3), rs.randint(60, 256, 3)
            t = (xx * int(rs.randint(0, 3)) + yy * int(rs.randint(1, 3)))
            t = t * 256 // (t.max() + 1)
            img = (c0 * (256 - t[..., None]) + c1 * t[..., None]) // 256 + rs.randint(-3, 4, (size, size, 3))
        else:                            # dense nuclei on stroma
            img = np.empty((size, size, 3), np.int64)
            img[:] = _HE_PALETTE[0] + rs.randint(-10, 11, 3)
            img = blobs(img, int(rs.randint(40, 90)), 3, 9, _HE_PALETTE[2:5])
        out[i] = np.clip(img, 0, 255).astype(np.uint8)
    return out
