"""Golden vectors for sequoia-pub_amd/patchgen.py from REAL scikit-image / scipy -- the functions
/root/reference/pre_processing/patch_gen_hdf5.py:25-39,70-72,108-111 calls (rgb2hsv, threshold_otsu, is_low_contrast,
binary_dilation, binary_erosion) -- plus the reference's own get_mask_image text is NOT used: the mask is recomputed
here from those library calls exactly as :25-39 combines them.

Run with an interpreter that has scikit-image -- in the build image:
    /opt/conda/bin/python3.9 tests/golden/make_patchgen_golden.py        (scikit-image 0.18.3, scipy 1.7.1)
Writes tests/golden/patchgen.npz: seeded synthetic H&E-like images and, per image, Otsu thresholds of R, G, B and
of the HSV saturation, the tissue mask before / after 3 x dilation + 3 x erosion and the low-contrast flag."""
import os

import numpy as np
from scipy.ndimage import binary_dilation, binary_erosion
from skimage.color import rgb2hsv
from skimage.exposure import is_low_contrast
from skimage.filters import threshold_otsu

HERE = os.path.dirname(os.path.abspath(__file__))


def synthetic_he(seed, h, w, tissue_frac=0.45, flat=False):
    """White background with pink / purple blobs (tissue) and grey pen marks; flat=True: nearly uniform (low contrast)."""
    rs = np.random.RandomState(seed)
    img = np.full((h, w, 3), 240, dtype=np.float64) + rs.randn(h, w, 3) * 3
    if not flat:
        yy, xx = np.mgrid[0:h, 0:w]
        for _ in range(max(3, int(tissue_frac * 12))):
            cy, cx, r = rs.randint(0, h), rs.randint(0, w), rs.randint(min(h, w) // 8, min(h, w) // 3)
            blob = (yy - cy) ** 2 + (xx - cx) ** 2 < r * r
            colour = np.array([200, 120, 170]) + rs.randn(3) * 15
            img[blob] = colour + rs.randn(int(blob.sum()), 3) * 18
        img[h // 10:h // 10 + 3, :] = 40                              # a dark pen line: fails the RGB_min test
    return np.clip(img, 0, 255).astype(np.uint8)


if __name__ == "__main__":
    out = {}
    cases = [(1, 96, 128, False), (2, 64, 64, False), (3, 256, 256, False), (4, 256, 256, True), (5, 40, 72, False)]
    for seed, h, w, flat in cases:
        img = synthetic_he(seed, h, w, flat=flat)
        hsv = rgb2hsv(img)
        thr = [threshold_otsu(img[:, :, c]) for c in range(3)] + [threshold_otsu(hsv[:, :, 1])]
        bright = (img[:, :, 0] > thr[0]) & (img[:, :, 1] > thr[1]) & (img[:, :, 2] > thr[2])
        mask = (hsv[:, :, 1] > thr[3]) & np.logical_not(bright) & (img[:, :, 0] > 50) & (img[:, :, 1] > 50) & (img[:, :, 2] > 50)
        k = f"case{seed}"
        out[k + "::img"] = img
        out[k + "::thresholds"] = np.array(thr, dtype=np.float64)
        if h * w <= 96 * 128:
            out[k + "::saturation"] = hsv[:, :, 1]
        out[k + "::mask"] = mask
        out[k + "::mask_closed"] = binary_erosion(binary_dilation(mask, iterations=3), iterations=3)
        out[k + "::mask_dilated"] = binary_dilation(mask, iterations=3)
        out[k + "::low_contrast"] = np.array(is_low_contrast(img))
    import skimage
    import scipy
    out["versions"] = np.array([skimage.__version__, scipy.__version__])
    np.savez_compressed(os.path.join(HERE, "patchgen.npz"), **out)
    print("patchgen golden:", {k: (out[k + '::thresholds'].round(3).tolist(), bool(out[k + '::low_contrast']), float(out[k + '::mask'].mean())) for k in [f"case{c[0]}" for c in cases]})
