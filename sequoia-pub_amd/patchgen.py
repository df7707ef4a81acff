"""WSI -> patches: host-side counterpart of /root/reference/pre_processing/patch_gen_hdf5.py --
``get_mask_image`` (:25-39), ``get_mask`` (:41-50) and ``extract_patches`` (:51-137): Otsu tissue mask on the lowest
pyramid level, 3 x dilation + 3 x erosion, seed-5 shuffled grid of level-0 tiles, per-tile tissue / contrast filter,
one uint8 ``"{x}_{y}"`` dataset per kept tile in ``<patch_path>/<slide>/<slide>.hdf5`` + ``complete.txt``.

This stage sits in front of the accelerated path and is bound by slide decoding, so it stays numpy; what is restated
here are the scikit-image functions the reference calls (scikit-image is not installed in this image):
``rgb2hsv`` (saturation channel), ``threshold_otsu`` (integer and float histograms), ``is_low_contrast`` -- pinned
against scikit-image 0.18.3 itself by tests/golden/patchgen.npz (made with the image's conda interpreter,
tests/golden/make_patchgen_golden.py).  ``binary_dilation`` / ``binary_erosion`` are scipy's, as in the reference.

A slide is anything with OpenSlide's interface subset (``level_dimensions``, ``read_region(location, level, size)``,
``properties``); ``ArraySlide`` provides it over numpy arrays so the whole flow runs without openslide."""
import os

import numpy as np
from scipy.ndimage import binary_dilation, binary_erosion

from . import store


# ---- scikit-image restatements ---------------------------------------------------------------------------------
def saturation(img_rgb_u8):
    """skimage.color.rgb2hsv(img)[..., 1] for a uint8 RGB image: delta / max on img_as_float(img) (float64; scikit-image
    scales by MULTIPLYING with 1/255), 0 where delta == 0."""
    arr = np.asarray(img_rgb_u8).astype(np.float64) * (1.0 / 255.0)
    v = arr.max(-1)
    delta = np.ptp(arr, -1)
    with np.errstate(invalid="ignore", divide="ignore"):
        s = delta / v
    s[delta == 0.0] = 0.0
    return s


def threshold_otsu(image, nbins=256):
    """skimage.filters.threshold_otsu: integer images -> one bin per integer value in [min, max]; float images -> `nbins`
    equal bins over [min, max], thresholds at bin centres; the class-separability maximum (first one) wins."""
    image = np.asarray(image)
    first = image.ravel()[0]
    if np.all(image == first):
        return first
    flat = image.ravel()
    if np.issubdtype(flat.dtype, np.integer):
        lo, hi = int(flat.min()), int(flat.max())
        counts = np.bincount(flat.astype(np.int64) - lo, minlength=hi - lo + 1)
        centers = np.arange(lo, hi + 1)
    else:
        counts, edges = np.histogram(flat, bins=nbins, range=(flat.min(), flat.max()))
        centers = (edges[:-1] + edges[1:]) / 2.0
    counts = counts.astype(float)
    weight1 = np.cumsum(counts)
    weight2 = np.cumsum(counts[::-1])[::-1]
    mean1 = np.cumsum(counts * centers) / weight1
    mean2 = (np.cumsum((counts * centers)[::-1]) / weight2[::-1])[::-1]
    variance12 = weight1[:-1] * weight2[1:] * (mean1[:-1] - mean2[1:]) ** 2
    return centers[int(np.argmax(variance12))]


def is_low_contrast(image_rgb_u8, fraction_threshold=0.05, lower_percentile=1, upper_percentile=99):
    """skimage.exposure.is_low_contrast on an RGB uint8 image: luminance 0.2125 R + 0.7154 G + 0.0721 B of img / 255,
    percentile spread relative to the float dtype range (-1, 1)."""
    arr = np.asarray(image_rgb_u8).astype(np.float64) * (1.0 / 255.0)
    gray = arr @ np.array([0.2125, 0.7154, 0.0721])
    lo, hi = np.percentile(gray, [lower_percentile, upper_percentile])
    return bool((hi - lo) / 2.0 < fraction_threshold)


# ---- patch_gen_hdf5.py -----------------------------------------------------------------------------------------
def get_mask_image(img_rgb, rgb_min=50):
    """patch_gen_hdf5.py:25-39: tissue = saturated AND not (bright in all three channels) AND every channel > rgb_min."""
    img = np.asarray(img_rgb)
    bright = np.ones(img.shape[:2], dtype=bool)
    for c in range(3):
        bright &= img[:, :, c] > threshold_otsu(img[:, :, c])
    s = saturation(img)
    return (s > threshold_otsu(s)) & ~bright & (img > rgb_min).all(-1)


def get_mask(slide, level='max', rgb_min=50):
    """patch_gen_hdf5.py:41-50: Otsu mask of the whole slide at pyramid level `level`; the array is indexed [x, y]."""
    if level == 'max':
        level = len(slide.level_dimensions) - 1
    img = np.transpose(np.asarray(slide.read_region((0, 0), level, slide.level_dimensions[level]))[:, :, :3], axes=[1, 0, 2])
    return get_mask_image(img, rgb_min), level


class ArraySlide:
    """OpenSlide's interface subset over in-memory pyramid levels (each [height, width, 3] uint8, level 0 first)."""

    def __init__(self, levels, properties=None):
        self.levels = [np.asarray(a) for a in levels]
        self.level_dimensions = [(a.shape[1], a.shape[0]) for a in self.levels]       # (width, height) like OpenSlide
        self.dimensions = self.level_dimensions[0]
        self.properties = dict(properties or {})

    def read_region(self, location, level, size):
        """location = level-0 (x, y) of the top-left corner, size = (width, height) at `level`; outside the slide: white."""
        a = self.levels[level]
        sx = self.level_dimensions[0][0] / self.level_dimensions[level][0]
        sy = self.level_dimensions[0][1] / self.level_dimensions[level][1]
        x0, y0 = int(location[0] / sx), int(location[1] / sy)
        out = np.full((size[1], size[0], 3), 255, dtype=np.uint8)
        h, w = max(0, min(size[1], a.shape[0] - y0)), max(0, min(size[0], a.shape[1] - x0))
        out[:h, :w] = a[y0:y0 + h, x0:x0 + w, :3]
        return out


def _resize_like_reference(patch, size):
    """patch_gen_hdf5.py:117 ``patch.resize(patch_size)`` on the PIL image of a 40x region: Pillow's default filter
    (BICUBIC since Pillow 2.7; requirements.txt pins pillow==10.3.0).  Pillow is the reference's own resampler, so the
    stored pixels are the reference's; without it there is no faithful substitute and the call fails loudly."""
    try:
        from PIL import Image
    except ImportError as e:                         # pragma: no cover
        raise RuntimeError("40x slides are shrunk with PIL.Image.resize (bicubic) as the reference does: Pillow is required") from e
    return np.asarray(Image.fromarray(np.ascontiguousarray(patch)).resize((int(size[0]), int(size[1]))))


def extract_patches(slide, mask_path, patch_size, patches_output_dir, slide_id, max_patches_per_slide=2000,
                    background_threshold=0.2):
    """patch_gen_hdf5.py:51-137 for an already opened slide.  Returns the number of patches written (None when the slide
    had been completed before)."""
    patch_folder = os.path.join(patches_output_dir, slide_id)
    os.makedirs(patch_folder, exist_ok=True)
    mask_folder = os.path.join(mask_path, slide_id)
    os.makedirs(mask_folder, exist_ok=True)
    if os.path.exists(os.path.join(patch_folder, "complete.txt")):
        print(f'{slide_id}: patches have already been extreacted')
        return None
    hdf = store.File(os.path.join(patch_folder, f"{slide_id}.hdf5"), 'w')
    n_written = 0
    try:          # patch_gen_hdf5.py:78,135-137: one unreadable slide prints its error and must not end the whole run
        mask, mask_level = get_mask(slide)
        mask = binary_erosion(binary_dilation(mask, iterations=3), iterations=3)
        np.save(os.path.join(mask_folder, "mask.npy"), mask)
        ratio_x = slide.level_dimensions[0][0] / slide.level_dimensions[mask_level][0]
        ratio_y = slide.level_dimensions[0][1] / slide.level_dimensions[mask_level][1]
        xmax, ymax = slide.level_dimensions[0]
        resize_factor = float(slide.properties.get('aperio.AppMag', 20)) / 20.0          # 40x slides: read 2x the size, shrink
        size_read = (int(resize_factor * patch_size[0]), int(resize_factor * patch_size[1]))
        print(f"patch size for {slide_id}: {size_read}")
        indices = [(x, y) for x in range(0, xmax, size_read[0]) for y in range(0, ymax, size_read[0])]
        if max_patches_per_slide is None:
            max_patches_per_slide = len(indices)
        np.random.seed(5)
        np.random.shuffle(indices)
        for x, y in indices:
            if n_written >= max_patches_per_slide:
                break
            if mask[int(x / ratio_x), int(y / ratio_y)] != 1:
                continue
            patch = np.asarray(slide.read_region((x, y), 0, size_read))[:, :, :3]
            tissue = binary_dilation(get_mask_image(patch), iterations=3)
            if tissue.sum() > background_threshold * tissue.size and not is_low_contrast(patch):
                if resize_factor != 1.0:
                    patch = _resize_like_reference(patch, patch_size)
                hdf.create_dataset(f"{x}_{y}", data=np.ascontiguousarray(patch))
                n_written += 1
    except Exception as e:
        print("error with slide id {} patch {}".format(slide_id, n_written))
        print(e)
        return None
    finally:
        hdf.close()
    if n_written == 0:
        print("no patch extracted for slide {}".format(slide_id))
    else:
        with open(os.path.join(patch_folder, "complete.txt"), 'w') as f:
            f.write('Process complete!\n')
            f.write(f"Total n patch = {n_written}")
        print(f"{slide_id} complete, total n patch = {n_written}")
    return n_written
