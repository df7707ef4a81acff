"""patch_gen_hdf5.py counterpart: the scikit-image functions restated in sequoia-pub_amd/patchgen.py
against golden vectors made with real scikit-image 0.18.3 / scipy (tests/golden/make_patchgen_golden.py), and the
tiling flow of extract_patches on an in-memory slide."""
import os

import numpy as np
from scipy.ndimage import binary_dilation, binary_erosion

from sequoia_pub_amd import patchgen, store

CASES = ["case1", "case2", "case3", "case4", "case5"]


def test_mask_pieces_equal_scikit_image(golden_dir):
    z = np.load(os.path.join(golden_dir, "patchgen.npz"))
    for k in CASES:
        img = z[k + "::img"]
        thr = [patchgen.threshold_otsu(img[:, :, c]) for c in range(3)]
        s = patchgen.saturation(img)
        thr.append(patchgen.threshold_otsu(s))
        assert np.array_equal(np.array(thr, dtype=np.float64), z[k + "::thresholds"]), (k, thr, z[k + "::thresholds"])
        if k + "::saturation" in z.files:
            assert np.array_equal(s, z[k + "::saturation"])
        mask = patchgen.get_mask_image(img)
        assert np.array_equal(mask, z[k + "::mask"])
        assert np.array_equal(binary_dilation(mask, iterations=3), z[k + "::mask_dilated"])
        assert np.array_equal(binary_erosion(binary_dilation(mask, iterations=3), iterations=3), z[k + "::mask_closed"])
        assert patchgen.is_low_contrast(img) == bool(z[k + "::low_contrast"])
    assert bool(z["case4::low_contrast"]) and not bool(z["case1::low_contrast"])          # both branches are pinned


def test_constant_image_threshold():
    assert patchgen.threshold_otsu(np.full((5, 5), 7, dtype=np.uint8)) == 7               # skimage returns the single value


def _slide(seed=0, tiles=(16, 12), ps=32):
    """Level 0: tiles[0] x tiles[1] tiles of ps pixels, left half tissue-like, right half blank; level 1: 8x smaller."""
    rs = np.random.RandomState(seed)
    W, H = tiles[0] * ps, tiles[1] * ps
    img = np.full((H, W, 3), 242, dtype=np.float64) + rs.randn(H, W, 3) * 2
    tissue = np.zeros((H, W), dtype=bool)
    tissue[:, : W // 2] = True
    col = np.array([190, 110, 160])
    img[tissue] = col + rs.randn(int(tissue.sum()), 3) * 25
    img = np.clip(img, 0, 255).astype(np.uint8)
    small = img[::8, ::8].copy()
    return patchgen.ArraySlide([img, small]), tissue


def test_extract_patches_flow(tmp_path):
    slide, tissue = _slide()
    ps = 32
    n = patchgen.extract_patches(slide, str(tmp_path / "masks"), (ps, ps), str(tmp_path / "patches"), "S1", max_patches_per_slide=None)
    mask = np.load(tmp_path / "masks" / "S1" / "mask.npy")
    assert mask.shape == (slide.level_dimensions[1][0], slide.level_dimensions[1][1])     # indexed [x, y] (patch_gen_hdf5.py:46-47)
    half = mask.shape[0] // 2                     # (erosion with a zero border eats 3 pixels at the slide's edge)
    assert mask[4:half - 2, 4:-4].mean() > 0.95 and mask[half + 2:].mean() < 0.05
    with store.File(str(tmp_path / "patches" / "S1" / "S1.hdf5"), "r") as f:
        keys = sorted(f.keys())
        assert len(keys) == n and n > 0
        for k in keys:
            x, y = map(int, k.split("_"))
            assert x % ps == 0 and y % ps == 0 and x < slide.level_dimensions[0][0] // 2          # tissue half only
            t = np.asarray(f[k][:])
            assert t.shape == (ps, ps, 3) and t.dtype == np.uint8
            assert np.array_equal(t, slide.levels[0][y:y + ps, x:x + ps])
    assert (tmp_path / "patches" / "S1" / "complete.txt").read_text().endswith(f"Total n patch = {n}")
    # resume guard: a completed slide is not touched again (patch_gen_hdf5.py:61-63)
    assert patchgen.extract_patches(slide, str(tmp_path / "masks"), (ps, ps), str(tmp_path / "patches"), "S1") is None
    # the cap and the seed-5 visiting order: the first k kept tiles of the full run, in shuffled-grid order
    n2 = patchgen.extract_patches(slide, str(tmp_path / "masks2"), (ps, ps), str(tmp_path / "patches2"), "S1", max_patches_per_slide=3)
    assert n2 == 3
    with store.File(str(tmp_path / "patches2" / "S1" / "S1.hdf5"), "r") as f:
        assert set(f.keys()) <= set(keys) and len(f.keys()) == 3


def test_40x_slides_are_read_at_double_size_and_shrunk(tmp_path):
    slide, _ = _slide(seed=2, tiles=(16, 10), ps=32)
    slide.properties['aperio.AppMag'] = '40'
    n = patchgen.extract_patches(slide, str(tmp_path / "m"), (16, 16), str(tmp_path / "p"), "S2", max_patches_per_slide=None)
    with store.File(str(tmp_path / "p" / "S2" / "S2.hdf5"), "r") as f:
        assert n == len(f.keys()) > 0
        for k in f.keys():
            x, y = map(int, k.split("_"))
            t = np.asarray(f[k][:])
            assert x % 32 == 0 and y % 32 == 0 and t.shape == (16, 16, 3)
            # patch_gen_hdf5.py:117: the stored pixels are PIL's default (bicubic) resize of the 32 x 32 region
            from PIL import Image
            want = np.asarray(Image.fromarray(slide.levels[0][y:y + 32, x:x + 32]).resize((16, 16)))
            assert np.array_equal(t, want)
            assert not np.array_equal(t, slide.levels[0][y:y + 32:2, x:x + 32:2])          # (not a nearest-neighbour pick)


def test_an_unreadable_slide_is_reported_not_raised(tmp_path, capsys):
    """patch_gen_hdf5.py:78,135-137: the slide's body runs under try / except (one bad slide must not kill the pool's
    map), and the patch store is closed on the way out."""
    class Broken(patchgen.ArraySlide):
        def read_region(self, location, level, size):
            if level == 0:
                raise OSError("tile decode failed")
            return super().read_region(location, level, size)
    good, _ = _slide(seed=3)
    slide = Broken(good.levels)
    assert patchgen.extract_patches(slide, str(tmp_path / "m"), (32, 32), str(tmp_path / "p"), "S3") is None
    assert "error with slide id S3" in capsys.readouterr().out
    assert not (tmp_path / "p" / "S3" / "complete.txt").exists()
    with store.File(str(tmp_path / "p" / "S3" / "S3.hdf5"), "r") as f:                     # closed properly: readable, empty
        assert len(f.keys()) == 0
