"""Host-side window bookkeeping of the sliding-window path (spatial.py), CPU only: the device-side enumeration and the sync-free
vote lists against the numpy restatement of visualize.py:35-102's loop order."""
import numpy as np
import pytest
import torch

import sequoia_pub_amd  # noqa: F401
from sequoia_pub_amd.spatial import enumerate_windows, enumerate_windows_device, tile_window_lists


@pytest.mark.parametrize("stride", [1, 3, 10])
@pytest.mark.parametrize("holes,shuffled", [(False, False), (True, False), (True, True)])
def test_device_enumeration_and_bounded_vote_lists_equal_the_numpy_form(stride, holes, shuffled):
    rng = np.random.default_rng(7 + stride)
    xs, ys = np.meshgrid(np.arange(37), np.arange(29), indexing="ij")
    x, y = xs.ravel(), ys.ravel()
    if holes:
        keep = rng.random(x.size) < 0.8
        x, y = x[keep], y[keep]
    if shuffled:                                   # df order need not follow the grid: members are sorted by df position
        perm = rng.permutation(x.size)
        x, y = x[perm], y[perm]
    m, _ = enumerate_windows(x, y, stride)
    md = enumerate_windows_device(x, y, stride, "cpu")
    assert m.shape == tuple(md.shape) and np.array_equal(m, md.numpy())
    l0, c0 = tile_window_lists(m, x.size, "cpu")
    l1, c1 = tile_window_lists(md, x.size, "cpu", max_votes=(-(-10 // stride)) ** 2)
    V = l0.shape[1]
    assert torch.equal(c0, c1) and torch.equal(l0, l1[:, :V]) and bool((l1[:, V:] == -1).all())


def test_device_enumeration_rejects_duplicates_and_handles_empty_grids():
    with pytest.raises(ValueError):
        enumerate_windows_device(np.array([0, 1, 1]), np.array([0, 2, 2]), 1, "cpu")
    assert enumerate_windows_device(np.array([0]), np.array([0]), 1, "cpu").shape == (0, 100)
    assert enumerate_windows_device(np.arange(5), np.arange(5), 1, "cpu").shape == (0, 100)      # windows exist, none holds > 50 tiles
