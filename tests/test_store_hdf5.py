"""Real HDF5 I/O of the feature / patch stores: files written by real h5py are read, and files
written here are read by real h5py.  The HDF5 backend in this image is h5lite (libhdf5 through ctypes)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from sequoia_pub_amd import h5lite, store

CONDA_PY = "/opt/conda/bin/python3.9"


def _arrays():
    # the generator script imports h5py at module level: re-create its seeded arrays without importing it
    rs = np.random.RandomState(41)
    feats = rs.randn(7, 16).astype(np.float32)
    clusters = rs.randn(100, 16).astype(np.float32)
    tiles = {f"{256 * i}_{512 * j}": rs.randint(0, 256, (8, 8, 3), dtype=np.uint8) for i in range(3) for j in range(2)}
    return feats, clusters, tiles


needs_hdf5 = pytest.mark.skipif(store.backend() == "npy", reason="no HDF5 backend (neither h5py nor libhdf5) on this machine")


@needs_hdf5
def test_reads_files_written_by_real_h5py(golden_dir):
    feats, clusters, tiles = _arrays()
    with store.File(os.path.join(golden_dir, "h5py_features.h5"), "r") as f:
        assert sorted(f.keys()) == ["cluster_features", "resnet_features"]
        assert "cluster_features" in f and "uni_features" not in f
        got = f["resnet_features"]
        assert tuple(got.shape) == (7, 16) and np.dtype(got.dtype) == np.float32
        assert np.array_equal(np.asarray(got[:]), feats)
        assert np.array_equal(np.asarray(f["cluster_features"][:]), clusters)
        with pytest.raises(KeyError):
            f["uni_features"]
    with store.File(os.path.join(golden_dir, "h5py_patches.hdf5"), "r") as f:
        assert sorted(f.keys()) == sorted(tiles)
        for name, t in tiles.items():
            a = np.asarray(f[name][:])
            assert a.dtype == np.uint8 and np.array_equal(a, t)


@needs_hdf5
def test_resume_guard_and_modes_on_a_real_hdf5_file(tmp_path):
    p = str(tmp_path / "TCGA-XX-0001.h5")
    with store.File(p, "w") as f:
        f.create_dataset("resnet_features", data=np.arange(24, dtype=np.float32).reshape(6, 4))
    assert open(p, "rb").read(8) == b"\x89HDF\r\n\x1a\n"                       # an HDF5 file, not the .npy mirror
    assert not os.path.isdir(p + ".d")
    with store.File(p, "r+") as f:                                                 # kmean_features.py:75,91-94,108
        assert "cluster_features" not in f.keys()
        f.create_dataset("cluster_features", data=np.ones((100, 4), np.float32))
        with pytest.raises(Exception):
            f.create_dataset("cluster_features", data=np.ones((100, 4), np.float32))
    with store.File(p, "r") as f:
        assert sorted(f.keys()) == ["cluster_features", "resnet_features"]
        with pytest.raises(Exception):
            f.create_dataset("x", data=np.zeros(3, np.float32))
    with pytest.raises(OSError):
        store.File(str(tmp_path / "missing.h5"), "r")


@pytest.mark.skipif(store.backend() != "h5lite" or not os.path.exists(CONDA_PY), reason="needs the libhdf5 backend and an interpreter with real h5py")
def test_files_written_here_are_read_by_real_h5py(tmp_path):
    feats, clusters, tiles = _arrays()
    p = str(tmp_path / "slide.h5")
    with store.File(p, "w") as f:
        f.create_dataset("resnet_features", data=feats)
        f.create_dataset("cluster_features", data=clusters)
        for name, t in tiles.items():
            f.create_dataset(name, data=t)
        f.create_dataset("labels", data=np.arange(10, dtype=np.int32))
    code = ("import h5py, numpy as np, sys\n"
            "f = h5py.File(sys.argv[1], 'r')\n"
            "print(sorted(f.keys()))\n"
            "for k in sorted(f.keys()):\n"
            "    a = f[k][:]\n"
            "    print(k, a.dtype.str, a.shape, float(a.astype('f8').sum()))\n")
    r = subprocess.run([CONDA_PY, "-W", "ignore", "-c", code, p], capture_output=True, text=True, timeout=120)
    if "No module named 'h5py'" in r.stderr:
        pytest.skip("that interpreter has no h5py")
    assert r.returncode == 0, r.stderr[-500:]
    lines = r.stdout.strip().splitlines()
    want = {"resnet_features": feats, "cluster_features": clusters, "labels": np.arange(10, dtype=np.int32), **tiles}
    assert eval(lines[0]) == sorted(want)
    for line in lines[1:]:
        name, dt, rest = line.split(" ", 2)
        shape_s, total = rest.rsplit(" ", 1)
        a = want[name]
        assert np.dtype(dt) == a.dtype and eval(shape_s) == a.shape
        assert abs(float(total) - float(a.astype("f8").sum())) <= 1e-6 * max(1.0, abs(float(a.astype("f8").sum())))


def test_npy_mirror_still_works_when_forced(tmp_path, monkeypatch):
    monkeypatch.setenv("SEQUOIA_STORE", "npy")
    p = str(tmp_path / "b.h5")
    with store.File(p, "w") as f:
        f.create_dataset("cluster_features", data=np.ones((100, 8), np.float32))
    assert os.path.isdir(p + ".d") and store.exists(p)
    with store.File(p, "r") as f:
        assert np.asarray(f["cluster_features"][:]).shape == (100, 8)
