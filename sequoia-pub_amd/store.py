"""Feature store with the reference's on-disk names.

The reference keeps per-slide tensors in HDF5 files ``<feature_path>/<project>/<WSI>/<WSI>.h5`` with datasets
``resnet_features | uni_features [n, D]`` and ``cluster_features [100, D]`` (compute_features_hdf5.py:134-135,
kmean_features.py:108), and patches in ``<patch_path>/<slide>/<slide>.hdf5`` with one uint8 ``[S, S, 3]``
dataset per tile named ``"{x}_{y}"`` (patch_gen_hdf5.py:119-120).  Backends, in order: ``h5py`` when importable;
``h5lite`` -- the HDF5 C library through ctypes (this image ships libhdf5 but not h5py) -- so the files are REAL HDF5
either way and real SEQUOIA stores work unchanged; only when neither exists do the same dataset names live as
``.npy`` files inside a directory with the HDF5 file's path + ``.d`` (same keys, dtypes, shapes).
``SEQUOIA_STORE=npy`` forces that mirror (tests)."""
import os

import numpy as np

try:
    import h5py  # noqa: F401
    HAVE_H5PY = True
except Exception:
    h5py = None
    HAVE_H5PY = False

from . import h5lite


def backend():
    """'h5py', 'h5lite' (libhdf5 through ctypes) or 'npy' (directory mirror)."""
    if os.environ.get("SEQUOIA_STORE") == "npy":
        return "npy"
    if HAVE_H5PY:
        return "h5py"
    return "h5lite" if h5lite.available() else "npy"


class _NpyDirFile:
    def __init__(self, path, mode):
        self.dir = path + ".d"
        self.mode = mode
        if mode == "w":
            os.makedirs(self.dir, exist_ok=True)
            for f in os.listdir(self.dir):
                os.remove(os.path.join(self.dir, f))
        elif not os.path.isdir(self.dir):
            raise OSError(f"Unable to open file (no such store: {self.dir})")

    def keys(self):
        return sorted(f[:-4] for f in os.listdir(self.dir) if f.endswith(".npy"))

    def __contains__(self, name):
        return os.path.exists(os.path.join(self.dir, name + ".npy"))

    def __getitem__(self, name):
        p = os.path.join(self.dir, name + ".npy")
        if not os.path.exists(p):
            raise KeyError(name)
        return np.load(p, mmap_mode="r")

    def create_dataset(self, name, data):
        if self.mode == "r":
            raise OSError("store opened read-only")
        p = os.path.join(self.dir, name + ".npy")
        if os.path.exists(p):
            raise ValueError(f"Unable to create dataset (name already exists): {name}")
        np.save(p, np.asarray(data))

    def close(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def File(path, mode="r"):
    """h5py.File-compatible handle (subset: keys / [] / create_dataset / close / context manager).
    Modes: "r", "w", "r+" (append to an existing store)."""
    be = backend()
    if be != "npy" and ((mode == "w" and not os.path.isdir(path + ".d")) or os.path.isfile(path)):
        return h5py.File(path, mode) if be == "h5py" else h5lite.File(path, mode)
    return _NpyDirFile(path, "a" if mode == "r+" else mode)


def exists(path):
    return os.path.isfile(path) or os.path.isdir(path + ".d")
