"""Feature store with the reference's on-disk names.

The reference keeps per-slide tensors in HDF5 files ``<feature_path>/<project>/<WSI>/<WSI>.h5`` with datasets
``resnet_features | uni_features [n, D]`` and ``cluster_features [100, D]`` (compute_features_hdf5.py:134-135,
kmean_features.py:108), and patches in ``<patch_path>/<slide>/<slide>.hdf5`` with one uint8 ``[S, S, 3]``
dataset per tile named ``"{x}_{y}"`` (patch_gen_hdf5.py:119-120).  ``h5py`` is used when importable (real
SEQUOIA stores then work unchanged); otherwise the same dataset names live as ``.npy`` files inside a
directory with the HDF5 file's path + ``.d`` -- same keys, same dtypes, same shapes."""
import os

import numpy as np

try:
    import h5py  # noqa: F401
    HAVE_H5PY = True
except Exception:
    h5py = None
    HAVE_H5PY = False


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
    if HAVE_H5PY and ((mode == "w" and not os.path.isdir(path + ".d")) or os.path.isfile(path)):
        return h5py.File(path, mode)
    return _NpyDirFile(path, "a" if mode == "r+" else mode)


def exists(path):
    return os.path.isfile(path) or os.path.isdir(path + ".d")
