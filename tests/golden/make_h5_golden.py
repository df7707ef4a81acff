"""HDF5 fixtures written by REAL h5py, the library the reference uses for every file between its stages
(compute_features_hdf5.py:134-135, kmean_features.py:108, patch_gen_hdf5.py:119-120).

Run with an interpreter that has h5py -- in the build image:  /opt/conda/bin/python3.9 tests/golden/make_h5_golden.py
(h5py 3.3.0 / HDF5 1.10.6).  Writes tests/golden/h5py_features.h5 (the per-slide feature file layout) and
tests/golden/h5py_patches.hdf5 (the per-slide patch file layout); the arrays are seeded, the tests regenerate them."""
import os

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def arrays():
    rs = np.random.RandomState(41)
    feats = rs.randn(7, 16).astype(np.float32)
    clusters = rs.randn(100, 16).astype(np.float32)
    tiles = {f"{256 * i}_{512 * j}": rs.randint(0, 256, (8, 8, 3), dtype=np.uint8) for i in range(3) for j in range(2)}
    return feats, clusters, tiles


if __name__ == "__main__":
    feats, clusters, tiles = arrays()
    with h5py.File(os.path.join(HERE, "h5py_features.h5"), "w") as f:
        f.create_dataset("resnet_features", data=feats)
        f.create_dataset("cluster_features", data=clusters)
    with h5py.File(os.path.join(HERE, "h5py_patches.hdf5"), "w") as f:
        for name, t in tiles.items():
            f.create_dataset(name, data=t)
    print("wrote h5py fixtures with h5py", h5py.__version__, "HDF5", h5py.version.hdf5_version)
