"""h5py.File-shaped access to REAL HDF5 files through the system's HDF5 C library (ctypes on libhdf5.so).

The reference keeps everything between its stages in HDF5 (``h5py.File``): patches
(pre_processing/patch_gen_hdf5.py:66,119-120), per-slide features (compute_features_hdf5.py:110,134-135;
kmean_features.py:75-80,108) and the training reads (src/read_data.py:47-49, src/utils.py:30-33).  h5py is not
installed in this image, but the HDF5 library itself is (conda's ``libhdf5.so`` 1.10); this module binds the dozen C
entry points those call sites need, so the stores written here are ordinary HDF5 files that h5py / the reference
read unchanged, and real SEQUOIA feature files can be consumed as they are.

Subset: ``File(path, mode in {"r", "r+", "w", "a"})``, ``keys()``, ``in``, ``f[name]`` -> dataset with ``[...]``,
``shape``, ``dtype`` and ``np.asarray``; ``create_dataset(name, data=array)`` (contiguous layout, native byte order);
context manager.  Numeric dtypes: (u)int8/16/32/64, float32/64.  No groups below the root, no attributes, no
filters -- none of the reference's call sites use them."""
import ctypes
import ctypes.util
import glob
import os

import numpy as np

hid_t = ctypes.c_int64
hsize_t = ctypes.c_uint64

_CANDIDATES = ["/opt/conda/lib/libhdf5.so*", "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so*", "/usr/lib/x86_64-linux-gnu/libhdf5*.so*",
               "/usr/lib64/libhdf5.so*", "/usr/local/lib/libhdf5.so*"]
_lib = None
_lib_error = None

H5F_ACC_RDONLY, H5F_ACC_RDWR, H5F_ACC_TRUNC, H5F_ACC_EXCL = 0, 1, 2, 4
H5T_INTEGER, H5T_FLOAT = 0, 1


def _find():
    env = os.environ.get("SEQUOIA_HDF5_LIB")
    if env:
        return [env]
    paths = []
    name = ctypes.util.find_library("hdf5")
    if name:
        paths.append(name)
    for pat in _CANDIDATES:
        paths += sorted(p for p in glob.glob(pat) if "_hl" not in p and "_cpp" not in p and "fortran" not in p)
    return paths


def library():
    """The loaded libhdf5 (ctypes.CDLL) or None when no usable HDF5 C library is present."""
    global _lib, _lib_error
    if _lib is not None or _lib_error is not None:
        return _lib
    for path in _find():
        try:
            lib = ctypes.CDLL(path)
            if lib.H5open() < 0:
                continue
            _declare(lib)
            _lib = lib
            return _lib
        except OSError as e:
            _lib_error = e
        except AttributeError as e:          # a library without the symbols we need
            _lib_error = e
    _lib_error = _lib_error or OSError("no libhdf5 found")
    return None


def available():
    return library() is not None


_ITER_CB = ctypes.CFUNCTYPE(ctypes.c_int, hid_t, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_void_p)


def _declare(lib):
    c, vp = ctypes, ctypes.c_void_p
    sig = {
        "H5Fcreate": (hid_t, [c.c_char_p, c.c_uint, hid_t, hid_t]), "H5Fopen": (hid_t, [c.c_char_p, c.c_uint, hid_t]),
        "H5Fclose": (c.c_int, [hid_t]), "H5Fflush": (c.c_int, [hid_t, c.c_int]),
        "H5Screate_simple": (hid_t, [c.c_int, c.POINTER(hsize_t), c.POINTER(hsize_t)]), "H5Sclose": (c.c_int, [hid_t]),
        "H5Dcreate2": (hid_t, [hid_t, c.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]),
        "H5Dopen2": (hid_t, [hid_t, c.c_char_p, hid_t]), "H5Dclose": (c.c_int, [hid_t]),
        "H5Dwrite": (c.c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, vp]), "H5Dread": (c.c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, vp]),
        "H5Dget_space": (hid_t, [hid_t]), "H5Dget_type": (hid_t, [hid_t]),
        "H5Sget_simple_extent_ndims": (c.c_int, [hid_t]),
        "H5Sget_simple_extent_dims": (c.c_int, [hid_t, c.POINTER(hsize_t), c.POINTER(hsize_t)]),
        "H5Tget_class": (c.c_int, [hid_t]), "H5Tget_size": (c.c_size_t, [hid_t]), "H5Tget_sign": (c.c_int, [hid_t]),
        "H5Tclose": (c.c_int, [hid_t]), "H5Lexists": (c.c_int, [hid_t, c.c_char_p, hid_t]),
        "H5Eset_auto2": (c.c_int, [hid_t, vp, vp]),
        "H5Pcreate": (hid_t, [hid_t]), "H5Pclose": (c.c_int, [hid_t]), "H5Pset_obj_track_times": (c.c_int, [hid_t, c.c_uint]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    it = getattr(lib, "H5Literate", None) or getattr(lib, "H5Literate1")
    it.restype, it.argtypes = c.c_int, [hid_t, c.c_int, c.c_int, c.POINTER(hsize_t), _ITER_CB, vp]
    lib._sq_iterate = it
    lib.H5Eset_auto2(0, None, None)          # errors come back as negative ids; no stderr stack dumps


_NATIVE = {"u1": "UINT8", "i1": "INT8", "u2": "UINT16", "i2": "INT16", "u4": "UINT32", "i4": "INT32", "u8": "UINT64",
           "i8": "INT64", "f4": "FLOAT", "f8": "DOUBLE"}


def _native_type(dtype):
    dt = np.dtype(dtype)
    key = dt.kind + str(dt.itemsize)
    if key not in _NATIVE:
        raise TypeError(f"h5lite: dtype {dt} is not supported (numeric (u)int8..64 / float32 / float64 only)")
    return hid_t.in_dll(library(), f"H5T_NATIVE_{_NATIVE[key]}_g").value


class Dataset:
    """A dataset handle that behaves like the slice of h5py the call sites use: ``ds[:]``, ``ds[...]``, ``ds[i:j]``,
    ``np.asarray(ds)``, ``.shape``, ``.dtype``.  The whole dataset is read on first access (features are a few MB)."""

    def __init__(self, lib, fid, name):
        self._lib, self.name = lib, name
        did = lib.H5Dopen2(fid, name.encode(), 0)
        if did < 0:
            raise KeyError(f"Unable to open object (object '{name}' doesn't exist)")
        try:
            sid = lib.H5Dget_space(did)
            nd = lib.H5Sget_simple_extent_ndims(sid)
            dims = (hsize_t * max(nd, 1))()
            if nd > 0:
                lib.H5Sget_simple_extent_dims(sid, dims, None)
            lib.H5Sclose(sid)
            tid = lib.H5Dget_type(did)
            cls, size, sign = lib.H5Tget_class(tid), lib.H5Tget_size(tid), lib.H5Tget_sign(tid)
            lib.H5Tclose(tid)
            if cls == H5T_FLOAT:
                dt = np.dtype(f"f{size}")
            elif cls == H5T_INTEGER:
                dt = np.dtype(("i" if sign == 1 else "u") + str(size))
            else:
                raise TypeError(f"h5lite: dataset '{name}' has HDF5 type class {cls}; only integer / float datasets are supported")
            self.shape = tuple(int(dims[i]) for i in range(nd))
            self.dtype = dt
            buf = np.empty(self.shape, dtype=dt)
            if buf.size and lib.H5Dread(did, _native_type(dt), 0, 0, 0, buf.ctypes.data_as(ctypes.c_void_p)) < 0:
                raise OSError(f"h5lite: reading dataset '{name}' failed")
            self._data = buf
        finally:
            lib.H5Dclose(did)

    def __getitem__(self, key):
        return self._data[key]

    def __array__(self, dtype=None, copy=None):
        return self._data if dtype is None else self._data.astype(dtype)

    def __len__(self):
        return self.shape[0]


class File:
    def __init__(self, path, mode="r"):
        lib = library()
        if lib is None:
            raise OSError(f"h5lite: no HDF5 C library available ({_lib_error})")
        self._lib, self.filename, self.mode = lib, path, mode
        p = os.fsencode(path)
        if mode == "r":
            fid = lib.H5Fopen(p, H5F_ACC_RDONLY, 0)
        elif mode == "r+":
            fid = lib.H5Fopen(p, H5F_ACC_RDWR, 0)
        elif mode == "w":
            fid = lib.H5Fcreate(p, H5F_ACC_TRUNC, 0, 0)
        elif mode == "a":
            fid = lib.H5Fopen(p, H5F_ACC_RDWR, 0) if os.path.exists(path) else lib.H5Fcreate(p, H5F_ACC_EXCL, 0, 0)
        else:
            raise ValueError(f"h5lite: mode {mode!r}")
        if fid < 0:
            raise OSError(f"Unable to open file (unable to open file: name = '{path}', mode = '{mode}')")
        self._fid = fid

    # -- reading -------------------------------------------------------------------------------------------------
    def keys(self):
        names = []

        @_ITER_CB
        def cb(g, name, info, data):
            names.append(name.decode())
            return 0
        idx = hsize_t(0)
        if self._lib._sq_iterate(self._fid, 0, 0, ctypes.byref(idx), cb, None) < 0:      # H5_INDEX_NAME, H5_ITER_INC
            raise OSError("h5lite: link iteration failed")
        return names

    def __iter__(self):
        return iter(self.keys())

    def __len__(self):
        return len(self.keys())

    def __contains__(self, name):
        return self._lib.H5Lexists(self._fid, name.encode(), 0) > 0

    def __getitem__(self, name):
        if name not in self:
            raise KeyError(f"Unable to open object (object '{name}' doesn't exist)")
        return Dataset(self._lib, self._fid, name)

    # -- writing -------------------------------------------------------------------------------------------------
    def create_dataset(self, name, data=None, shape=None, dtype=None):
        if self.mode == "r":
            raise OSError("h5lite: file opened read-only")
        if data is None:
            data = np.zeros(shape, dtype=dtype or np.float32)
        arr = np.ascontiguousarray(np.asarray(data))
        if dtype is not None:
            arr = np.ascontiguousarray(arr.astype(dtype))
        if arr.dtype.byteorder == ">":
            arr = arr.astype(arr.dtype.newbyteorder("="))
        lib = self._lib
        if name in self:
            raise ValueError(f"Unable to create dataset (name already exists): {name}")
        tid = _native_type(arr.dtype)
        dims = (hsize_t * max(arr.ndim, 1))(*arr.shape)
        sid = lib.H5Screate_simple(arr.ndim, dims, None)
        # h5py's default (track_times=False): no creation / modification stamps in the object header, so a file's bytes are a
        # function of its contents -- the files of a sharded run can be compared with the one-rank run's byte for byte
        try:
            dcpl = lib.H5Pcreate(hid_t.in_dll(lib, "H5P_CLS_DATASET_CREATE_ID_g").value)
        except ValueError:                       # a build that does not export the property-list class id: default creation properties
            dcpl = -1
        if dcpl >= 0:
            lib.H5Pset_obj_track_times(dcpl, 0)
        did = lib.H5Dcreate2(self._fid, name.encode(), tid, sid, 0, max(dcpl, 0), 0)
        try:
            if did < 0:
                raise OSError(f"h5lite: creating dataset '{name}' failed")
            if arr.size and lib.H5Dwrite(did, tid, 0, 0, 0, arr.ctypes.data_as(ctypes.c_void_p)) < 0:
                raise OSError(f"h5lite: writing dataset '{name}' failed")
        finally:
            if did >= 0:
                lib.H5Dclose(did)
            if dcpl >= 0:
                lib.H5Pclose(dcpl)
            lib.H5Sclose(sid)
        return Dataset(lib, self._fid, name)

    def flush(self):
        self._lib.H5Fflush(self._fid, 1)

    def close(self):
        if self._fid is not None:
            self._lib.H5Fclose(self._fid)
            self._fid = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            if getattr(self, "_fid", None) is not None:
                self.close()
        except Exception:
            pass
