"""A small h5py-shaped binding of the HDF5 C library (libhdf5, through ctypes) for the GQA data path.

The reference reads its question bytecode and object-feature chunks with h5py (data_pipeline.py:328-389,
batch_gqa_boxfeatures_pipeline.py:29-55) and writes the bytecode with it (gqa_preprocess.py:87-93).  h5py is not installed in
this image, but the HDF5 library itself is (libhdf5.so, HDF5 1.10), so the same files are read and written here through the C API.
Only what those call sites use is covered: `File(path, 'r' | 'w')` as a context manager, `file[name]` -> a dataset with
`.shape`, `.dtype`, `len()`, `[...]`, `[i]`, `[a:b]` (hyperslab reads along the first axis), `file.create_dataset(name, data=array)`,
`file.keys()`, `name in file`.  Numeric datasets only (integers and floats of 1 - 8 bytes).

`import_h5py()` returns the real h5py when it is importable and this module otherwise; both present the same surface.
"""

import ctypes
import ctypes.util
import glob
import os

import numpy as np

_lib = None
_hid = ctypes.c_int64            # hid_t is 64-bit since HDF5 1.10
_hsize = ctypes.c_uint64

H5F_ACC_RDONLY, H5F_ACC_TRUNC = 0x0000, 0x0002
H5P_DEFAULT = 0
H5S_ALL = 0
H5S_SELECT_SET = 0
H5T_INTEGER, H5T_FLOAT = 0, 1
H5T_SGN_NONE = 0


class H5Error(IOError):
    pass


def _find_library():
    names = []
    env = os.environ.get("DFOL_HDF5_LIB")
    if env:
        names.append(env)
    found = ctypes.util.find_library("hdf5")
    if found:
        names.append(found)
    for pat in ("/opt/conda/lib/libhdf5.so*", "/usr/lib/x86_64-linux-gnu/libhdf5*.so*", "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so*",
                "/usr/local/lib/libhdf5.so*"):
        names += sorted(p for p in glob.glob(pat) if "_hl" not in p and "_cpp" not in p and "fortran" not in p and "_tools" not in p)
    for n in names:
        try:
            return ctypes.CDLL(n)
        except OSError:
            continue
    return None


def available():
    try:
        _load()
        return True
    except H5Error:
        return False


def _load():
    global _lib
    if _lib is not None:
        return _lib
    lib = _find_library()
    if lib is None:
        raise H5Error("libhdf5 not found (set DFOL_HDF5_LIB to its path, or install h5py); .npz containers work without it")
    lib.H5open.restype = ctypes.c_int
    if lib.H5open() < 0:
        raise H5Error("H5open failed")
    maj, mnr, rel = ctypes.c_uint(), ctypes.c_uint(), ctypes.c_uint()
    lib.H5get_libversion(ctypes.byref(maj), ctypes.byref(mnr), ctypes.byref(rel))
    if (maj.value, mnr.value) < (1, 10):
        raise H5Error("HDF5 >= 1.10 needed (64-bit hid_t), found %d.%d.%d" % (maj.value, mnr.value, rel.value))
    sig = {
        "H5Fcreate": (_hid, [ctypes.c_char_p, ctypes.c_uint, _hid, _hid]),
        "H5Fopen": (_hid, [ctypes.c_char_p, ctypes.c_uint, _hid]),
        "H5Fclose": (ctypes.c_int, [_hid]),
        "H5Dopen2": (_hid, [_hid, ctypes.c_char_p, _hid]),
        "H5Dcreate2": (_hid, [_hid, ctypes.c_char_p, _hid, _hid, _hid, _hid, _hid]),
        "H5Dclose": (ctypes.c_int, [_hid]),
        "H5Dget_space": (_hid, [_hid]),
        "H5Dget_type": (_hid, [_hid]),
        "H5Dread": (ctypes.c_int, [_hid, _hid, _hid, _hid, _hid, ctypes.c_void_p]),
        "H5Dwrite": (ctypes.c_int, [_hid, _hid, _hid, _hid, _hid, ctypes.c_void_p]),
        "H5Screate_simple": (_hid, [ctypes.c_int, ctypes.POINTER(_hsize), ctypes.POINTER(_hsize)]),
        "H5Sclose": (ctypes.c_int, [_hid]),
        "H5Sget_simple_extent_ndims": (ctypes.c_int, [_hid]),
        "H5Sget_simple_extent_dims": (ctypes.c_int, [_hid, ctypes.POINTER(_hsize), ctypes.POINTER(_hsize)]),
        "H5Sselect_hyperslab": (ctypes.c_int, [_hid, ctypes.c_int, ctypes.POINTER(_hsize), ctypes.POINTER(_hsize), ctypes.POINTER(_hsize),
                                               ctypes.POINTER(_hsize)]),
        "H5Tclose": (ctypes.c_int, [_hid]),
        "H5Tget_class": (ctypes.c_int, [_hid]),
        "H5Tget_size": (ctypes.c_size_t, [_hid]),
        "H5Tget_sign": (ctypes.c_int, [_hid]),
        "H5Lexists": (ctypes.c_int, [_hid, ctypes.c_char_p, _hid]),
        "H5Gget_num_objs": (ctypes.c_int, [_hid, ctypes.POINTER(_hsize)]),
        "H5Gget_objname_by_idx": (ctypes.c_ssize_t, [_hid, _hsize, ctypes.c_char_p, ctypes.c_size_t]),
        "H5Eset_auto2": (ctypes.c_int, [_hid, ctypes.c_void_p, ctypes.c_void_p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    lib.H5Eset_auto2(0, None, None)                          # errors are reported through return codes -> H5Error, not printed
    _lib = lib
    return lib


def _native_type(dtype):
    """The library's native memory type id for a numpy dtype (H5T_NATIVE_* are globals initialised by H5open)."""
    lib = _load()
    dtype = np.dtype(dtype)
    names = {("i", 1): "H5T_NATIVE_INT8_g", ("i", 2): "H5T_NATIVE_INT16_g", ("i", 4): "H5T_NATIVE_INT32_g", ("i", 8): "H5T_NATIVE_INT64_g",
             ("u", 1): "H5T_NATIVE_UINT8_g", ("u", 2): "H5T_NATIVE_UINT16_g", ("u", 4): "H5T_NATIVE_UINT32_g", ("u", 8): "H5T_NATIVE_UINT64_g",
             ("f", 4): "H5T_NATIVE_FLOAT_g", ("f", 8): "H5T_NATIVE_DOUBLE_g", ("b", 1): "H5T_NATIVE_INT8_g"}
    key = (dtype.kind, dtype.itemsize)
    if key not in names:
        raise H5Error("unsupported dtype %s (integers and floats of 1-8 bytes only)" % dtype)
    return _hid.in_dll(lib, names[key]).value


class Dataset(object):
    def __init__(self, file, name):
        lib = _load()
        self._file, self.name = file, name
        self._id = lib.H5Dopen2(file._id, name.encode(), H5P_DEFAULT)
        if self._id < 0:
            raise KeyError(name)
        space = lib.H5Dget_space(self._id)
        nd = lib.H5Sget_simple_extent_ndims(space)
        dims = (_hsize * max(nd, 1))()
        if nd > 0:
            lib.H5Sget_simple_extent_dims(space, dims, None)
        lib.H5Sclose(space)
        self.shape = tuple(int(d) for d in dims[:nd])
        t = lib.H5Dget_type(self._id)
        cls, size = lib.H5Tget_class(t), int(lib.H5Tget_size(t))
        sign = lib.H5Tget_sign(t) if cls == H5T_INTEGER else 1
        lib.H5Tclose(t)
        if cls == H5T_INTEGER:
            self.dtype = np.dtype("%s%d" % ("u" if sign == H5T_SGN_NONE else "i", size))
        elif cls == H5T_FLOAT and size in (4, 8):
            self.dtype = np.dtype("f%d" % size)
        else:
            raise H5Error("dataset %s: only integer and float32/64 datasets are supported" % name)

    def __len__(self):
        return self.shape[0]

    @property
    def ndim(self):
        return len(self.shape)

    def _read(self, start, count):
        """Rows [start, start + count) along the first axis (the whole dataset when it is 0-dimensional)."""
        lib = _load()
        if not self.shape:
            out = np.empty((), self.dtype)
            rc = lib.H5Dread(self._id, _native_type(self.dtype), H5S_ALL, H5S_ALL, H5P_DEFAULT, out.ctypes.data_as(ctypes.c_void_p))
            if rc < 0:
                raise H5Error("H5Dread failed on %s" % self.name)
            return out
        shape = (count,) + self.shape[1:]
        out = np.empty(shape, self.dtype)
        if out.size == 0:
            return out
        nd = len(self.shape)
        fspace = lib.H5Dget_space(self._id)
        st = (_hsize * nd)(*([start] + [0] * (nd - 1)))
        cn = (_hsize * nd)(*shape)
        lib.H5Sselect_hyperslab(fspace, H5S_SELECT_SET, st, None, cn, None)
        mspace = lib.H5Screate_simple(nd, cn, None)
        rc = lib.H5Dread(self._id, _native_type(self.dtype), mspace, fspace, H5P_DEFAULT, out.ctypes.data_as(ctypes.c_void_p))
        lib.H5Sclose(mspace)
        lib.H5Sclose(fspace)
        if rc < 0:
            raise H5Error("H5Dread failed on %s" % self.name)
        return out

    def __getitem__(self, key):
        if key is Ellipsis or (isinstance(key, tuple) and len(key) == 0):
            return self._read(0, self.shape[0] if self.shape else 0)
        rest = ()
        if isinstance(key, tuple):
            key, rest = key[0], key[1:]
            if key is Ellipsis:
                return self._read(0, self.shape[0])[(Ellipsis,) + rest]
        if isinstance(key, (int, np.integer)):
            n = self.shape[0]
            i = int(key) + (n if key < 0 else 0)
            if not 0 <= i < n:
                raise IndexError(key)
            row = self._read(i, 1)[0]
            return row[rest] if rest else row
        if isinstance(key, slice):
            a, b, step = key.indices(self.shape[0])
            if step == 1:
                block = self._read(a, max(0, b - a))
            else:
                lo, hi = (a, b) if step > 0 else (b + 1, a + 1)
                block = self._read(lo, max(0, hi - lo))[::step] if step > 0 else self._read(lo, max(0, hi - lo))[::-1][::-step]
            return block[(slice(None),) + rest] if rest else block
        # index arrays / lists along the first axis: read the covering range once
        idx = np.asarray(key)
        if idx.dtype == bool:
            idx = np.nonzero(idx)[0]
        if idx.size == 0:
            return np.empty((0,) + self.shape[1:], self.dtype)
        lo, hi = int(idx.min()), int(idx.max()) + 1
        block = self._read(lo, hi - lo)[idx - lo]
        return block[(slice(None),) + rest] if rest else block

    def __array__(self, dtype=None):
        a = self[...]
        return a if dtype is None else a.astype(dtype)

    def close(self):
        if self._id is not None and self._id >= 0:
            _load().H5Dclose(self._id)
        self._id = None


class File(object):
    """h5py.File for the two modes the reference uses: 'r' and 'w' (truncate)."""

    def __init__(self, path, mode="r"):
        lib = _load()
        self.filename, self.mode = path, mode
        self._datasets = {}
        if mode == "r":
            self._id = lib.H5Fopen(os.fsencode(path), H5F_ACC_RDONLY, H5P_DEFAULT)
        elif mode == "w":
            self._id = lib.H5Fcreate(os.fsencode(path), H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT)
        else:
            raise ValueError("mode must be 'r' or 'w'")
        if self._id < 0:
            self._id = None
            raise H5Error("cannot open %s (mode %s)" % (path, mode))

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __contains__(self, name):
        return self._id is not None and _load().H5Lexists(self._id, name.encode(), H5P_DEFAULT) > 0

    def __getitem__(self, name):
        if name not in self._datasets:
            if name not in self:
                raise KeyError(name)
            self._datasets[name] = Dataset(self, name)
        return self._datasets[name]

    def keys(self):
        lib = _load()
        n = _hsize()
        lib.H5Gget_num_objs(self._id, ctypes.byref(n))
        out = []
        for i in range(n.value):
            size = lib.H5Gget_objname_by_idx(self._id, i, None, 0)
            buf = ctypes.create_string_buffer(size + 1)
            lib.H5Gget_objname_by_idx(self._id, i, buf, size + 1)
            out.append(buf.value.decode())
        return out

    @property
    def files(self):                                          # np.load-style listing, so either container can be enumerated alike
        return self.keys()

    def create_dataset(self, name, data=None, shape=None, dtype=None):
        lib = _load()
        if self.mode != "w":
            raise H5Error("file is open read-only")
        if data is None:
            data = np.zeros(shape, dtype or np.float32)
        arr = np.ascontiguousarray(data if dtype is None else np.asarray(data, dtype))
        if arr.dtype == np.bool_:
            arr = arr.astype(np.int8)
        nd = arr.ndim
        dims = (_hsize * max(nd, 1))(*arr.shape)
        space = lib.H5Screate_simple(nd, dims, None)
        tid = _native_type(arr.dtype)
        did = lib.H5Dcreate2(self._id, name.encode(), tid, space, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT)
        if did < 0:
            lib.H5Sclose(space)
            raise H5Error("cannot create dataset %s" % name)
        rc = 0
        if arr.size:
            rc = lib.H5Dwrite(did, tid, H5S_ALL, H5S_ALL, H5P_DEFAULT, arr.ctypes.data_as(ctypes.c_void_p))
        lib.H5Dclose(did)
        lib.H5Sclose(space)
        if rc < 0:
            raise H5Error("H5Dwrite failed on %s" % name)
        return self[name]

    def close(self):
        for d in self._datasets.values():
            d.close()
        self._datasets = {}
        if self._id is not None:
            _load().H5Fclose(self._id)
            self._id = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def import_h5py():
    """The real h5py when it is installed; this module (same surface for the data path's needs) otherwise."""
    try:
        import h5py
        if hasattr(h5py, "File"):
            return h5py
    except ImportError:
        pass
    _load()
    import sys
    return sys.modules[__name__]
