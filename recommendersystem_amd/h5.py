"""HDF5 (+ blosc) files at the data seams of the training path, over librsys_h5.so (include/rsys_h5.h).

    read_h5(fn)               ==  with h5py.File(fn) as f: {k: f[k][:] for k in f}     (transformer.py:86-89, model.py:380-383)
    write_h5(fn, d, blosc=3)  ==  h5open(fn, "w") do file; file[k, blosc = 3] = v; end  (transformer.jl:73-77, 228-231)

Arrays come back / go in with the file's row-major dimension order, i.e. what h5py shows the reference; a Julia
(M, V) matrix is the (V, M) array here.  The adapter is host-side C against the image's libhdf5 + c-blosc; it fails
loudly when the library was not built (`make -C recommendersystem_amd/csrc`).
"""
import ctypes
import os

import numpy as np

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "librsys_h5.so")
PLUGIN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "h5plugin")   # HDF5_PLUGIN_PATH for other clients

EXPORTED = ["rsys_h5_open", "rsys_h5_close", "rsys_h5_num_datasets", "rsys_h5_dataset_name", "rsys_h5_dataset_info",
            "rsys_h5_read", "rsys_h5_write", "rsys_h5_last_error"]

# enum rsys_h5_dtype
DTYPES = [np.float32, np.float64, np.int32, np.int64, np.uint8, np.int8, np.int16, np.uint16, np.uint32, np.uint64]
_CODE = {np.dtype(t): i for i, t in enumerate(DTYPES)}

_L = None


class H5Error(RuntimeError):
    pass


def lib():
    global _L
    if _L is None:
        if not os.path.exists(LIB_PATH):
            raise H5Error(f"{LIB_PATH} is not built (needs libhdf5 + c-blosc: make -C recommendersystem_amd/csrc)")
        L = ctypes.CDLL(LIB_PATH)
        vp, c = ctypes.c_void_p, ctypes.c_char_p
        i32p, i64p = ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int64)
        L.rsys_h5_open.argtypes = [c, ctypes.c_int, ctypes.POINTER(vp)]
        L.rsys_h5_close.argtypes = [vp]
        L.rsys_h5_num_datasets.argtypes = [vp, i32p]
        L.rsys_h5_dataset_name.argtypes = [vp, ctypes.c_int32, ctypes.c_char_p, ctypes.c_int32]
        L.rsys_h5_dataset_info.argtypes = [vp, c, i32p, i32p, i64p, i32p]
        L.rsys_h5_read.argtypes = [vp, c, vp, ctypes.c_int64]
        L.rsys_h5_write.argtypes = [vp, c, ctypes.c_int32, ctypes.c_int32, i64p, vp, ctypes.c_int32]
        L.rsys_h5_last_error.restype = c
        _L = L
    return _L


def _check(rc):
    if rc != 0:
        raise H5Error(f"rsys_h5 error {rc}: {lib().rsys_h5_last_error().decode()}")


class File:
    """`h5py.File(fn)` for the subset the path uses: iterate dataset names, read one whole, write one whole."""

    def __init__(self, path, mode="r"):
        assert mode in ("r", "w")
        self._h = ctypes.c_void_p()
        _check(lib().rsys_h5_open(os.fspath(path).encode(), 0 if mode == "r" else 1, ctypes.byref(self._h)))

    def close(self):
        if self._h:
            h, self._h = self._h, ctypes.c_void_p()
            _check(lib().rsys_h5_close(h))

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def keys(self):
        n = ctypes.c_int32()
        _check(lib().rsys_h5_num_datasets(self._h, ctypes.byref(n)))
        buf = ctypes.create_string_buffer(256)
        out = []
        for i in range(n.value):
            _check(lib().rsys_h5_dataset_name(self._h, i, buf, 256))
            out.append(buf.value.decode())
        return out

    def __iter__(self):
        return iter(self.keys())

    def info(self, name):
        """(numpy dtype, shape, blosc level or -1)"""
        dt, nd, lvl = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
        dims = (ctypes.c_int64 * 4)()
        _check(lib().rsys_h5_dataset_info(self._h, name.encode(), ctypes.byref(dt), ctypes.byref(nd), dims, ctypes.byref(lvl)))
        return np.dtype(DTYPES[dt.value]), tuple(dims[i] for i in range(nd.value)), lvl.value

    def __getitem__(self, name):
        dtype, shape, _ = self.info(name)
        out = np.empty(shape, dtype)
        _check(lib().rsys_h5_read(self._h, name.encode(), out.ctypes.data_as(ctypes.c_void_p), out.nbytes))
        return out

    def write(self, name, arr, blosc=3):
        a = np.asarray(arr)
        if a.ndim:
            a = np.ascontiguousarray(a)
        if a.dtype not in _CODE:
            raise H5Error(f"{name}: dtype {a.dtype} has no HDF5 mapping here")
        if a.ndim > 4:
            raise H5Error(f"{name}: rank {a.ndim} not supported")
        dims = (ctypes.c_int64 * 4)(*a.shape)
        _check(lib().rsys_h5_write(self._h, name.encode(), _CODE[a.dtype], a.ndim, dims,
                                   a.ctypes.data_as(ctypes.c_void_p), -1 if blosc is None else int(blosc)))


def read_h5(path):
    with File(path) as f:
        return {k: f[k] for k in f}


def write_h5(path, d, blosc=3):
    with File(path, "w") as f:
        for k, v in d.items():
            f.write(k, v, blosc)
