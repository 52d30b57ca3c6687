"""HDF5 file access for the writer callbacks (hdf5.py): the small part of ``h5py``'s interface they and their tests
use -- ``File(name, mode)``, ``create_dataset``, ``f[name][...]``, ``f.attrs[...]``.

``h5py`` is used when it is importable.  Otherwise the same calls go to the HDF5 C library through ``ctypes``
(``libhdf5`` >= 1.10: 64-bit ``hid_t``); it is looked for in ``$LPA_HDF5_LIB``, the loader's search path and the usual
prefixes.  Neither present: ``ImportError`` when a writer is constructed -- nothing is written in another format.

Files come out as h5py would write them from the reference's calls (`callback/hdf5.py:218-273,362-372,675-699`):
chunked or contiguous native-endian datasets, Python ``int`` / ``float`` attributes as scalar int64 / float64, ``str``
attributes as variable-length UTF-8 strings.
"""
from __future__ import annotations

import ctypes as C
import ctypes.util
import os

import numpy as np

try:                                   # the real thing
    import h5py as _h5py
except ImportError:
    _h5py = None

_hid, _hsize = C.c_int64, C.c_uint64
_LIB = None
_CANDIDATES = ("libhdf5.so", "libhdf5_serial.so", "/usr/lib/x86_64-linux-gnu/libhdf5_serial.so", "/usr/lib64/libhdf5.so",
               "/usr/local/lib/libhdf5.so", "/opt/conda/lib/libhdf5.so")


class _Lib:
    """libhdf5 bound through ctypes"""

    _SIGS = {
        "H5open": (C.c_int, []),
        "H5get_libversion": (C.c_int, [C.POINTER(C.c_uint)] * 3),
        "H5Fcreate": (_hid, [C.c_char_p, C.c_uint, _hid, _hid]),
        "H5Fopen": (_hid, [C.c_char_p, C.c_uint, _hid]),
        "H5Fclose": (C.c_int, [_hid]),
        "H5Screate_simple": (_hid, [C.c_int, C.POINTER(_hsize), C.POINTER(_hsize)]),
        "H5Screate": (_hid, [C.c_int]),
        "H5Sclose": (C.c_int, [_hid]),
        "H5Sget_simple_extent_ndims": (C.c_int, [_hid]),
        "H5Sget_simple_extent_dims": (C.c_int, [_hid, C.POINTER(_hsize), C.POINTER(_hsize)]),
        "H5Sselect_hyperslab": (C.c_int, [_hid, C.c_int] + [C.POINTER(_hsize)] * 4),
        "H5Dcreate2": (_hid, [_hid, C.c_char_p, _hid, _hid, _hid, _hid, _hid]),
        "H5Dopen2": (_hid, [_hid, C.c_char_p, _hid]),
        "H5Dwrite": (C.c_int, [_hid, _hid, _hid, _hid, _hid, C.c_void_p]),
        "H5Dread": (C.c_int, [_hid, _hid, _hid, _hid, _hid, C.c_void_p]),
        "H5Dget_space": (_hid, [_hid]),
        "H5Dget_type": (_hid, [_hid]),
        "H5Dclose": (C.c_int, [_hid]),
        "H5Dvlen_reclaim": (C.c_int, [_hid, _hid, _hid, C.c_void_p]),
        "H5Pcreate": (_hid, [_hid]),
        "H5Pset_chunk": (C.c_int, [_hid, C.c_int, C.POINTER(_hsize)]),
        "H5Pclose": (C.c_int, [_hid]),
        "H5Tcopy": (_hid, [_hid]),
        "H5Tset_size": (C.c_int, [_hid, C.c_size_t]),
        "H5Tset_cset": (C.c_int, [_hid, C.c_int]),
        "H5Tget_class": (C.c_int, [_hid]),
        "H5Tget_size": (C.c_size_t, [_hid]),
        "H5Tget_sign": (C.c_int, [_hid]),
        "H5Tis_variable_str": (C.c_int, [_hid]),
        "H5Tclose": (C.c_int, [_hid]),
        "H5Acreate2": (_hid, [_hid, C.c_char_p, _hid, _hid, _hid, _hid]),
        "H5Aopen": (_hid, [_hid, C.c_char_p, _hid]),
        "H5Awrite": (C.c_int, [_hid, _hid, C.c_void_p]),
        "H5Aread": (C.c_int, [_hid, _hid, C.c_void_p]),
        "H5Aget_type": (_hid, [_hid]),
        "H5Aget_space": (_hid, [_hid]),
        "H5Aexists": (C.c_int, [_hid, C.c_char_p]),
        "H5Adelete": (C.c_int, [_hid, C.c_char_p]),
        "H5Aclose": (C.c_int, [_hid]),
        "H5Lexists": (C.c_int, [_hid, C.c_char_p, _hid]),
    }
    _ITER = C.CFUNCTYPE(C.c_int, _hid, C.c_char_p, C.c_void_p, C.c_void_p)

    def __init__(self, path):
        self.dll = C.CDLL(path)
        # libhdf5 >= 1.12 exports only the versioned H5Literate1 / H5Literate2 (the callback here ignores the info struct, so
        # either serves) and deprecates H5Dvlen_reclaim in favour of H5Treclaim (same signature): take what the library has
        alias = {"H5Dvlen_reclaim": ("H5Dvlen_reclaim", "H5Treclaim"), "H5Literate": ("H5Literate", "H5Literate2", "H5Literate1")}

        def sym(name):
            for cand in alias.get(name, (name,)):
                if hasattr(self.dll, cand):
                    return getattr(self.dll, cand)
            raise AttributeError(f"{path}: none of {alias.get(name, (name,))} exported")

        for name, (res, args) in self._SIGS.items():
            fn = sym(name)
            fn.restype, fn.argtypes = res, args
            setattr(self, name, fn)
        for name in ("H5Literate", "H5Aiterate2"):
            fn = sym(name)
            fn.restype = C.c_int
            fn.argtypes = [_hid, C.c_int, C.c_int, C.POINTER(_hsize), self._ITER, C.c_void_p]
            setattr(self, name, fn)
        if self.H5open() < 0:
            raise OSError("H5open failed")
        v = [C.c_uint() for _ in range(3)]
        self.H5get_libversion(*[C.byref(x) for x in v])
        self.version = tuple(x.value for x in v)
        if self.version < (1, 10, 0):
            raise OSError(f"libhdf5 {self.version}: need >= 1.10 (64-bit identifiers)")
        g = lambda s: _hid.in_dll(self.dll, s).value                      # noqa: E731
        self.types = {np.dtype("f8"): g("H5T_NATIVE_DOUBLE_g"), np.dtype("f4"): g("H5T_NATIVE_FLOAT_g"),
                      np.dtype("i8"): g("H5T_NATIVE_INT64_g"), np.dtype("u8"): g("H5T_NATIVE_UINT64_g"),
                      np.dtype("i4"): g("H5T_NATIVE_INT32_g"), np.dtype("u4"): g("H5T_NATIVE_UINT32_g"),
                      np.dtype("u1"): g("H5T_NATIVE_UINT8_g"), np.dtype("i1"): g("H5T_NATIVE_INT8_g")}
        self.c_s1 = g("H5T_C_S1_g")
        self.dcpl_class = g("H5P_CLS_DATASET_CREATE_ID_g")

    def check(self, rc, what):
        if rc < 0:
            raise OSError(f"HDF5: {what} failed")
        return rc

    def vlen_str(self):
        t = self.check(self.H5Tcopy(self.c_s1), "H5Tcopy")
        self.H5Tset_size(t, C.c_size_t(-1).value)          # H5T_VARIABLE
        self.H5Tset_cset(t, 1)                             # H5T_CSET_UTF8
        return t

    def np_dtype(self, t):
        """numpy dtype of a stored numeric type (read through the native type of the same class / size)"""
        cls, size = self.H5Tget_class(t), self.H5Tget_size(t)
        if cls == 1:
            return np.dtype(f"f{size}")
        if cls == 0:
            return np.dtype(("i" if self.H5Tget_sign(t) else "u") + str(size))
        raise TypeError(f"HDF5 type class {cls} is not supported")

    def names(self, iterate, loc):
        out = []
        cb = self._ITER(lambda _loc, name, _info, _data: out.append(name.decode()) or 0)
        n = _hsize(0)
        self.check(iterate(loc, 0, 0, C.byref(n), cb, None), "iterate")     # H5_INDEX_NAME, H5_ITER_INC
        return out


def _lib():
    global _LIB
    if _LIB is None:
        tried = []
        for cand in (os.environ.get("LPA_HDF5_LIB"), ctypes.util.find_library("hdf5"), *_CANDIDATES):
            if not cand:
                continue
            try:
                _LIB = _Lib(cand)
                break
            except (OSError, AttributeError, ValueError) as e:
                tried.append(f"{cand}: {e}")
        else:
            raise ImportError("HDF5 output needs h5py or libhdf5 >= 1.10 (set LPA_HDF5_LIB); tried " + "; ".join(tried))
    return _LIB


def require():
    """``ImportError`` unless files can be written"""
    if _h5py is None:
        _lib()


def available() -> bool:
    if _h5py is not None:
        return True
    try:
        _lib()
        return True
    except ImportError:
        return False


def backend() -> str:
    return "h5py" if _h5py is not None else "libhdf5 %d.%d.%d (ctypes)" % _lib().version


def _dims(shape):
    return (_hsize * max(len(shape), 1))(*shape)


class _Attrs:
    def __init__(self, L, loc):
        self.L, self.loc = L, loc

    def __contains__(self, name):
        return self.L.check(self.L.H5Aexists(self.loc, name.encode()), "H5Aexists") > 0

    def keys(self):
        return self.L.names(self.L.H5Aiterate2, self.loc)

    def __iter__(self):
        return iter(self.keys())

    def __setitem__(self, name, value):
        L = self.L
        if name in self:
            L.check(L.H5Adelete(self.loc, name.encode()), "H5Adelete")
        space = L.check(L.H5Screate(0), "H5Screate")                         # H5S_SCALAR
        if isinstance(value, (str, bytes)):
            raw = value.encode() if isinstance(value, str) else value
            t, own = L.vlen_str(), True
            buf = C.c_char_p(raw)
            ptr = C.cast(C.pointer(buf), C.c_void_p)
        else:
            arr = np.asarray(value)
            if arr.ndim:
                raise TypeError("only scalar attributes are supported")
            if arr.dtype.kind == "b":
                arr = arr.astype("i1")
            if arr.dtype not in L.types:
                raise TypeError(f"attribute of dtype {arr.dtype}")
            arr = np.ascontiguousarray(arr)
            t, own, ptr = L.types[arr.dtype], False, arr.ctypes.data_as(C.c_void_p)
        a = L.check(L.H5Acreate2(self.loc, name.encode(), t, space, 0, 0), "H5Acreate2")
        try:
            L.check(L.H5Awrite(a, t, ptr), "H5Awrite")
        finally:
            L.H5Aclose(a)
            L.H5Sclose(space)
            if own:
                L.H5Tclose(t)

    def __getitem__(self, name):
        L = self.L
        if name not in self:
            raise KeyError(name)
        a = L.check(L.H5Aopen(self.loc, name.encode(), 0), "H5Aopen")
        t, space = L.H5Aget_type(a), L.H5Aget_space(a)
        try:
            if L.H5Sget_simple_extent_ndims(space) != 0:
                raise TypeError("only scalar attributes are supported")
            if L.H5Tget_class(t) == 3:                                       # H5T_STRING
                if L.H5Tis_variable_str(t) > 0:
                    buf = C.c_char_p()
                    L.check(L.H5Aread(a, t, C.byref(buf)), "H5Aread")
                    out = (buf.value or b"").decode()
                    L.H5Dvlen_reclaim(t, space, 0, C.byref(buf))
                    return out
                raw = C.create_string_buffer(L.H5Tget_size(t) + 1)
                L.check(L.H5Aread(a, t, raw), "H5Aread")
                return raw.value.decode()
            out = np.empty((), L.np_dtype(t))
            L.check(L.H5Aread(a, L.types[out.dtype], out.ctypes.data_as(C.c_void_p)), "H5Aread")
            return out[()]
        finally:
            L.H5Tclose(t)
            L.H5Sclose(space)
            L.H5Aclose(a)


class _Dataset:
    def __init__(self, L, did):
        self.L, self.id = L, did
        space, t = L.H5Dget_space(did), L.H5Dget_type(did)
        nd = L.check(L.H5Sget_simple_extent_ndims(space), "ndims")
        dims = _dims((0,) * nd)
        L.H5Sget_simple_extent_dims(space, dims, None)
        self.shape = tuple(int(dims[k]) for k in range(nd))
        self.dtype = L.np_dtype(t)
        L.H5Tclose(t)
        L.H5Sclose(space)
        self.attrs = _Attrs(L, did)

    def __len__(self):
        return self.shape[0]

    @property
    def size(self):
        return int(np.prod(self.shape))

    def _block(self, idx):
        """(start, count) of a selection made of unit-stride slices / ints, or None"""
        if not isinstance(idx, tuple):
            idx = (idx,)
        if any(i is Ellipsis for i in idx):
            k = idx.index(Ellipsis)
            idx = idx[:k] + (slice(None),) * (len(self.shape) - len(idx) + 1) + idx[k + 1:]
        idx = idx + (slice(None),) * (len(self.shape) - len(idx))
        start, count, drop = [], [], []
        for i, n in zip(idx, self.shape):
            if isinstance(i, (int, np.integer)):
                i = int(i) + (n if i < 0 else 0)
                start.append(i), count.append(1), drop.append(True)
            elif isinstance(i, slice):
                a, b, st = i.indices(n)
                if st != 1:
                    return None
                start.append(a), count.append(max(b - a, 0)), drop.append(False)
            else:
                return None
        return start, count, drop

    def _io(self, fn, start, count, arr):
        L = self.L
        fspace = L.H5Dget_space(self.id)
        mspace = L.check(L.H5Screate_simple(len(count), _dims(count), None), "H5Screate_simple")
        try:
            L.check(L.H5Sselect_hyperslab(fspace, 0, _dims(start), None, _dims(count), None), "H5Sselect_hyperslab")
            L.check(fn(self.id, L.types[arr.dtype], mspace, fspace, 0, arr.ctypes.data_as(C.c_void_p)), "dataset I/O")
        finally:
            L.H5Sclose(mspace)
            L.H5Sclose(fspace)

    def __getitem__(self, idx):
        blk = self._block(idx)
        if blk is None or not self.shape:      # strided / fancy selections: read everything, let numpy select
            out = np.empty(self.shape, self.dtype)
            if out.size:
                self.L.check(self.L.H5Dread(self.id, self.L.types[self.dtype], 0, 0, 0,
                                            out.ctypes.data_as(C.c_void_p)), "H5Dread")
            return out[idx]
        start, count, drop = blk
        out = np.empty(count, self.dtype)
        if out.size:
            self._io(self.L.H5Dread, start, count, out)
        return out.reshape([c for c, d in zip(count, drop) if not d])

    def __setitem__(self, idx, value):
        blk = self._block(idx)
        if blk is None:
            raise NotImplementedError("writes take unit-stride blocks")
        start, count, drop = blk
        val = np.asarray(value, self.dtype)
        if val.shape != tuple(count):
            val = np.broadcast_to(val, [c for c, d in zip(count, drop) if not d])
        arr = np.ascontiguousarray(val).reshape(count)
        if arr.size:
            self._io(self.L.H5Dwrite, start, count, arr)


class _File:
    def __init__(self, name, mode="r", **_ignored):
        self.L = L = _lib()
        name = os.fspath(name).encode()
        if mode == "w":
            self.id = L.H5Fcreate(name, 2, 0, 0)                              # H5F_ACC_TRUNC
        elif mode in ("a", "r+"):
            self.id = L.H5Fopen(name, 1, 0) if os.path.exists(name) else L.H5Fcreate(name, 2, 0, 0)
        elif mode == "r":
            if not os.path.exists(name):
                raise FileNotFoundError(f"no such file: {name.decode()!r}")
            self.id = L.H5Fopen(name, 0, 0)
        else:
            raise ValueError(f"mode {mode!r}")
        if self.id < 0:
            raise OSError(f"cannot open {name.decode()!r} (mode {mode})")
        self.attrs = _Attrs(L, self.id)
        self._open = []

    def create_dataset(self, name, shape=None, dtype=None, data=None, chunks=None):
        L = self.L
        if data is not None:
            data = np.ascontiguousarray(data, dtype)
            shape = data.shape
        dt = np.dtype(dtype if dtype is not None else (data.dtype if data is not None else "f8"))
        if isinstance(shape, (int, np.integer)):
            shape = (int(shape),)
        space = L.check(L.H5Screate_simple(len(shape), _dims(shape), None), "H5Screate_simple")
        dcpl = 0
        if chunks is not None and all(c > 0 for c in chunks) and len(shape):
            dcpl = L.check(L.H5Pcreate(L.dcpl_class), "H5Pcreate")
            L.check(L.H5Pset_chunk(dcpl, len(shape), _dims(chunks)), "H5Pset_chunk")
        did = L.H5Dcreate2(self.id, name.encode(), L.types[dt], space, 0, dcpl, 0)
        if dcpl:
            L.H5Pclose(dcpl)
        L.H5Sclose(space)
        L.check(did, f"H5Dcreate2({name})")
        if data is not None and data.size:
            L.check(L.H5Dwrite(did, L.types[dt], 0, 0, 0, data.ctypes.data_as(C.c_void_p)), "H5Dwrite")
        self._open.append(did)
        return _Dataset(L, did)

    def __contains__(self, name):
        return self.L.H5Lexists(self.id, name.encode(), 0) > 0

    def __getitem__(self, name):
        if name not in self:
            raise KeyError(name)
        did = self.L.check(self.L.H5Dopen2(self.id, name.encode(), 0), "H5Dopen2")
        self._open.append(did)
        return _Dataset(self.L, did)

    def keys(self):
        return self.L.names(self.L.H5Literate, self.id)

    def __iter__(self):
        return iter(self.keys())

    def close(self):
        if self.id >= 0:
            for did in self._open:
                self.L.H5Dclose(did)
            self._open = []
            self.L.check(self.L.H5Fclose(self.id), "H5Fclose")
            self.id = -1

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def File(name, mode="r", **kw):
    """``h5py.File`` when h5py is there, the ctypes binding otherwise"""
    if _h5py is not None:
        return _h5py.File(name, mode, **kw)
    return _File(name, mode, **kw)
