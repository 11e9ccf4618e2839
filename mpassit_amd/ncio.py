"""ctypes front-end of the host-side NetCDF I/O (include/mpassit_ncio.h: hostio/ncclassic.c for the classic formats, hostio/nc4hdf5.h for
NetCDF-4 through libhdf5 where the build found it).

Numpy in, numpy out; shapes follow the file (slowest dimension first, i.e. a Fortran `var(nVertLevels, nCells, Time)`
is `[Time][nCells][nVertLevels]` here).  Used by io_nc.py for the MPAS grid / diag / history files the reference reads
with nf90_get_var (model_grid.F90:287-417, input_data.F90:145-812) and for the WRF-style output it writes
(write_data.F90:173-1498)."""
import ctypes as C
import os

import numpy as np

BYTE, CHAR, SHORT, INT, FLOAT, DOUBLE, UBYTE, USHORT, UINT, INT64, UINT64 = range(1, 12)
GLOBAL = -1
_NP = {BYTE: np.int8, CHAR: "S1", SHORT: np.int16, INT: np.int32, FLOAT: np.float32, DOUBLE: np.float64, UBYTE: np.uint8,
       USHORT: np.uint16, UINT: np.uint32, INT64: np.int64, UINT64: np.uint64}
_FROM_NP = {np.dtype(v): k for k, v in _NP.items() if k != CHAR}

_lib = None


class NcioError(RuntimeError):
    def __init__(self, rc, msg):
        super().__init__("ncio rc=%d: %s" % (rc, msg))
        self.rc = rc


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostio", "libmpassit_ncio.so")
        if not os.path.exists(path):
            raise ImportError("libmpassit_ncio.so not built: run `python -m mpassit_amd.build`")
        L = C.CDLL(path)
        L.ncio_strerror.restype = C.c_char_p
        L.ncio_numrecs.restype = C.c_int64
        _lib = L
    return _lib


def has_netcdf4():
    """True when this build of the library has the HDF5 backend: NetCDF-4 files are read (Reader) and written (Writer(path, format=4))."""
    return bool(lib().ncio_has_netcdf4())


def _check(rc):
    if rc < 0:
        raise NcioError(rc, lib().ncio_strerror().decode("utf-8", "replace"))
    return rc


class Reader:
    def __init__(self, path):
        self._h = C.c_void_p()
        self.path = str(path)
        _check(lib().ncio_open(str(path).encode(), C.byref(self._h)))
        self.format = lib().ncio_format(self._h)
        self.numrecs = lib().ncio_numrecs(self._h)
        self.dims, self.unlimited = {}, None
        for d in range(lib().ncio_ndims(self._h)):
            name = C.create_string_buffer(256)
            n, unl = C.c_int64(), C.c_int()
            _check(lib().ncio_inq_dim_by_id(self._h, d, name, 256, C.byref(n), C.byref(unl)))
            self.dims[name.value.decode("utf-8", "replace")] = n.value
            if unl.value:
                self.unlimited = name.value.decode("utf-8", "replace")
        self.vars = {}
        dim_names = list(self.dims)
        for v in range(lib().ncio_nvars(self._h)):
            name = C.create_string_buffer(256)
            t, nd, rec = C.c_int(), C.c_int(), C.c_int()
            shape, ids = (C.c_int64 * 8)(), (C.c_int * 8)()
            _check(lib().ncio_inq_var(self._h, v, name, 256, C.byref(t), C.byref(nd), shape, ids, C.byref(rec)))
            self.vars[name.value.decode("utf-8", "replace")] = dict(id=v, type=t.value, shape=tuple(shape[:nd.value]),
                                                  dims=tuple(dim_names[i] for i in ids[:nd.value]), record=bool(rec.value))

    def get(self, name, rec=None, dtype=None):
        """Whole variable (record variables: all records stacked, or one with rec=k) as `dtype` (default: the file's type)."""
        v = self.vars[name]
        ftype = v["type"]
        if dtype is None:
            mem = ftype
        else:
            mem = _FROM_NP[np.dtype(dtype)]
        shape = v["shape"][1:] if v["record"] else v["shape"]
        recs = [rec] if rec is not None else (range(v["shape"][0]) if v["record"] else [0])
        out = []
        for r in recs:
            a = np.empty(shape, _NP[mem])
            _check(lib().ncio_get_var(self._h, v["id"], C.c_int64(r), mem, a.ctypes.data_as(C.c_void_p)))
            out.append(a)
        if v["record"] and rec is None:
            return np.stack(out) if out else np.empty((0,) + shape, _NP[mem])
        return out[0]

    def raw(self, name, rec=0):
        """Read-only memory map of the bytes of a variable (of record `rec`) exactly as stored: big-endian elements of
        the variable's type (ncio_var_extent).  Meant for device-side ingest: upload untouched, swap on the GPU."""
        off, nb = self.extent(name, rec)
        return np.memmap(self.path, dtype=np.uint8, mode="r", offset=off, shape=(nb,))

    def extent(self, name, rec=0):
        """(offset, nbytes) of the variable's (record's) bytes in the file (ncio_var_extent)."""
        off, nb = C.c_int64(), C.c_int64()
        _check(lib().ncio_var_extent(self._h, self.vars[name]["id"], C.c_int64(rec), C.byref(off), C.byref(nb)))
        return off.value, nb.value

    def att(self, name, var=None):
        vid = GLOBAL if var is None else self.vars[var]["id"]
        buf = C.create_string_buffer(4096)
        if lib().ncio_get_att_text(self._h, vid, name.encode(), buf, 4096) == 0:
            return buf.value.decode("utf-8", "replace")
        n = C.c_int()
        vals = (C.c_double * 64)()
        _check(lib().ncio_get_att_double(self._h, vid, name.encode(), vals, 64, C.byref(n)))
        return np.array(vals[:min(n.value, 64)])

    def atts(self, var=None):
        """{name: value} of every attribute of a variable (None: of the file), in the file's order."""
        vid = GLOBAL if var is None else self.vars[var]["id"]
        out = {}
        for i in range(_check(lib().ncio_natts(self._h, vid))):
            name = C.create_string_buffer(256)
            _check(lib().ncio_inq_att(self._h, vid, i, name, 256, None, None))
            out[name.value.decode("utf-8", "replace")] = self.att(name.value.decode("utf-8", "replace"), var)
        return out

    def close(self):
        if self._h:
            lib().ncio_close(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def reserve_start(path, nbytes):
    """Start allocating `nbytes` (an upper bound) for an output file on a helper thread; the next Writer(path) keeps that
    file instead of truncating it and close() trims it (ncio_reserve_start)."""
    _check(lib().ncio_reserve_start(str(path).encode(), C.c_int64(int(nbytes))))


class Writer:
    """nf90_create ... nf90_close.  Define dimensions / variables / attributes, then put()."""

    def __init__(self, path, format=5):
        self._h = C.c_void_p()
        self.path = str(path)
        self.format = int(format)              # 1, 2, 5: classic; 4: NetCDF-4 (needs the HDF5 backend, has_netcdf4())
        _check(lib().ncio_create(str(path).encode(), format, C.byref(self._h)))
        self._dims, self._dimlen, self._vars, self._defining = {}, {}, {}, True

    def def_dim(self, name, length):
        """length None / 0 = unlimited."""
        d = C.c_int()
        _check(lib().ncio_def_dim(self._h, name.encode(), C.c_int64(length or 0), C.byref(d)))
        self._dims[name] = d.value
        self._dimlen[name] = int(length or 0)
        return d.value

    def def_var(self, name, nctype, dims, **atts):
        ids = (C.c_int * max(len(dims), 1))(*[self._dims[d] for d in dims])
        v = C.c_int()
        _check(lib().ncio_def_var(self._h, name.encode(), nctype, len(dims), ids, C.byref(v)))
        count = int(np.prod([self._dimlen[d] for d in dims if self._dimlen[d] > 0], dtype=np.int64))
        self._vars[name] = (v.value, nctype, count)
        for k, val in atts.items():
            self.put_att(k, val, var=name)
        return v.value

    def put_att(self, name, value, var=None):
        vid = GLOBAL if var is None else self._vars[var][0]
        L = lib()
        if isinstance(value, str):
            return _check(L.ncio_put_att_text(self._h, vid, name.encode(), value.encode()))
        a = np.atleast_1d(np.asarray(value))
        if a.dtype.kind in "iu":
            a = np.ascontiguousarray(a, np.int32)
            return _check(L.ncio_put_att_int(self._h, vid, name.encode(), a.ctypes.data_as(C.c_void_p), a.size))
        if a.dtype == np.float32:
            return _check(L.ncio_put_att_float(self._h, vid, name.encode(), a.ctypes.data_as(C.c_void_p), a.size))
        a = np.ascontiguousarray(a, np.float64)
        return _check(L.ncio_put_att_double(self._h, vid, name.encode(), a.ctypes.data_as(C.c_void_p), a.size))

    def enddef(self):
        if self._defining:
            _check(lib().ncio_enddef(self._h))
            self._defining = False

    def put(self, name, array, rec=0):
        """Whole variable, or record `rec` of a record variable; the array's dtype is converted to the variable's type."""
        self.enddef()
        vid, nctype, count = self._vars[name]
        if nctype == CHAR:
            a = np.frombuffer(array if isinstance(array, bytes) else np.asarray(array).tobytes(), np.uint8)
            mem = CHAR
        else:
            a = np.ascontiguousarray(array)
            if a.dtype not in _FROM_NP:
                a = a.astype(np.float64)
            mem = _FROM_NP[a.dtype]
        if a.size != count:
            raise ValueError("%s: %d elements given, the variable (record) holds %d" % (name, a.size, count))
        _check(lib().ncio_put_var(self._h, vid, C.c_int64(rec), mem, a.ctypes.data_as(C.c_void_p)))

    def raw(self, name, rec=0):
        """Writable memory map of the bytes of a variable (of record `rec`): the caller stores big-endian elements of the
        variable's type itself (device-side egress: swap on the GPU, copy straight into the file mapping)."""
        off, nb = self.extent(name, rec)
        return np.memmap(self.path, dtype=np.uint8, mode="r+", offset=off, shape=(nb,))

    def extent(self, name, rec=0):
        """(offset, nbytes) of the variable's (record's) bytes; the file is extended so that the range exists."""
        self.enddef()
        off, nb = C.c_int64(), C.c_int64()
        _check(lib().ncio_var_extent(self._h, self._vars[name][0], C.c_int64(rec), C.byref(off), C.byref(nb)))
        return off.value, nb.value

    def close(self):
        if self._h:
            h, self._h = self._h, None
            _check(lib().ncio_close(h))

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
