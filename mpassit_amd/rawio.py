"""MPGRAW1 raw-binary named-array container: the NetCDF stand-in shared by the Fortran driver and Python
(no NetCDF library exists in the build image; file I/O is outside the hot path).

record = name (32 bytes, space padded) | dtype int32 (0 float64, 1 int32) | ndim int32 |
         dims 3 x int64 in FORTRAN order (fastest first, unused = 1) | data (C order of the numpy array)
"""
import struct

import numpy as np

MAGIC = b"MPGRAW1 "


def write(path, arrays):
    """arrays: dict name -> numpy array (float64 or int32, <= 3 dims)."""
    with open(path, "wb") as f:
        f.write(MAGIC)
        for name, a in arrays.items():
            a = np.ascontiguousarray(a)
            if a.dtype == np.float64:
                dt = 0
            elif a.dtype == np.int32:
                dt = 1
            else:
                raise TypeError("%s: only float64 / int32" % name)
            if a.ndim > 3 or len(name) > 32:
                raise ValueError(name)
            dims = list(a.shape[::-1]) + [1] * (3 - a.ndim)
            f.write(name.encode().ljust(32))
            f.write(struct.pack("<ii3q", dt, a.ndim, *dims))
            f.write(a.tobytes())


def read(path):
    out = {}
    with open(path, "rb") as f:
        if f.read(8) != MAGIC:
            raise ValueError("not an MPGRAW1 file")
        while True:
            hdr = f.read(32 + 4 + 4 + 24)
            if len(hdr) < 64:
                break
            name = hdr[:32].decode().strip()
            dt, ndim, d0, d1, d2 = struct.unpack("<ii3q", hdr[32:])
            dtype = np.float64 if dt == 0 else np.int32
            n = d0 * d1 * d2
            a = np.frombuffer(f.read(n * np.dtype(dtype).itemsize), dtype=dtype)
            out[name] = a.reshape([d2, d1, d0][3 - ndim:])
    return out
