"""Independent reader of a NetCDF CLASSIC header (CDF-1 / CDF-2 / CDF-5), written from the format grammar for the tests:
dimensions, global attributes and variables with their attributes IN FILE ORDER and with their stored types -- what the
header-parity tests compare with the list transcribed from the reference's writer (write_data.F90:177-560).
Not part of the product (the product's reader is hostio/ncclassic.c)."""
import struct

import numpy as np

_TYPES = {1: ("byte", "i1"), 2: ("char", "S1"), 3: ("short", ">i2"), 4: ("int", ">i4"), 5: ("float", ">f4"), 6: ("double", ">f8"),
          7: ("ubyte", "u1"), 8: ("ushort", ">u2"), 9: ("uint", ">u4"), 10: ("int64", ">i8"), 11: ("uint64", ">u8")}


class _Cur:
    def __init__(self, b, wide):
        self.b, self.p, self.wide = b, 0, wide

    def u32(self):
        v = struct.unpack_from(">I", self.b, self.p)[0]
        self.p += 4
        return v

    def nn(self):   # NON_NEG: 4 bytes in CDF-1/2, 8 in CDF-5
        if self.wide:
            v = struct.unpack_from(">Q", self.b, self.p)[0]
            self.p += 8
            return v
        return self.u32()

    def name(self):
        n = self.nn()
        s = self.b[self.p:self.p + n].decode()
        self.p += (n + 3) // 4 * 4
        return s

    def atts(self):
        tag, n = self.u32(), self.nn()
        assert tag in (0, 0x0C), "attribute list tag %r" % tag
        out = []
        for _ in range(n):
            nm = self.name()
            t = self.u32()
            cnt = self.nn()
            tn, dt = _TYPES[t]
            nb = cnt * np.dtype(dt).itemsize
            raw = self.b[self.p:self.p + nb]
            self.p += (nb + 3) // 4 * 4
            val = raw.decode() if tn == "char" else np.frombuffer(raw, dt).astype(dt.lstrip(">")).tolist()
            out.append((nm, tn, val))
        return out


def parse(path, max_header=1 << 22):
    with open(path, "rb") as f:
        b = f.read(max_header)
    assert b[:3] == b"CDF", "not a classic NetCDF file"
    version = b[3]
    c = _Cur(b, version == 5)
    c.p = 4
    numrecs = c.nn()
    tag, n = c.u32(), c.nn()
    assert tag in (0, 0x0A)
    dims = []
    for _ in range(n):
        nm = c.name()
        dims.append((nm, c.nn()))
    gatts = c.atts()
    tag, n = c.u32(), c.nn()
    assert tag in (0, 0x0B)
    vars_ = []
    for _ in range(n):
        nm = c.name()
        nd = c.nn()
        dimids = [c.nn() for _ in range(nd)]
        va = c.atts()
        t = c.u32()
        c.nn()                                                   # vsize
        c.p += 4 if version == 1 else 8                          # begin
        vars_.append(dict(name=nm, dims=[dims[i][0] for i in dimids], atts=va, type=_TYPES[t][0]))
    return dict(version=version, numrecs=numrecs, dims=dims, gatts=gatts, vars=vars_)
