"""NetCDF-4 (HDF5 container) through the ncio interface (hostio/nc4hdf5.h on libhdf5) against an INDEPENDENT implementation both ways:
h5py of the image's second Python (/opt/conda/bin/python3.9, libhdf5 1.10.6), run as a child process.  Files written here must carry
what the NetCDF-4 format appendix prescribes (dimension scales with the netCDF NAME text or the coordinate variable, scales attached per
axis, _Netcdf4Dimid, unlimited = chunked with unlimited maxshape, fixed-length scalar text attributes, 1-byte-string char variables) and
the data; files h5py writes in that convention (chunked, deflated + shuffled, big-endian, variable-length string attributes) must read
back here.  libnetcdf itself is absent from the image: this pins the HDF5 side of the format, not libnetcdf's acceptance."""
import json
import os
import subprocess

import numpy as np
import pytest

PY39 = "/opt/conda/bin/python3.9"


@pytest.fixture(scope="module")
def ncio():
    from mpassit_amd import build, ncio
    build.build_ncio()
    if not ncio.has_netcdf4():
        pytest.skip("libmpassit_ncio was built without HDF5")
    return ncio


def _h5py(code, *args):
    if not os.path.exists(PY39):
        pytest.skip("no second Python with h5py in this image")
    r = subprocess.run([PY39, "-c", code, *map(str, args)], capture_output=True, text=True, timeout=300)
    if r.returncode != 0 and "No module named" in r.stderr:
        pytest.skip("h5py not importable: " + r.stderr.strip().splitlines()[-1])
    assert r.returncode == 0, r.stderr
    return r.stdout


DUMP = r"""
import sys, json, h5py, numpy as np
out = {"datasets": {}, "order": [], "gatts": {}}
def shp(s): return None if s is None else list(s)
def att(v):
    if isinstance(v, h5py.Empty): return ""
    if isinstance(v, bytes): return v.decode()
    if isinstance(v, np.ndarray): return v.tolist() if v.dtype.kind != "O" else None
    if isinstance(v, np.generic): return v.item()
    return v if isinstance(v, (str, int, float)) else None
with h5py.File(sys.argv[1], "r") as f:
    for k, v in f.attrs.items(): out["gatts"][k] = {"value": att(v), "dtype": str(f.attrs.get_id(k).dtype), "shape": shp(f.attrs.get_id(k).shape)}
    names = []
    f.id.links.iterate(lambda n: names.append(n.decode()), idx_type=h5py.h5.INDEX_CRT_ORDER, order=h5py.h5.ITER_INC)   # creation order (tracked)
    for name in names:
        d = f[name]
        out["order"].append(name)
        e = {"shape": list(d.shape), "maxshape": [None if m is None else int(m) for m in d.maxshape], "dtype": str(d.dtype), "chunks": d.chunks and list(d.chunks),
             "is_scale": bool(h5py.h5ds.is_scale(d.id)), "scales": [[s.name for s in d.dims[i].values()] for i in range(d.ndim)], "atts": {}}
        for k, v in d.attrs.items():
            if k in ("DIMENSION_LIST", "REFERENCE_LIST"): continue
            e["atts"][k] = {"value": att(v), "dtype": str(d.attrs.get_id(k).dtype), "shape": shp(d.attrs.get_id(k).shape)}
        if d.dtype.kind == "S": e["data"] = d[...].tobytes().decode("latin1")
        elif d.size <= 4096: e["data"] = d[...].astype(float).ravel().tolist()
        e["sum"] = float(d[...].astype(float).sum()) if d.dtype.kind != "S" else 0.0
        out["datasets"][name] = e
print(json.dumps(out))
"""


def test_files_written_here_are_netcdf4_to_h5py(ncio, tmp_path):
    rng = np.random.default_rng(3)
    path = tmp_path / "out4.nc"
    T = rng.normal(290, 10, (2, 3, 5, 7)).astype(np.float32)
    xlat = rng.uniform(20, 50, (5, 7)).astype(np.float32)
    soil = np.array([0.05, 0.25, 0.7, 1.5])
    w = ncio.Writer(path, format=4)
    for n, L in (("Time", None), ("DateStrLen", 19), ("west_east", 7), ("south_north", 5), ("bottom_top", 3), ("soil_layers", 4)):
        w.def_dim(n, L)
    w.put_att("TITLE", "OUTPUT FROM MPASSIT")
    w.put_att("MAP_PROJ", np.array([1], np.int32))
    w.put_att("TRUELAT1", np.array([38.5], np.float32))
    w.put_att("corners", np.array([1.5, 2.5, 3.5]))
    w.put_att("EMPTY", "")
    w.def_var("Times", ncio.CHAR, ("Time", "DateStrLen"))
    w.def_var("T", ncio.FLOAT, ("Time", "bottom_top", "south_north", "west_east"), units="K", FieldType=np.array([104], np.int32), MemoryOrder="XYZ")
    w.def_var("XLAT", ncio.FLOAT, ("south_north", "west_east"), units="degree_north")
    w.def_var("soil_layers", ncio.DOUBLE, ("soil_layers",), units="m")              # a coordinate variable: it IS its dimension's scale
    w.def_var("ITIMESTEP", ncio.INT, ("Time",))
    w.def_var("P_TOP", ncio.DOUBLE, ())
    for rec, stamp in enumerate((b"2024-08-07_00:00:00", b"2024-08-07_01:00:00")):
        w.put("Times", stamp, rec=rec)
        w.put("T", T[rec], rec=rec)
        w.put("ITIMESTEP", np.array([rec * 180], np.int32), rec=rec)
    w.put("XLAT", xlat)
    w.put("soil_layers", soil)
    w.put("P_TOP", np.array([5000.0]))
    w.close()
    assert open(path, "rb").read(8) == b"\x89HDF\r\n\x1a\n"
    d = json.loads(_h5py(DUMP, path))
    ds = d["datasets"]
    # definition order survives (dimensions first, then the variables as defined)
    assert d["order"] == ["Time", "DateStrLen", "west_east", "south_north", "bottom_top", "Times", "T", "XLAT", "soil_layers", "ITIMESTEP", "P_TOP"]
    # dimensions without a variable: scales carrying libnetcdf's NAME text and their number; the unlimited one extensible
    for k, (name, n) in enumerate((("Time", 2), ("DateStrLen", 19), ("west_east", 7), ("south_north", 5), ("bottom_top", 3))):
        e = ds[name]
        assert e["is_scale"] and e["shape"] == [n] and e["atts"]["_Netcdf4Dimid"]["value"] == k
        want = "This is a netCDF dimension but not a netCDF variable.%10d" % (0 if name == "Time" else n)
        assert e["atts"]["NAME"]["value"] == want and e["atts"]["CLASS"]["value"] == "DIMENSION_SCALE"
        assert e["maxshape"] == ([None] if name == "Time" else [n])
    assert ds["soil_layers"]["is_scale"] and ds["soil_layers"]["atts"]["_Netcdf4Dimid"]["value"] == 5 and ds["soil_layers"]["atts"]["NAME"]["value"] == "soil_layers"
    # every axis of every variable points at its dimension
    assert ds["T"]["scales"] == [["/Time"], ["/bottom_top"], ["/south_north"], ["/west_east"]]
    assert ds["Times"]["scales"] == [["/Time"], ["/DateStrLen"]] and ds["XLAT"]["scales"] == [["/south_north"], ["/west_east"]]
    assert ds["ITIMESTEP"]["scales"] == [["/Time"]] and ds["P_TOP"]["shape"] == []
    # variables along the unlimited dimension: chunked, one record per chunk, extensible; the others contiguous
    assert ds["T"]["shape"] == [2, 3, 5, 7] and ds["T"]["maxshape"] == [None, 3, 5, 7] and ds["T"]["chunks"] == [1, 3, 5, 7]
    assert ds["XLAT"]["chunks"] is None and ds["T"]["dtype"] == "float32" and ds["ITIMESTEP"]["dtype"] == "int32" and ds["Times"]["dtype"] == "|S1"
    # the data
    assert np.array_equal(np.array(ds["T"]["data"], np.float32).reshape(T.shape), T)
    assert np.array_equal(np.array(ds["XLAT"]["data"], np.float32).reshape(xlat.shape), xlat)
    assert ds["soil_layers"]["data"] == soil.tolist() and ds["ITIMESTEP"]["data"] == [0.0, 180.0] and ds["P_TOP"]["data"] == [5000.0]
    assert ds["Times"]["data"] == "2024-08-07_00:00:002024-08-07_01:00:00"
    # attributes: text = fixed-length scalar string, numbers = 1-D arrays of the given type
    a = ds["T"]["atts"]
    assert a["units"] == {"value": "K", "dtype": "|S1", "shape": []} and a["FieldType"] == {"value": [104], "dtype": "int32", "shape": [1]}
    g = d["gatts"]
    assert g["TITLE"]["value"] == "OUTPUT FROM MPASSIT" and g["TITLE"]["dtype"] == "|S19" and g["MAP_PROJ"]["value"] == [1]
    assert g["TRUELAT1"] == {"value": [38.5], "dtype": "float32", "shape": [1]} and g["corners"] == {"value": [1.5, 2.5, 3.5], "dtype": "float64", "shape": [3]}
    h5dump = "/opt/conda/bin/h5dump"
    if os.path.exists(h5dump):                                          # the HDF5 tools read the header without complaint
        r = subprocess.run([h5dump, "-H", str(path)], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0 and "DIMENSION_LIST" in r.stdout and "error" not in r.stderr.lower(), r.stderr
    # ... and the file reads back here
    with ncio.Reader(path) as r:
        assert r.format == 4 and r.unlimited == "Time" and r.numrecs == 2 and list(r.dims) == ["Time", "DateStrLen", "west_east", "south_north", "bottom_top", "soil_layers"]
        assert r.vars["T"]["dims"] == ("Time", "bottom_top", "south_north", "west_east") and r.vars["T"]["record"] and not r.vars["XLAT"]["record"]
        assert np.array_equal(r.get("T"), T) and np.array_equal(r.get("T", rec=1, dtype=np.float64), T[1].astype(np.float64))
        assert r.get("Times", rec=1).tobytes() == b"2024-08-07_01:00:00" and np.array_equal(r.get("soil_layers"), soil)
        assert r.att("units", "T") == "K" and r.att("TITLE") == "OUTPUT FROM MPASSIT" and r.att("EMPTY") == "" and r.att("corners").tolist() == [1.5, 2.5, 3.5]
        assert list(r.vars) == ["Times", "T", "XLAT", "soil_layers", "ITIMESTEP", "P_TOP"]
        with pytest.raises(ncio.NcioError, match="no raw byte range"):
            r.extent("T")


MAKE = r"""
import sys, h5py, numpy as np
rng = np.random.default_rng(11)
nC, nV, nZ, nT = 37, 70, 5, 3
with h5py.File(sys.argv[1], "w", track_order=(sys.argv[2] == "tracked")) as f:
    f.attrs["_NCProperties"] = np.bytes_("version=2,netcdf=4.9.2,hdf5=1.10.6")
    f.attrs["model_name"] = np.bytes_("mpas")                       # fixed-length string
    f.attrs["core_name"] = "atmosphere"                             # variable-length string (what netCDF4-python writes for str)
    f.attrs["sphere_radius"] = np.array([6371229.0])
    f.attrs["config_dt"] = np.array([20], np.int32)
    dims = {}
    def dim(name, n, unlimited=False, k=[0]):
        d = f.create_dataset(name, (n,), maxshape=(None,) if unlimited else (n,), dtype=">f4", chunks=(4,) if unlimited else None)
        d.make_scale("This is a netCDF dimension but not a netCDF variable.%10d" % n)
        d.attrs["_Netcdf4Dimid"] = np.int32(k[0]); k[0] += 1
        dims[name] = d
    dim("Time", nT, True); dim("nCells", nC); dim("nVertLevels", nZ); dim("StrLen", 64); dim("maxEdges", 6)
    def var(name, data, dnames, **kw):
        v = f.create_dataset(name, data=data, **kw)
        for i, dn in enumerate(dnames): v.dims[i].attach_scale(dims[dn])
        return v
    lat = var("latCell", rng.uniform(-1.5, 1.5, nC), ["nCells"]); lat.attrs["units"] = np.bytes_("rad"); lat.attrs["long_name"] = "Latitude of cells"
    var("verticesOnCell", rng.integers(0, nV + 1, (nC, 6)).astype(np.int32), ["nCells", "maxEdges"])
    th = var("theta", rng.normal(300, 20, (nT, nC, nZ)).astype(np.float32), ["Time", "nCells", "nVertLevels"], maxshape=(None, nC, nZ), chunks=(1, 16, nZ),
             compression="gzip", shuffle=True)
    th.attrs["units"] = np.bytes_("K"); th.attrs["valid_range"] = np.array([100.0, 500.0], np.float32)
    var("rho_be", rng.uniform(0.1, 1.3, (nT, nC, nZ)).astype(">f8"), ["Time", "nCells", "nVertLevels"], maxshape=(None, nC, nZ), chunks=(2, nC, 2))
    var("flag", rng.integers(-100, 100, nC).astype(np.int16), ["nCells"])
    xt = np.zeros((nT, 64), "S1")
    for t in range(nT): xt[t, :19] = list("2024-08-07_0%d:00:00" % t)
    var("xtime", xt, ["Time", "StrLen"], maxshape=(None, 64), chunks=(1, 64))
    z = f.create_dataset("nVertLevelsP1", data=np.arange(6.0))      # a coordinate variable: its own scale
    z.make_scale("nVertLevelsP1"); z.attrs["_Netcdf4Dimid"] = np.int32(5)
    var("zgrid", rng.uniform(0, 3e4, (nC, 6)), ["nCells", "nVertLevelsP1"]) if False else None
    zg = f.create_dataset("zgrid", data=rng.uniform(0, 3e4, (nC, 6))); zg.dims[0].attach_scale(dims["nCells"]); zg.dims[1].attach_scale(z)
    f.create_dataset("names", data=np.array([b"abc", b"defgh"], "S5"))   # an array of longer strings: not a classic-model variable
np.savez(sys.argv[3], **{k: (f2 := None) or v for k, v in {}.items()})
with h5py.File(sys.argv[1], "r") as f:
    np.savez(sys.argv[3], **{k: f[k][...] for k in ("latCell", "verticesOnCell", "theta", "rho_be", "flag", "zgrid", "nVertLevelsP1")}, xtime=f["xtime"][...].view(np.uint8))
"""


@pytest.mark.parametrize("order", ["tracked", "by_name"])
def test_netcdf4_files_written_by_h5py_are_read(ncio, tmp_path, order):
    path, ref = tmp_path / "mpas4.nc", tmp_path / "ref.npz"
    _h5py(MAKE, path, order, ref)
    want = np.load(ref)
    with ncio.Reader(path) as r:
        assert r.format == 4 and r.unlimited == "Time" and r.numrecs == 3
        assert r.dims == {"Time": 3, "nCells": 37, "nVertLevels": 5, "StrLen": 64, "maxEdges": 6, "nVertLevelsP1": 6}
        assert list(r.dims) == ["Time", "nCells", "nVertLevels", "StrLen", "maxEdges", "nVertLevelsP1"]        # _Netcdf4Dimid order, whatever the link order
        assert "names" not in r.vars and "Time" not in r.vars and "nVertLevelsP1" in r.vars                   # dimension-only datasets are no variables
        v = r.vars
        assert v["theta"]["dims"] == ("Time", "nCells", "nVertLevels") and v["theta"]["record"] and v["theta"]["type"] == ncio.FLOAT
        assert v["rho_be"]["type"] == ncio.DOUBLE and v["flag"]["type"] == ncio.SHORT and v["verticesOnCell"]["dims"] == ("nCells", "maxEdges")
        assert v["zgrid"]["dims"] == ("nCells", "nVertLevelsP1") and v["nVertLevelsP1"]["dims"] == ("nVertLevelsP1",) and v["xtime"]["type"] == ncio.CHAR
        for k in ("latCell", "verticesOnCell", "theta", "rho_be", "flag", "zgrid", "nVertLevelsP1"):            # chunked + deflated + shuffled, big-endian, int16 ...
            assert np.array_equal(r.get(k), want[k]), k
        assert np.array_equal(r.get("theta", rec=2, dtype=np.float64), want["theta"][2].astype(np.float64))    # conversion by the library
        assert np.array_equal(r.get("verticesOnCell", dtype=np.int32), want["verticesOnCell"])
        assert r.get("xtime", rec=1).tobytes()[:19] == b"2024-08-07_01:00:00" and np.array_equal(r.get("xtime").view(np.uint8), want["xtime"])
        assert r.att("model_name") == "mpas" and r.att("core_name") == "atmosphere" and r.att("sphere_radius")[0] == 6371229.0 and r.att("config_dt")[0] == 20
        assert r.att("units", "theta") == "K" and r.att("valid_range", "theta").tolist() == [100.0, 500.0] and r.att("long_name", "latCell") == "Latitude of cells"
        with pytest.raises(ncio.NcioError):
            r.att("_NCProperties")                                     # the container's own attributes are not the file's
        with pytest.raises(ncio.NcioError, match="out of range"):
            r.get("theta", rec=3)


PLAIN = r"""
import sys, h5py, numpy as np
with h5py.File(sys.argv[1], "w") as f:
    f.create_dataset("a", data=np.arange(12.0).reshape(3, 4)); f.create_dataset("b", data=np.arange(4, dtype=np.int32)); f.create_dataset("c", data=np.arange(20.0).reshape(4, 5))
    g = f.create_group("sub"); g.create_dataset("hidden", data=np.zeros(3))
"""


def test_plain_hdf5_without_scales_gets_anonymous_dimensions(ncio, tmp_path):
    path = tmp_path / "plain.h5"
    _h5py(PLAIN, path)
    with ncio.Reader(path) as r:
        assert r.dims == {"phony_dim_0": 3, "phony_dim_1": 4, "phony_dim_2": 5} and r.unlimited is None       # one per distinct length, as libnetcdf does
        assert r.vars["a"]["dims"] == ("phony_dim_0", "phony_dim_1") and r.vars["b"]["dims"] == ("phony_dim_1",) and r.vars["c"]["dims"] == ("phony_dim_1", "phony_dim_2")
        assert "hidden" not in r.vars and np.array_equal(r.get("a"), np.arange(12.0).reshape(3, 4)) and np.array_equal(r.get("b"), np.arange(4))


def test_every_type_round_trips_and_misuse_is_refused(ncio, tmp_path):
    rng = np.random.default_rng(5)
    path = tmp_path / "types.nc"
    data = {ncio.BYTE: rng.integers(-128, 127, 9).astype(np.int8), ncio.UBYTE: rng.integers(0, 255, 9).astype(np.uint8),
            ncio.SHORT: rng.integers(-3e4, 3e4, 9).astype(np.int16), ncio.USHORT: rng.integers(0, 6e4, 9).astype(np.uint16),
            ncio.INT: rng.integers(-2e9, 2e9, 9).astype(np.int32), ncio.UINT: rng.integers(0, 4e9, 9).astype(np.uint32),
            ncio.INT64: rng.integers(-9e18, 9e18, 9), ncio.UINT64: rng.integers(0, 2 ** 63, 9).astype(np.uint64) * 2,
            ncio.FLOAT: rng.normal(size=9).astype(np.float32), ncio.DOUBLE: rng.normal(size=9)}
    w = ncio.Writer(path, format=4)
    w.def_dim("n", 9)
    for t in data:
        w.def_var("v%d" % t, t, ("n",))
    with pytest.raises(ncio.NcioError, match="named like a dimension"):
        w2 = ncio.Writer(tmp_path / "bad.nc", format=4)
        w2.def_dim("x", 3)
        w2.def_dim("y", 2)
        w2.def_var("x", ncio.FLOAT, ("y", "x"))
        w2.enddef()
    for t, a in data.items():
        w.put("v%d" % t, a)
    import ctypes as C
    txt = np.frombuffer(b"abcdefghi", np.uint8).copy()                      # text converts to text only (asked of the C interface directly)
    assert ncio.lib().ncio_put_var(w._h, w._vars["v%d" % ncio.FLOAT][0], C.c_int64(0), ncio.CHAR, txt.ctypes.data_as(C.c_void_p)) == -5
    w.close()
    with ncio.Reader(path) as r:
        for t, a in data.items():
            got = r.get("v%d" % t)
            assert got.dtype == a.dtype and np.array_equal(got, a), t
        assert np.allclose(r.get("v%d" % ncio.INT, dtype=np.float64), data[ncio.INT].astype(np.float64))
    # a large record variable is cut along its slowest axes into chunks of about 8 MiB: levels of a [z][y][x] field, runs of rows of a [cell][level] one
    big = tmp_path / "big.nc"
    w = ncio.Writer(big, format=4)
    w.def_dim("Time", None)
    w.def_dim("z", 40)
    w.def_dim("y", 700)
    w.def_dim("x", 900)
    w.def_var("P", ncio.FLOAT, ("Time", "z", "y", "x"))
    plane = rng.normal(size=(40, 700, 900)).astype(np.float32)
    w.put("P", plane, rec=0)
    w.close()
    w = ncio.Writer(tmp_path / "big64.nc", format=4)                      # float64 handed to a float32 variable: the threaded conversion pass
    w.def_dim("Time", None)
    w.def_dim("n", plane.size)
    w.def_var("P", ncio.FLOAT, ("Time", "n"))
    w.put("P", plane.astype(np.float64).ravel(), rec=0)
    w.close()
    with ncio.Reader(tmp_path / "big64.nc") as r:
        assert np.array_equal(r.get("P", rec=0).reshape(plane.shape), plane)
    d64 = json.loads(_h5py(DUMP, tmp_path / "big64.nc"))["datasets"]["P"]          # an MPAS-shaped variable: 25.2 M x 1 x 4 B in chunks of 2 M elements
    assert d64["chunks"] == [1, 2097152]
    cells = tmp_path / "cells.nc"
    w = ncio.Writer(cells, format=4)
    w.def_dim("Time", None)
    w.def_dim("nCells", 300000)
    w.def_dim("nVertLevels", 55)
    w.def_var("theta", ncio.FLOAT, ("Time", "nCells", "nVertLevels"))
    th = rng.normal(size=(300000, 55)).astype(np.float32)
    w.put("theta", th, rec=0)
    w.close()
    dth = json.loads(_h5py(DUMP, cells))["datasets"]["theta"]
    assert dth["chunks"] == [1, 38130, 55]                                             # runs of whole rows, 8 MiB each: never one row per chunk
    with ncio.Reader(cells) as r:
        assert np.array_equal(r.get("theta", rec=0), th)
    with ncio.Reader(big) as r:
        assert r.numrecs == 1 and np.array_equal(r.get("P", rec=0), plane) and np.array_equal(r.get("P", rec=0, dtype=np.float64), plane.astype(np.float64))
    d = json.loads(_h5py(DUMP, big))["datasets"]["P"]
    assert d["chunks"] == [1, 3, 700, 900] and abs(d["sum"] - float(plane.astype(float).sum())) < 1e-3


def test_a_build_without_hdf5_says_so(tmp_path):
    """The same sources compiled without the HDF5 backend: NetCDF-4 files are recognised, refused with the way out, nothing else changes."""
    import ctypes as C
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = str(tmp_path / "libncio_plain.so")
    r = subprocess.run(["gcc", "-O1", "-shared", "-fPIC", "-pthread", "-o", so, os.path.join(root, "mpassit_amd", "hostio", "ncclassic.c")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    L = C.CDLL(so)
    L.ncio_strerror.restype = C.c_char_p
    assert L.ncio_has_netcdf4() == 0
    p = tmp_path / "x.nc"
    p.write_bytes(b"\x89HDF\r\n\x1a\n" + bytes(100))
    h = C.c_void_p()
    assert L.ncio_open(str(p).encode(), C.byref(h)) == -3 and b"nccopy -k cdf5" in L.ncio_strerror() and b"MPASSIT_HDF5_ROOT" in L.ncio_strerror()
    assert L.ncio_create(str(tmp_path / "y.nc").encode(), 4, C.byref(h)) == -3
    assert L.ncio_create(str(tmp_path / "y.nc").encode(), 5, C.byref(h)) == 0 and L.ncio_close(h) == 0


def test_two_threads_inside_the_hdf5_backend(ncio, tmp_path):
    """Hosts use two threads on files (io_nc.run_series reads the next file while the current one is written) and libhdf5's usual build
    is not thread-safe: every entry of the backend takes one lock.  Three threads write and read their own NetCDF-4 files over and over
    (ctypes drops the GIL inside the calls): every value comes back."""
    import threading
    errs = []

    def work(k):
        try:
            rng = np.random.default_rng(k)
            for it in range(12):
                a = rng.normal(size=(3, 40, 50)).astype(np.float32)
                p = tmp_path / ("t%d_%d.nc" % (k, it))
                w = ncio.Writer(p, format=4)
                w.def_dim("Time", None)
                w.def_dim("y", 40)
                w.def_dim("x", 50)
                w.def_var("v", ncio.FLOAT, ("Time", "y", "x"), units="K")
                for r in range(3):
                    w.put("v", a[r], rec=r)
                w.close()
                with ncio.Reader(p) as rd:
                    assert rd.numrecs == 3 and np.array_equal(rd.get("v"), a) and rd.att("units", "v") == "K"
        except Exception as e:      # noqa: BLE001
            errs.append((k, repr(e)))

    ts = [threading.Thread(target=work, args=(k,)) for k in range(3)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs


def test_the_hosts_hdf5_error_handler_survives_ncio_calls(tmp_path, ncio):
    """Round-5 advisor: nc4_open_ / nc4_create_ switched libhdf5's automatic error printing off for the WHOLE process.  The handler a host
    installed (another HDF5 user's printing, a harness's hook) is now saved on entry and put back on the way out of every ncio call --
    asked of the very libhdf5 the library is linked against, around a failing open and a successful write."""
    import ctypes
    from mpassit_amd import build
    h5 = build.find_hdf5()
    lib = ctypes.CDLL(os.path.join(h5[1], "libhdf5.so"), mode=ctypes.RTLD_GLOBAL)
    lib.H5open()
    CB = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_int64, ctypes.c_void_p)
    calls = []
    cb = CB(lambda stack, data: calls.append(1) or 0)
    lib.H5Eset_auto2.argtypes = [ctypes.c_int64, CB, ctypes.c_void_p]
    lib.H5Eget_auto2.argtypes = [ctypes.c_int64, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p)]
    tag = ctypes.c_void_p(0x5EED)
    assert lib.H5Eset_auto2(0, cb, tag) >= 0                      # H5E_DEFAULT = 0
    try:
        bad = str(tmp_path / "bad.nc")
        open(bad, "wb").write(b"\x89HDF\r\n\x1a\n" + b"\0" * 64)   # the magic, then nothing libhdf5 can use
        with pytest.raises(Exception):
            ncio.Reader(bad)
        good = str(tmp_path / "good.nc")
        with ncio.Writer(good, format=4) as w:
            w.def_dim("x", 3)
            w.def_var("v", ncio.FLOAT, ("x",))
            w.put("v", np.arange(3, dtype=np.float32))
        fn, data = ctypes.c_void_p(), ctypes.c_void_p()
        assert lib.H5Eget_auto2(0, ctypes.byref(fn), ctypes.byref(data)) >= 0
        assert fn.value == ctypes.cast(cb, ctypes.c_void_p).value and data.value == 0x5EED
        assert not calls                                           # and the library's own failures stayed silent
    finally:
        lib.H5Eset_auto2(0, CB(0), None)
