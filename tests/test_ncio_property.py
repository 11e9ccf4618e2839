"""Randomised round trips of the NetCDF classic reader / writer (hypothesis): arbitrary dimension sets, variable types,
record counts and attribute payloads written with ncio must come back identical through ncio and -- for CDF-1/2 --
through scipy's independent reader."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st
from scipy.io import netcdf_file

TYPES = ["BYTE", "SHORT", "INT", "FLOAT", "DOUBLE"]
NP = {"BYTE": np.int8, "SHORT": np.int16, "INT": np.int32, "FLOAT": np.float32, "DOUBLE": np.float64}


@pytest.fixture(scope="module")
def ncio():
    from mpassit_amd import build, ncio
    build.build_ncio()
    return ncio


var_spec = st.tuples(st.sampled_from(TYPES), st.lists(st.integers(0, 2), min_size=0, max_size=3), st.booleans())


@settings(max_examples=40, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(fmt=st.sampled_from([1, 2, 5]), dims=st.lists(st.integers(1, 7), min_size=3, max_size=3),
       specs=st.lists(var_spec, min_size=1, max_size=6), nrec=st.integers(1, 3), seed=st.integers(0, 2 ** 16))
def test_random_files_round_trip(ncio, tmp_path, fmt, dims, specs, nrec, seed):
    rng = np.random.default_rng(seed)
    path = tmp_path / ("r%d_%d.nc" % (fmt, seed))
    names = ["d0", "d1", "d2"]
    data = {}
    rec_vars = [k for k, (_, _, rec) in enumerate(specs) if rec]
    # a lone record variable of 1- or 2-byte elements with an odd record size is refused by the writer (see ncclassic.c)
    if len(rec_vars) == 1:
        t, dd, _ = specs[rec_vars[0]]
        size = int(np.prod([dims[d] for d in dd], dtype=np.int64)) * np.dtype(NP[t]).itemsize
        if size % 4:
            specs = specs + [("INT", [], True)]
    with ncio.Writer(path, format=fmt) as w:
        w.def_dim("Time", None)
        for n, L in zip(names, dims):
            w.def_dim(n, L)
        w.put_att("title", "prop-%d" % seed)
        w.put_att("vals", rng.normal(size=3))
        for k, (t, dd, rec) in enumerate(specs):
            vdims = (("Time",) if rec else ()) + tuple(names[d] for d in dd)
            w.def_var("v%d" % k, getattr(ncio, t), vdims, units="u%d" % k)
            shape = tuple(dims[d] for d in dd)
            n = nrec if rec else 1
            arr = (rng.normal(0, 50, (n,) + shape)).astype(NP[t])
            data["v%d" % k] = (arr, rec)
        for name, (arr, rec) in data.items():
            for r in range(arr.shape[0]):
                w.put(name, arr[r], rec=r)
    has_rec = any(rec for _, rec in data.values())
    with ncio.Reader(path) as r:
        assert r.format == fmt and r.numrecs == (nrec if has_rec else 0)
        assert r.att("title") == "prop-%d" % seed
        for name, (arr, rec) in data.items():
            got = r.get(name)
            want = arr if rec else arr[0]
            assert got.dtype == want.dtype and np.array_equal(got, want), name
            assert r.att("units", var=name) == "u" + name[1:]
    if fmt != 5:
        f = netcdf_file(str(path), "r", mmap=False)
        for name, (arr, rec) in data.items():
            got = np.asarray(f.variables[name][:] if arr.ndim > 1 or rec else f.variables[name].getValue())
            want = arr if rec else arr[0]
            assert np.array_equal(got.reshape(want.shape), want), name
        f.close()


@pytest.mark.parametrize("fmt", [1, 2, 5])
def test_damaged_headers_fail_cleanly(ncio, tmp_path, fmt):
    """A truncated or corrupted header must end in NcioError (NCIO_EFORMAT / NCIO_EIO), never in a crash, a huge
    allocation or a NULL dereference: every truncation point of the header, and every 4-byte word of it overwritten with
    0x7fffffff / 0xffffffff (attribute counts, name lengths, dimension and variable counts, types).  Runs under
    AddressSanitizer in tools/sanitize_cpu.sh."""
    good = tmp_path / "good.nc"
    with ncio.Writer(str(good), format=fmt) as w:
        w.def_dim("Time", None)
        w.def_dim("x", 5)
        w.put_att("title", "damaged header test")
        w.put_att("numbers", np.arange(4, dtype=np.float64))
        w.def_var("a", ncio.FLOAT, ("Time", "x"), units="m", long_name="a variable", valid=np.array([1, 2], np.int32))
        w.def_var("b", ncio.DOUBLE, ("x",))
        w.put("a", np.arange(5, dtype=np.float32), rec=0)
        w.put("b", np.arange(5, dtype=np.float64))
    blob = good.read_bytes()
    with ncio.Reader(str(good)) as r:
        hdr = min(r.extent("a", 0)[0], r.extent("b")[0])          # the header ends where the first variable's data begin
    bad = tmp_path / "bad.nc"
    opened = 0
    for cut in range(0, hdr):
        bad.write_bytes(blob[:cut])
        with pytest.raises(ncio.NcioError):
            ncio.Reader(str(bad))
    for off in range(4, hdr - 3, 4):
        for word in (b"\x7f\xff\xff\xff", b"\xff\xff\xff\xff"):
            bad.write_bytes(blob[:off] + word + blob[off + 4:])
            try:
                with ncio.Reader(str(bad)) as r:                   # some words are payload (attribute values, offsets): still a valid file
                    opened += 1
                    for name in r.vars:
                        try:
                            r.get(name)
                        except ncio.NcioError:
                            pass
            except ncio.NcioError:
                pass
    assert opened > 0
