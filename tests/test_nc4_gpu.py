"""NetCDF-4 at either end of the file job (the reference reads whatever nf90_open accepts and writes NF90_NETCDF4, write_data.F90:173):
the same job with NetCDF-4 inputs and / or a NetCDF-4 output gives, variable for variable and attribute for attribute, what the classic
files give -- through the Python host (io_nc.run, host arrays and device-resident fields) and through the Fortran driver
(MPASSIT_OUTPUT_FORMAT=netcdf4).  The output is also read by h5py (tests/test_nc4.py has the format-level checks)."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ncio4():
    from mpassit_amd import build, ncio
    build.build_ncio()
    if not ncio.has_netcdf4():
        pytest.skip("libmpassit_ncio was built without HDF5")
    return ncio


def _same_files(ncio, a, b):
    """every dimension, variable (type, dimensions, values bit for bit) and attribute of file a is in file b and vice versa"""
    with ncio.Reader(a) as ra, ncio.Reader(b) as rb:
        assert ra.dims == rb.dims and list(ra.dims) == list(rb.dims) and ra.unlimited == rb.unlimited and ra.numrecs == rb.numrecs
        assert list(ra.vars) == list(rb.vars)
        for k, va in ra.vars.items():
            vb = rb.vars[k]
            assert va["type"] == vb["type"] and va["dims"] == vb["dims"] and va["record"] == vb["record"], k
            xa, xb = ra.get(k), rb.get(k)
            assert xa.tobytes() == xb.tobytes(), k
        def same(da, db, what):
            assert list(da) == list(db), what
            for k in da:
                assert (da[k] == db[k]) if isinstance(da[k], str) else np.array_equal(da[k], db[k]), (what, k)
        ga = ra.atts()
        assert len(ga) > 30 and "MAP_PROJ" in ga
        same(ga, rb.atts(), "global")
        for k in ra.vars:
            same(ra.atts(k), rb.atts(k), k)


def test_python_host_netcdf4_in_and_out_equals_the_classic_run(gpu_lib, ncio4, tmp_path):
    from mpassit_amd import interp as I, io_nc, synth, target_grid as T, workloads
    m, _, nz, _ = workloads.workload("tiny")
    nsoil = 4
    rng = np.random.default_rng(17)
    f32 = lambda a: np.asarray(a, np.float32)           # noqa: E731
    hist = {"xland": f32(np.floor(rng.uniform(1, 3, m.nCells))), "skintemp": f32(rng.uniform(250, 320, m.nCells)),
            "snow": f32(synth.snow_field(m.latCell, m.lonCell)[0]),
            "zgrid": f32(np.sort(rng.uniform(0, 2.0e4, (m.nCells, nz + 1)), axis=1)), "theta": f32(rng.uniform(280, 500, (m.nCells, nz))),
            "uReconstructZonal": f32(rng.normal(0, 10, (m.nCells, nz))), "uReconstructMeridional": f32(rng.normal(0, 10, (m.nCells, nz))),
            "pressure": f32(-np.sort(-rng.uniform(2.0e3, 1.0e5, (m.nCells, nz)), axis=1)), "rho": f32(rng.uniform(0.1, 1.2, (m.nCells, nz))),
            "tslb": f32(rng.uniform(260, 300, (m.nCells, nsoil)))}
    diag = {"t2m": f32(rng.uniform(250, 310, m.nCells)), "u10": f32(rng.normal(0, 8, m.nCells)), "v10": f32(rng.normal(0, 8, m.nCells)),
            "refl10cm": f32(rng.uniform(-30, 60, (m.nCells, nz)))}
    ter = rng.uniform(0, 3000, m.nCells)
    zs = [0.05, 0.25, 0.7, 1.5]
    cfg = I.InterpConfig(interp_diag=True, wrf_mod_vars=True,
                         diag_list=[("t2m", "T2"), ("u10", "U10"), ("v10", "V10"), ("refl10cm", "REFL_10CM")],
                         hist_2d=[("xland", "XLAND"), ("skintemp", "TSK"), ("snow", "SNOW")],
                         hist_3d=[("zgrid", "PHB"), ("theta", "T"), ("uReconstructZonal", "U"), ("uReconstructMeridional", "V"),
                                  ("pressure", "P_HYD"), ("rho", "MUB")], hist_soil=[("tslb", "TSLB")])
    nml = dict(dx=30000.0, ref_lat=38.5, ref_lon=-97.5, truelat1=38.5, truelat2=38.5, stand_lon=-97.5)
    target = T.define_target_grid_params("lambert", 181, 107, dy=30000.0, arrays=False, **nml)
    paths = {}
    for fmt in (5, 4):
        d = tmp_path / ("in%d" % fmt)
        d.mkdir()
        paths[fmt] = (d / "grid.nc", d / "hist.nc", d / "diag.nc")
        io_nc.write_mpas_files(paths[fmt][0], paths[fmt][1], m, ter, zs, hist, nz, nsoil, diag_path=paths[fmt][2], diag=diag, xtime="2024-08-07_06:00:00",
                               fmt=fmt)
        assert open(paths[fmt][1], "rb").read(4) == (b"\x89HDF" if fmt == 4 else b"CDF\x05")
    outs = {}
    for name, fin, fout, dev in (("classic", 5, 5, True), ("nc4_in", 4, 5, False), ("nc4_in_dev", 4, 5, True), ("nc4_out", 5, 4, False), ("nc4_out_dev", 5, 4, True),
                                 ("nc4_both", 4, 4, True)):
        outs[name] = tmp_path / (name + ".nc")
        g, h, dg = paths[fin]
        io_nc.run(g, h, outs[name], target, cfg, diag_path=dg, namelist=nml, fmt=fout, device_io=dev)
    assert open(outs["nc4_both"], "rb").read(4) == b"\x89HDF" and open(outs["nc4_in"], "rb").read(3) == b"CDF"
    assert open(outs["classic"], "rb").read() == open(outs["nc4_in"], "rb").read() == open(outs["nc4_in_dev"], "rb").read()   # the input's container changes nothing
    for name in ("nc4_out", "nc4_out_dev", "nc4_both"):
        _same_files(ncio4, outs["classic"], outs[name])
    py39 = "/opt/conda/bin/python3.9"
    if os.path.exists(py39):                                             # an independent reader of the NetCDF-4 output
        code = ("import sys, json, h5py, numpy as np\n"
                "f = h5py.File(sys.argv[1], 'r'); t = f['T']\n"
                "print(json.dumps(dict(shape=list(t.shape), dtype=str(t.dtype), scales=[[s.name for s in t.dims[i].values()] for i in range(t.ndim)], "
                "map_proj=int(f.attrs['MAP_PROJ'][0]), total=float(np.asarray(t[...], np.float64).sum()))))\n")
        r = subprocess.run([py39, "-c", code, str(outs["nc4_both"])], capture_output=True, text=True, timeout=120)
        if r.returncode == 0:
            import json
            got = json.loads(r.stdout)
            with ncio4.Reader(outs["classic"]) as rc:
                want = float(rc.get("T").astype(np.float64).sum())
            assert got["shape"] == [1, nz, 106, 180] and got["dtype"] == "float32" and got["map_proj"] == 1
            assert got["scales"] == [["/Time"], ["/bottom_top"], ["/south_north"], ["/west_east"]] and abs(got["total"] - want) <= 1e-9 * abs(want)
        else:
            assert "No module named" in r.stderr, r.stderr


def test_fortran_driver_netcdf4_in_and_out(gpu_lib, ncio4, tmp_path, regional_case):
    from mpassit_amd import io_nc
    from test_fortran_driver import NAMELIST, _driver
    from test_fortran_driver_nc_gpu import _write_inputs
    m, _ = regional_case
    d5, d4 = tmp_path / "c5", tmp_path / "n4"
    d5.mkdir()
    d4.mkdir()
    _write_inputs(str(d5), m)
    # the same inputs as NetCDF-4 files: every variable and attribute copied through ncio into HDF5 containers
    for name in ("init.nc", "hist.nc", "diag.nc"):
        with ncio4.Reader(d5 / name) as r:
            w = ncio4.Writer(d4 / name, format=4)
            for dn, n in r.dims.items():
                w.def_dim(dn, None if dn == r.unlimited else n)
            for att in ("config_start_time", "config_dt", "output_interval", "config_lsm_scheme", "config_microp_scheme", "config_convection_scheme", "sphere_radius",
                        "on_a_sphere", "model_name"):
                try:
                    w.put_att(att, r.att(att))
                except ncio4.NcioError:
                    pass
            for k, v in r.vars.items():
                w.def_var(k, v["type"], v["dims"])
            for k, v in r.vars.items():
                if v["record"]:
                    for rec in range(r.numrecs):
                        a = r.get(k, rec=rec)
                        w.put(k, a.tobytes() if v["type"] == ncio4.CHAR else a, rec=rec)
                else:
                    a = r.get(k)
                    w.put(k, a.tobytes() if v["type"] == ncio4.CHAR else a)
            w.close()
    for f in ("diaglist", "histlist_2d", "histlist_3d", "histlist_soil"):
        (d4 / f).write_text((d5 / f).read_text())
    exe = _driver()
    runs = {}
    for tag, d, env in (("classic", d5, {}), ("nc4_in", d4, {}), ("nc4_out", d5, {"MPASSIT_OUTPUT_FORMAT": "netcdf4"}), ("nc4_both", d4, {"MPASSIT_OUTPUT_FORMAT": "netcdf4"})):
        out = "out_%s.nc" % tag
        open(os.path.join(str(d), "namelist." + tag), "w").write(NAMELIST.format(d=str(d)).replace(".raw", ".nc").replace("out.nc", out))
        r = subprocess.run([exe, "namelist." + tag], cwd=str(d), capture_output=True, text=True, timeout=300, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stdout + r.stderr
        assert ("FIELDS STAY ON THE DEVICE" in r.stdout) == (tag == "classic"), r.stdout         # NetCDF-4 at either end: host arrays through libhdf5
        runs[tag] = d / out
    assert open(runs["nc4_out"], "rb").read(4) == b"\x89HDF" and open(runs["nc4_in"], "rb").read(4) == b"CDF\x05"
    assert open(runs["classic"], "rb").read() == open(runs["nc4_in"], "rb").read()
    _same_files(ncio4, runs["classic"], runs["nc4_out"])
    _same_files(ncio4, runs["classic"], runs["nc4_both"])
    r = subprocess.run([exe, "namelist.classic"], cwd=str(d5), capture_output=True, text=True, timeout=300, env=dict(os.environ, MPASSIT_OUTPUT_FORMAT="hdf"))
    assert r.returncode != 0 and "MPASSIT_OUTPUT_FORMAT must be" in r.stdout + r.stderr
