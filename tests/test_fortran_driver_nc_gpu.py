"""The Fortran driver on NetCDF files: `mpassit namelist.input` reads an MPAS grid / history / diag file set (classic
format, written here with this repo's writer), regrids on the GPU through the ISO_C_BINDING boundary and writes a
WRF-style CDF-5 file with the reference's post-ops; the result must equal the Python mirror (io_nc + post) bit for bit
in float32 (same device kernels, same device-built target grid, same float64 -> float32 conversion)."""
import os
import subprocess

import numpy as np
import pytest

from test_fields import HIST_2D, HIST_3D, SOIL
from test_fortran_driver import DIAG, NAMELIST, _driver

pytestmark = pytest.mark.gpu


def _write_inputs(d, m, nz=6, nsoil=4):
    """Synthetic MPAS grid / history / diag files (float32, CDF-5) + the parm lists in directory d."""
    from mpassit_amd import io_nc, synth
    rng = np.random.default_rng(23)
    hist, diag = {}, {}
    for k, (n, _) in enumerate(HIST_2D):
        if n == "xland":
            hist[n] = synth.category_field(m.nCells, 1, ncat=3)[0]
        elif n in ("snow", "snowh"):
            hist[n] = synth.snow_field(m.latCell, m.lonCell)[0] * (1 + k)
        else:
            hist[n] = synth.analytic_field(m.latCell, m.lonCell, 1, seed=20 + k)[0] + 280
    for k, (n, _) in enumerate(HIST_3D):
        L = nz + 1 if n in ("zgrid", "w") else nz
        hist[n] = synth.analytic_field(m.latCell, m.lonCell, L, seed=40 + k, cell_fast=False) * (1.0e4 if n == "pressure" else 1.0) + \
            (5.0e4 if n == "pressure" else 0.0)
    for k, (n, _) in enumerate(SOIL):
        hist[n] = rng.uniform(0, 1, (m.nCells, nsoil))
    for k, (n, _) in enumerate(DIAG):
        a = synth.analytic_field(m.latCell, m.lonCell, nz if n == "refl10cm" else 1, seed=70 + k, cell_fast=False)
        diag[n] = a if n == "refl10cm" else a[:, 0].copy()
    hist = {k: np.asarray(v, np.float32) for k, v in hist.items()}      # history files are single precision
    diag = {k: np.asarray(v, np.float32) for k, v in diag.items()}
    ter = synth.analytic_field(m.latCell, m.lonCell, 1, seed=1)[0] * 1000
    io_nc.write_mpas_files(os.path.join(d, "init.nc"), os.path.join(d, "hist.nc"), m, ter, [0.05, 0.25, 0.7, 1.5], hist, nz, nsoil,
                           diag_path=os.path.join(d, "diag.nc"), diag=diag, xtime="2024-08-07_12:00:00", fmt=5, start_time="2024-08-07_00:00:00")
    for fname, lst in (("diaglist", DIAG), ("histlist_2d", HIST_2D), ("histlist_3d", HIST_3D), ("histlist_soil", SOIL)):
        with open(os.path.join(d, fname), "w") as f:
            f.write("".join("%s\t\t%s\n" % p for p in lst))
    return ter


def test_driver_netcdf_in_netcdf_out(tmp_path, gpu_lib, regional_case):
    from mpassit_amd import interp as I, io_nc, ncio, post, regrid as R
    exe = _driver()
    m, g = regional_case
    d = str(tmp_path)
    nz, nsoil = 6, 4
    ter = _write_inputs(d, m, nz, nsoil)
    open(os.path.join(d, "namelist.input"), "w").write(NAMELIST.format(d=d).replace(".raw", ".nc"))
    r = subprocess.run([exe, "namelist.input"], cwd=d, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "FIELDS STAY ON THE DEVICE" in r.stdout                  # NetCDF in and out: the device-resident flow
    # the same job on host arrays (the flow of the raw container): the same file, byte for byte
    open(os.path.join(d, "namelist.host"), "w").write(NAMELIST.format(d=d).replace(".raw", ".nc").replace("out.nc", "out_host.nc"))
    r = subprocess.run([exe, "namelist.host"], cwd=d, capture_output=True, text=True, timeout=300, env=dict(os.environ, MPASSIT_HOST_ARRAYS="1"))
    assert r.returncode == 0 and "FIELDS STAY ON THE DEVICE" not in r.stdout, r.stdout + r.stderr
    assert open(os.path.join(d, "out.nc"), "rb").read() == open(os.path.join(d, "out_host.nc"), "rb").read()

    cfg = I.InterpConfig(wrf_mod_vars=True, diag_list=DIAG, hist_2d=HIST_2D, hist_3d=HIST_3D, hist_soil=SOIL)
    inp, _, valid = io_nc.read_input_data(os.path.join(d, "hist.nc"), cfg, ter, diag_path=os.path.join(d, "diag.nc"))
    assert valid == "2024-08-07_12:00:00"
    import copy
    gd = copy.copy(g)                                                   # both hosts build the target grid on the device
    mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_proj(gd)
    want = post.output_fields(I.interp_data(mesh, grid, gd, inp, cfg), cfg)
    mapfac_u = grid.mapfac(R.STAGGERLOC_EDGE1)
    mesh.destroy()
    grid.destroy()
    with ncio.Reader(os.path.join(d, "out.nc")) as f:
        assert f.format == 5 and f.unlimited == "Time" and f.numrecs == 1
        assert f.dims["west_east"] == g.nx and f.dims["south_north_stag"] == g.ny + 1 and f.dims["bottom_top"] == nz
        assert f.get("Times", rec=0).tobytes() == b"2024-08-07_12:00:00"
        assert f.att("MAP_PROJ")[0] == 1 and abs(f.att("TRUELAT1")[0] - 38.5) < 1e-6 and abs(f.att("DX")[0] - 30000.0) < 1e-3
        np.testing.assert_allclose(f.get("XLAT", rec=0), g.lat, atol=1e-5)
        np.testing.assert_allclose(f.get("COSALPHA", rec=0), g.cosa, atol=1e-6)
        missing = [k for k in want if k not in f.vars]
        assert not missing, missing
        for k, w in want.items():
            got = f.get(k, rec=0)
            if k == "P_TOP":
                assert got == w
                continue
            if k == "Z_C":
                assert got.shape[0] == nz + 1 and not got[nz].any()
                got = got[:nz]
            assert got.shape == w.shape and got.dtype == np.float32, k
            assert np.array_equal(got, w), k                            # same kernels, same grid, same casts: bit exact
        assert np.array_equal(f.get("MAPFAC_U", rec=0), mapfac_u.astype(np.float32))
        np.testing.assert_allclose(f.get("ZS", rec=0), [0.05, 0.25, 0.7, 1.5], rtol=1e-6)
        assert f.vars["U"]["dims"] == ("Time", "bottom_top", "south_north", "west_east_stag")
        assert f.vars["V"]["dims"] == ("Time", "bottom_top", "south_north_stag", "west_east")
        assert f.vars["TSLB"]["dims"][1] == "soil_layers_stag" and f.vars["PHB"]["dims"][1] == "bottom_top_stag"
        # XTIME / ITIMESTEP as the reference computes them (write_data.F90:1211-1240): datetime(start) - datetime(valid)
        assert float(f.get("XTIME", rec=0)) == np.float32(-720.0) and int(f.get("ITIMESTEP", rec=0)) == -2160
    # ---- header parity: both writers against the list transcribed from write_data.F90:177-560, and against each other
    from _header_check import check_output_header
    start = "2024-08-07_00:00:00"
    h_f = check_output_header(os.path.join(d, "out.nc"), True, 1, start, config_dt=20.0, physics=(2, 8, 3), prec_acc_dt=3600)
    io_nc.write_target_data(os.path.join(d, "out_py.nc"), gd, R.Grid.from_proj(gd), want, cfg, inp.nz, inp.nzp1, inp.nsoil, valid_time=valid,
                            zs=[0.05, 0.25, 0.7, 1.5], meta=inp.meta,
                            namelist=dict(dx=30000.0, ref_lat=38.5, ref_lon=-97.5, truelat1=38.5, truelat2=38.5, stand_lon=-97.5))
    h_p = check_output_header(os.path.join(d, "out_py.nc"), True, 1, start, config_dt=20.0, physics=(2, 8, 3), prec_acc_dt=3600)
    assert h_f["dims"] == h_p["dims"] and h_f["gatts"] == h_p["gatts"]
    vf, vp = {v["name"]: v for v in h_f["vars"]}, {v["name"]: v for v in h_p["vars"]}
    assert set(vf) == set(vp), set(vf) ^ set(vp)
    for name in vf:
        assert vf[name]["dims"] == vp[name]["dims"] and vf[name]["type"] == vp[name]["type"], name


def test_driver_with_a_file_defined_target_grid(tmp_path, gpu_lib, regional_case):
    """target_grid_type = 'file': the Fortran driver reads the grid from a WRF geo_em style file (classic format) and
    rebuilds the CORNER stagger with get_cell_corners as written; compared with the Python mirror of the same path."""
    from mpassit_amd import interp as I, io_nc, ncio, post, regrid as R, target_grid as T
    exe = _driver()
    m, g = regional_case
    d = str(tmp_path)
    nz, nsoil = 6, 4
    ter = _write_inputs(d, m, nz, nsoil)
    geo = os.path.join(d, "geo_em.d01.nc")
    with ncio.Writer(geo, format=2) as w:
        for name, n in (("Time", None), ("west_east", g.nx), ("south_north", g.ny), ("west_east_stag", g.nx + 1), ("south_north_stag", g.ny + 1)):
            w.def_dim(name, n)
        for k, v in (("DX", np.float32(30000.0)), ("DY", np.float32(30000.0)), ("CEN_LAT", np.float32(38.5)), ("CEN_LON", np.float32(-97.5)),
                     ("TRUELAT1", np.float32(38.5)), ("TRUELAT2", np.float32(38.5)), ("MOAD_CEN_LAT", np.float32(38.5)),
                     ("STAND_LON", np.float32(-97.5)), ("POLE_LAT", np.float32(90.0)), ("POLE_LON", np.float32(0.0)), ("MAP_PROJ", 1)):
            w.put_att(k, v)
        dims = {"M": ("Time", "south_north", "west_east"), "U": ("Time", "south_north", "west_east_stag"),
                "V": ("Time", "south_north_stag", "west_east")}
        data = {"XLAT_M": (g.lat, "M"), "XLONG_M": (g.lon, "M"), "XLAT_U": (g.lat_u, "U"), "XLONG_U": (g.lon_u, "U"),
                "XLAT_V": (g.lat_v, "V"), "XLONG_V": (g.lon_v, "V"), "SINALPHA": (g.sina, "M"), "COSALPHA": (g.cosa, "M"),
                "MAPFAC_M": (np.full_like(g.lat, 1.01), "M"), "MAPFAC_U": (np.full_like(g.lat_u, 1.02), "U"),
                "MAPFAC_V": (np.full_like(g.lat_v, 1.03), "V")}
        for name, (_, st) in data.items():
            w.def_var(name, ncio.FLOAT, dims[st])
        for name, (a, _) in data.items():
            w.put(name, a, rec=0)
    nml = NAMELIST.format(d=d).replace(".raw", ".nc").replace("target_grid_type = 'lambert'",
                                                               "target_grid_type = 'file'\n  file_target_grid=\"%s\"" % geo)
    nml = "\n".join(ln for ln in nml.splitlines() if ln.split("=")[0].strip() not in
                    ("nx", "ny", "dx", "dy", "ref_lat", "ref_lon", "truelat1", "truelat2", "stand_lon")) + "\n"
    open(os.path.join(d, "namelist.input"), "w").write(nml)
    r = subprocess.run([exe, "namelist.input"], cwd=d, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr

    t = T.define_target_grid_file(geo)
    cfg = I.InterpConfig(wrf_mod_vars=True, diag_list=DIAG, hist_2d=HIST_2D, hist_3d=HIST_3D, hist_soil=SOIL)
    inp, _, _ = io_nc.read_input_data(os.path.join(d, "hist.nc"), cfg, ter, diag_path=os.path.join(d, "diag.nc"))
    mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_target(t)
    want = post.output_fields(I.interp_data(mesh, grid, t, inp, cfg), cfg)
    mesh.destroy()
    grid.destroy()
    with ncio.Reader(os.path.join(d, "out.nc")) as f:
        assert f.dims["west_east"] == g.nx and f.dims["south_north"] == g.ny and abs(f.att("DX")[0] - 30000.0) < 1e-3
        assert np.array_equal(f.get("XLAT", rec=0), g.lat.astype(np.float32)) and np.array_equal(f.get("MAPFAC_V", rec=0), np.float32(1.03) + 0 * g.lat_v.astype(np.float32))
        assert np.array_equal(f.get("SINALPHA", rec=0), g.sina.astype(np.float32))
        for k, w_ in want.items():
            got = f.get(k, rec=0)
            if k == "P_TOP":
                assert abs(got - w_) <= 1e-6 * abs(w_)
                continue
            if k == "Z_C":
                got = got[:nz]
            assert got.shape == w_.shape, k
            if k in ("XLAND", "TSLB", "SMOIS", "SH2O", "MU", "PH", "P"):
                assert np.array_equal(got, w_), k
            else:   # the corners come from two libms (flang runtime / numpy): the conservative weights may differ in the last bits
                assert np.abs(got - w_).max() <= 2e-7 * max(1.0, float(np.abs(w_).max())), k
    # the Python file job takes the same file-defined grid (grid variables = the file's own)
    io_nc.run(os.path.join(d, "init.nc"), os.path.join(d, "hist.nc"), os.path.join(d, "out_py.nc"), t, cfg, diag_path=os.path.join(d, "diag.nc"),
              namelist=dict(dx=30000.0, ref_lat=38.5, ref_lon=-97.5, truelat1=38.5, truelat2=38.5, stand_lon=-97.5), device_io=True)
    with ncio.Reader(os.path.join(d, "out.nc")) as f, ncio.Reader(os.path.join(d, "out_py.nc")) as fp:
        for k in ("XLAT", "XLONG_U", "MAPFAC_M", "MAPFAC_V", "SINALPHA", "COSALPHA", "T", "U", "V", "XLAND", "TSLB", "P_TOP"):
            assert np.array_equal(f.get(k, rec=0), fp.get(k, rec=0)), k
    # ... and so do two driver images (row blocks of a grid that came from a file): the single image's file, byte for byte
    open(os.path.join(d, "namelist.two"), "w").write(nml.replace("out.nc", "out2.nc"))
    _run_images(d, "namelist.two", 2)
    assert open(os.path.join(d, "out.nc"), "rb").read() == open(os.path.join(d, "out2.nc"), "rb").read()


def test_driver_on_a_global_latlon_grid(tmp_path, gpu_lib):
    """BASELINE configuration 5 in small: global icosahedral mesh -> global lat-lon grid (is_regional=.false.: periodic in
    i, poles closed), wrf_mod_vars staggered winds, no wind rotation; Fortran driver == Python mirror bit for bit."""
    import copy

    from mpassit_amd import interp as I, io_nc, ncio, post, regrid as R, workloads
    exe = _driver()
    m, g, _, _ = workloads.workload("c5_small")
    d = str(tmp_path)
    nz, nsoil = 6, 4
    ter = _write_inputs(d, m, nz, nsoil)
    open(os.path.join(d, "namelist.input"), "w").write("""&config
  grid_file_input_grid="%s/init.nc"
  hist_file_input_grid="%s/hist.nc"
  diag_file_input_grid="%s/diag.nc"
  output_file="%s/out.nc"
  target_grid_type = 'lat-lon'
  interp_diag=.true.
  interp_hist=.true.
  wrf_mod_vars=.true.
  is_regional=.false.
  nx = 361
  ny = 181
  stand_lon = 0.0
/
""" % (d, d, d, d))
    r = subprocess.run([exe, "namelist.input"], cwd=d, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    cfg = I.InterpConfig(wrf_mod_vars=True, proj_is_lambert=False, diag_list=DIAG, hist_2d=HIST_2D, hist_3d=HIST_3D, hist_soil=SOIL)
    inp, _, _ = io_nc.read_input_data(os.path.join(d, "hist.nc"), cfg, ter, diag_path=os.path.join(d, "diag.nc"))
    gd = copy.copy(g)
    mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_proj(gd)
    want = post.output_fields(I.interp_data(mesh, grid, gd, inp, cfg), cfg)
    lon_u, _ = grid.coords(R.STAGGERLOC_EDGE1)
    mesh.destroy()
    grid.destroy()
    with ncio.Reader(os.path.join(d, "out.nc")) as f:
        assert f.dims["west_east"] == 360 and f.dims["south_north"] == 180 and f.att("MAP_PROJ")[0] == 0
        assert "SINALPHA" not in f.vars and abs(f.att("DX")[0] - 6370000.0 * 2 * np.pi / 360) < 1.0
        assert np.array_equal(f.get("XLONG_U", rec=0), lon_u.astype(np.float32))
        for k, w_ in want.items():
            got = f.get(k, rec=0)
            if k == "P_TOP":
                assert got == w_
                continue
            if k == "Z_C":
                got = got[:nz]
            assert got.shape == w_.shape and np.array_equal(got, w_), k


@pytest.mark.parametrize("kind", ["polar", "mercator"])
def test_driver_on_polar_stereographic_and_mercator_grids(tmp_path, gpu_lib, kind):
    """target_grid_type = 'polar' / 'mercator' (program_setup.F90:174-182; set_ps / set_merc, module_map_utils.F90): the
    global mesh under a regional grid containing the pole / crossing the date line; grid variables, map factors, header
    attributes and every field: Fortran driver == Python mirror bit for bit.  No wind rotation (proj_code /= PROJ_LC,
    interp.F90:138,291) and no SINALPHA / COSALPHA (write_data.F90:447)."""
    import copy

    from mpassit_amd import interp as I, io_nc, ncio, post, regrid as R, target_grid as tg, workloads
    exe = _driver()
    m, _, _, _ = workloads.workload("c5_small")
    par = (dict(nx=121, ny=101, dx=60000.0, ref_lat=86.0, ref_lon=10.0, truelat1=70.0, stand_lon=-30.0) if kind == "polar" else
           dict(nx=141, ny=81, dx=80000.0, ref_lat=5.0, ref_lon=175.0, truelat1=20.0, stand_lon=175.0))
    g = tg.define_target_grid_params(kind, par["nx"], par["ny"], dx=par["dx"], dy=par["dx"], ref_lat=par["ref_lat"], ref_lon=par["ref_lon"],
                                     truelat1=par["truelat1"], stand_lon=par["stand_lon"])
    d = str(tmp_path)
    nz, nsoil = 6, 4
    ter = _write_inputs(d, m, nz, nsoil)
    open(os.path.join(d, "namelist.input"), "w").write("""&config
  grid_file_input_grid="%s/init.nc"
  hist_file_input_grid="%s/hist.nc"
  diag_file_input_grid="%s/diag.nc"
  output_file="%s/out.nc"
  target_grid_type = '%s'
  interp_diag=.true.
  interp_hist=.true.
  wrf_mod_vars=.true.
  nx = %d
  ny = %d
  dx = %.1f
  dy = %.1f
  ref_lat = %.1f
  ref_lon = %.1f
  truelat1 = %.1f
  stand_lon = %.1f
/
""" % (d, d, d, d, kind, par["nx"], par["ny"], par["dx"], par["dx"], par["ref_lat"], par["ref_lon"], par["truelat1"], par["stand_lon"]))
    r = subprocess.run([exe, "namelist.input"], cwd=d, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    cfg = I.InterpConfig(wrf_mod_vars=True, proj_is_lambert=False, diag_list=DIAG, hist_2d=HIST_2D, hist_3d=HIST_3D, hist_soil=SOIL)
    inp, _, _ = io_nc.read_input_data(os.path.join(d, "hist.nc"), cfg, ter, diag_path=os.path.join(d, "diag.nc"))
    gd = copy.copy(g)
    mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_proj(gd)
    want = post.output_fields(I.interp_data(mesh, grid, gd, inp, cfg), cfg)
    lon_u, lat_u = grid.coords(R.STAGGERLOC_EDGE1)
    mf_v = grid.mapfac(R.STAGGERLOC_EDGE2)
    mesh.destroy()
    grid.destroy()
    assert np.abs(lat_u - g.lat_u).max() < 1e-11                       # device grid == numpy mirror of set_ps / set_merc
    with ncio.Reader(os.path.join(d, "out.nc")) as f:
        assert f.att("MAP_PROJ")[0] == (2 if kind == "polar" else 3)
        assert f.att("MAP_PROJ_CHAR").strip() == ("Polar Stereographic" if kind == "polar" else "Mercator")
        assert "SINALPHA" not in f.vars and "COSALPHA" not in f.vars
        assert np.array_equal(f.get("XLONG_U", rec=0), lon_u.astype(np.float32))
        assert np.array_equal(f.get("XLAT_U", rec=0), lat_u.astype(np.float32))
        assert np.array_equal(f.get("MAPFAC_V", rec=0), mf_v.astype(np.float32))
        assert float(np.abs(f.get("T2", rec=0)).max()) > 0
        for k, w_ in want.items():
            got = f.get(k, rec=0)
            if k == "P_TOP":
                assert got == w_
                continue
            if k == "Z_C":
                got = got[:nz]
            assert got.shape == w_.shape and np.array_equal(got, w_), k


def _run_images(d, namelist_name, ranks):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import mpassit_ranks
    res = mpassit_ranks.launch(namelist_name, ranks, gpus=1, exe=_driver(), cwd=d, timeout=600)
    for r, (code, so, se) in enumerate(res):
        assert code == 0, "image %d: %s\n%s" % (r, so[-2000:], se[-2000:])
    return res


def test_three_driver_images_write_the_same_file(tmp_path, gpu_lib, regional_case):
    """The Fortran driver as several images, one per GPU (here: three sharing the one card) -- the reference's
    `mpirun -np N` (mpassit.F90:84-96): every image reads the files whole (input_data.F90:645), regrids its block of
    target rows (regDecomp=(/1,npets/), model_grid.F90:693; one halo row each side for the CENTER -> EDGE destaggering)
    and writes its rows into the one CDF-5 file; P_TOP is combined from the images' reductions.  The file must be the one
    a single image writes, byte for byte."""
    m, g = regional_case
    d = str(tmp_path)
    _write_inputs(d, m, 6, 4)
    nml = NAMELIST.format(d=d).replace(".raw", ".nc")
    open(os.path.join(d, "namelist.one"), "w").write(nml)
    open(os.path.join(d, "namelist.three"), "w").write(nml.replace("out.nc", "out3.nc"))
    r = subprocess.run([_driver(), "namelist.one"], cwd=d, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    res = _run_images(d, "namelist.three", 3)
    assert "DRIVER IMAGE 2 OF 3" in res[2][1]
    one, three = open(os.path.join(d, "out.nc"), "rb").read(), open(os.path.join(d, "out3.nc"), "rb").read()
    assert len(one) == len(three)
    if one != three:                                   # say which variable differs before failing
        from mpassit_amd import ncio
        with ncio.Reader(os.path.join(d, "out.nc")) as a, ncio.Reader(os.path.join(d, "out3.nc")) as b:
            bad = [k for k in a.vars if not np.array_equal(a.get(k, rec=0) if a.vars[k]["record"] else a.get(k),
                                                            b.get(k, rec=0) if b.vars[k]["record"] else b.get(k))]
        assert not bad, bad
    assert one == three
    assert not [f for f in os.listdir(d) if ".ready" in f or ".done." in f]      # the marker files are gone


def test_block_decomp_file_is_checked_like_the_reference(tmp_path, gpu_lib, regional_case):
    """`block_decomp_file` of the namelist (program_setup.F90:38,148-152; read_block_decomp_file, model_grid.F90:2367-2426): the images
    of this driver take their cells by target rows, so the MPAS partition decides nothing here -- but the file is read and checked as
    the reference checks it (exactly nCells lines, made for exactly this many processes), every image reports the cells it would own,
    and the output stays the single image's file byte for byte."""
    m, g = regional_case
    d = str(tmp_path)
    _write_inputs(d, m, 6, 4)
    owner = np.random.default_rng(4).integers(0, 2, m.nCells)
    owner[:2] = [0, 1]
    open(os.path.join(d, "mesh.graph.info.part.2"), "w").write("\n".join("%d" % p for p in owner) + "\n")
    nml = NAMELIST.format(d=d).replace(".raw", ".nc")
    open(os.path.join(d, "namelist.one"), "w").write(nml)
    with_file = nml.replace("out.nc", "out2.nc").replace("/\n", '  block_decomp_file = "%s/mesh.graph.info.part.2"\n/\n' % d)
    assert "block_decomp_file" in with_file
    open(os.path.join(d, "namelist.two"), "w").write(with_file)
    r = subprocess.run([_driver(), "namelist.one"], cwd=d, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    res = _run_images(d, "namelist.two", 2)
    for rank in range(2):
        assert "BLOCK DECOMPOSITION FILE: %d OF %d CELLS BELONG TO THIS IMAGE" % (int((owner == rank).sum()), m.nCells) in res[rank][1]
    assert open(os.path.join(d, "out.nc"), "rb").read() == open(os.path.join(d, "out2.nc"), "rb").read()
    # made for two processes, run as one: the reference's error, its exit code
    r = subprocess.run([_driver(), "namelist.two"], cwd=d, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "GENERATED FOR" in r.stdout and "PROCESSORS USED" in r.stdout


def test_two_driver_images_on_a_global_grid(tmp_path, gpu_lib):
    """Row blocks of a periodic grid with pole caps (is_regional=.false.): each image keeps the cap it touches."""
    from mpassit_amd import workloads
    m, g, _, _ = workloads.workload("c5_small")
    d = str(tmp_path)
    _write_inputs(d, m, 6, 4)
    nml = """&config
  grid_file_input_grid="%s/init.nc"
  hist_file_input_grid="%s/hist.nc"
  diag_file_input_grid="%s/diag.nc"
  output_file="%s/out.nc"
  target_grid_type = 'lat-lon'
  interp_diag=.true.
  interp_hist=.true.
  wrf_mod_vars=.true.
  is_regional=.false.
  nx = 361
  ny = 181
  stand_lon = 0.0
/
""" % (d, d, d, d)
    open(os.path.join(d, "namelist.one"), "w").write(nml)
    open(os.path.join(d, "namelist.two"), "w").write(nml.replace("out.nc", "out2.nc"))
    r = subprocess.run([_driver(), "namelist.one"], cwd=d, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    _run_images(d, "namelist.two", 2)
    assert open(os.path.join(d, "out.nc"), "rb").read() == open(os.path.join(d, "out2.nc"), "rb").read()


def test_a_failing_image_takes_the_others_down(tmp_path, gpu_lib):
    """Multi-image run whose input file does not exist: every image stops (the one that fails first is not waited for by the
    others until their marker time-out), the launcher returns the codes within seconds and nothing is left behind."""
    import sys
    import time
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import mpassit_ranks
    d = str(tmp_path)
    open(os.path.join(d, "namelist.bad"), "w").write(NAMELIST.format(d=d).replace(".raw", ".nc"))   # no input files were written
    t0 = time.monotonic()
    res = mpassit_ranks.launch("namelist.bad", 3, gpus=1, exe=_driver(), cwd=d, timeout=120)
    assert time.monotonic() - t0 < 60
    assert all(code != 0 for code, _, _ in res)
    assert not [f for f in os.listdir(d) if ".ready" in f or ".done." in f or f.endswith("out.nc")]


def test_a_damaged_grid_file_stops_the_driver_with_the_librarys_message(tmp_path, gpu_lib, regional_case):
    """verticesOnCell holding a vertex number beyond nVertices (a truncated or damaged init file): the driver stops with the library's
    message and a non-zero code before any geometry kernel has used the number as an index; nothing is written.  The Python host
    raises the same error."""
    from mpassit_amd import _lib, io_nc, synth
    m, _ = regional_case
    voc = m.verticesOnCell.copy()
    voc[m.nCells // 3, 2] = m.nVertices + 1
    bad = synth.MpasMesh(m.latCell, m.lonCell, m.latVertex, m.lonVertex, voc)
    d = str(tmp_path)
    _write_inputs(d, bad)
    open(os.path.join(d, "namelist.input"), "w").write(NAMELIST.format(d=d).replace(".raw", ".nc"))
    r = subprocess.run([_driver(), "namelist.input"], cwd=d, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "verticesOnCell refers to vertex %d of %d" % (m.nVertices + 1, m.nVertices) in r.stdout + r.stderr, r.stdout + r.stderr
    assert not os.path.exists(os.path.join(d, "out.nc"))
    from mpassit_amd import regrid as R
    got = io_nc.read_grid(os.path.join(d, "init.nc"))[0]
    with pytest.raises(_lib.MpgError, match="verticesOnCell refers to vertex"):
        R.Mesh.from_mpas(got)


def test_the_references_launch_line_works_mpiexec_starts_the_images(tmp_path, gpu_lib, regional_case):
    """`mpiexec -n 3 mpassit namelist.input` -- the reference's own launch line (mpassit.F90:84-96 asks MPI for rank and size).  This driver
    links no MPI; it reads the launcher's environment (hydra: PMI_SIZE / PMI_RANK / MPI_LOCALRANKID; Open MPI's and srun's variables the
    same way) and its images find each other through the marker files, tagged with the launcher's process id.  The file is the single
    image's, byte for byte; a batch script's SLURM_NTASKS alone (no job step) must NOT turn a plain run into rank 0 of many."""
    mpiexec = "/opt/conda/bin/mpiexec"
    if not os.path.exists(mpiexec):
        pytest.skip("no mpiexec in this image")
    m, _ = regional_case
    d = str(tmp_path)
    _write_inputs(d, m, 6, 4)
    nml = NAMELIST.format(d=d).replace(".raw", ".nc")
    open(os.path.join(d, "namelist.one"), "w").write(nml)
    open(os.path.join(d, "namelist.mpi"), "w").write(nml.replace("out.nc", "out_mpi.nc"))
    env = {k: v for k, v in os.environ.items() if not k.startswith(("MPASSIT_", "PMI_", "OMPI_", "SLURM_"))}
    r = subprocess.run([_driver(), "namelist.one"], cwd=d, capture_output=True, text=True, timeout=300, env=dict(env, SLURM_NTASKS="8", SLURM_PROCID="0"))
    assert r.returncode == 0 and "LAUNCHED AS RANK" not in r.stdout, r.stdout + r.stderr          # no job step: one image
    r = subprocess.run([mpiexec, "-n", "3", _driver(), "namelist.mpi"], cwd=d, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    for k in range(3):
        assert "LAUNCHED AS RANK %d OF 3 BY AN MPI / SLURM LAUNCHER; RUN TAG hydra" % k in r.stdout, r.stdout[-3000:]
    assert open(os.path.join(d, "out.nc"), "rb").read() == open(os.path.join(d, "out_mpi.nc"), "rb").read()
    assert not [f for f in os.listdir(d) if ".ready" in f or ".done." in f]
    # several images and a NetCDF-4 output do not go together (no byte ranges to share): said, not attempted
    r = subprocess.run([mpiexec, "-n", "2", _driver(), "namelist.mpi"], cwd=d, capture_output=True, text=True, timeout=300, env=dict(env, MPASSIT_OUTPUT_FORMAT="netcdf4"))
    assert r.returncode != 0 and "several driver images need NetCDF CLASSIC files" in r.stdout + r.stderr, r.stdout[-2000:] + r.stderr[-2000:]
    # srun's variables, set by hand for two processes started side by side (a job step: SLURM_STEP_ID is there)
    open(os.path.join(d, "namelist.srun"), "w").write(nml.replace("out.nc", "out_srun.nc"))
    ps = [subprocess.Popen([_driver(), "namelist.srun"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                           env=dict(env, SLURM_NTASKS="2", SLURM_PROCID=str(k), SLURM_LOCALID=str(k), SLURM_STEP_ID="0", SLURM_JOB_ID="4711")) for k in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in ps]
    assert all(p.returncode == 0 for p in ps), outs
    assert "RUN TAG slurm0j4711" in outs[0] and "LAUNCHED AS RANK 1 OF 2" in outs[1]
    assert open(os.path.join(d, "out.nc"), "rb").read() == open(os.path.join(d, "out_srun.nc"), "rb").read()


def test_mpassit_tune_sets_the_librarys_choices_in_the_driver(tmp_path, gpu_lib, regional_case):
    """MPASSIT_TUNE (round 6): what a site's `esmf_pin compare` named goes into the shipped driver without a rebuild.  The other bilinear line
    type moves the bilinear fields by O(h^2) (tests/test_unpinned_bounds_gpu.py) and nothing else; an unknown key or a non-integer stops the run."""
    from mpassit_amd import ncio
    m, _ = regional_case
    d = str(tmp_path)
    _write_inputs(d, m, 6, 4)
    nml = NAMELIST.format(d=d).replace(".raw", ".nc")
    open(os.path.join(d, "namelist.a"), "w").write(nml)
    open(os.path.join(d, "namelist.b"), "w").write(nml.replace("out.nc", "out_lt1.nc"))
    env = {k: v for k, v in os.environ.items() if not k.startswith("MPASSIT_")}
    r = subprocess.run([_driver(), "namelist.a"], cwd=d, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "LIBRARY CHOICE" not in r.stdout, r.stdout + r.stderr
    r = subprocess.run([_driver(), "namelist.b"], cwd=d, capture_output=True, text=True, timeout=300,
                       env=dict(env, MPASSIT_TUNE="bilinear_linetype=1,grid_inside_tol_exp=10"))
    assert r.returncode == 0 and "LIBRARY CHOICE bilinear_linetype = 1" in r.stdout and "LIBRARY CHOICE grid_inside_tol_exp = 10" in r.stdout, r.stdout + r.stderr
    with ncio.Reader(os.path.join(d, "out.nc")) as fa, ncio.Reader(os.path.join(d, "out_lt1.nc")) as fb:
        ta, tb = fa.get("T", rec=0), fb.get("T", rec=0)
        assert ta.shape == tb.shape and not np.array_equal(ta, tb) and np.abs(ta - tb).max() < 1e-2 * max(1.0, np.abs(ta).max())
        assert np.array_equal(fa.get("XLAND", rec=0), fb.get("XLAND", rec=0))            # nearest-neighbour fields do not know about line types
    for bad in ("no_such_knob=1", "bilinear_linetype=x", "bilinear_linetype"):
        r = subprocess.run([_driver(), "namelist.b"], cwd=d, capture_output=True, text=True, timeout=300, env=dict(env, MPASSIT_TUNE=bad))
        assert r.returncode != 0 and "MPASSIT_TUNE" in r.stdout + r.stderr, (bad, r.stdout[-500:], r.stderr[-500:])


def test_a_multi_node_launch_needs_a_job_wide_tag_and_takes_mpassit_run_id(tmp_path, gpu_lib, regional_case):
    """Round-5 advisor: under MPICH / hydra the run tag came from getppid(), the PER-NODE proxy, so images on different nodes waited for
    markers that never appear -- and MPASSIT_RUN_ID was ignored on that path.  Two images started side by side with hydra's variables
    saying "2 ranks, 1 on this node": without MPASSIT_RUN_ID both stop AT ONCE saying what to set; with it they run, tagged with it,
    and write the single image's bytes."""
    import time
    m, _ = regional_case
    d = str(tmp_path)
    _write_inputs(d, m, 6, 4)
    nml = NAMELIST.format(d=d).replace(".raw", ".nc")
    open(os.path.join(d, "namelist.one"), "w").write(nml)
    open(os.path.join(d, "namelist.two"), "w").write(nml.replace("out.nc", "out_two.nc"))
    env = {k: v for k, v in os.environ.items() if not k.startswith(("MPASSIT_", "PMI_", "OMPI_", "SLURM_", "MPI_"))}
    r = subprocess.run([_driver(), "namelist.one"], cwd=d, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr

    def start(extra):
        return [subprocess.Popen([_driver(), "namelist.two"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                                 env=dict(env, PMI_SIZE="2", PMI_RANK=str(k), MPI_LOCALRANKID="0", MPI_LOCALNRANKS="1", MPASSIT_WAIT_S="20", **extra))
                for k in range(2)]
    t0 = time.monotonic()
    ps = start({})
    outs = [p.communicate(timeout=120)[0] for p in ps]
    assert all(p.returncode != 0 for p in ps) and time.monotonic() - t0 < 15, outs             # refused at start-up, not after the wait
    assert all("spans several nodes" in o and "MPASSIT_RUN_ID" in o for o in outs), outs
    assert not os.path.exists(os.path.join(d, "out_two.nc"))
    ps = start({"MPASSIT_RUN_ID": "job-4711.a"})
    outs = [p.communicate(timeout=600)[0] for p in ps]
    assert all(p.returncode == 0 for p in ps), outs
    assert all("RUN TAG userjob4711a" in o for o in outs), outs
    assert open(os.path.join(d, "out.nc"), "rb").read() == open(os.path.join(d, "out_two.nc"), "rb").read()


def test_an_image_whose_peers_never_come_stops_with_a_message(tmp_path, gpu_lib, regional_case):
    """Rank 1 of 2 by srun's variables, but nobody is rank 0 (a stray environment, a rank that died before its marker): the image waits
    MPASSIT_WAIT_S and stops saying how many images it expected -- it does not hang, and it writes nothing."""
    import time
    m, _ = regional_case
    d = str(tmp_path)
    _write_inputs(d, m, 6, 4)
    open(os.path.join(d, "namelist.input"), "w").write(NAMELIST.format(d=d).replace(".raw", ".nc"))
    env = {k: v for k, v in os.environ.items() if not k.startswith(("MPASSIT_", "PMI_", "OMPI_", "SLURM_"))}
    t0 = time.monotonic()
    r = subprocess.run([_driver(), "namelist.input"], cwd=d, capture_output=True, text=True, timeout=300,
                       env=dict(env, SLURM_NTASKS="2", SLURM_PROCID="1", SLURM_LOCALID="1", SLURM_STEP_ID="3", SLURM_JOB_ID="9", MPASSIT_WAIT_S="3"))
    assert r.returncode != 0 and time.monotonic() - t0 < 120
    assert "image 1 of 2 waited 3 s for another image's marker" in r.stdout + r.stderr, r.stdout[-2000:] + r.stderr[-2000:]
    assert not os.path.exists(os.path.join(d, "out.nc"))


@pytest.mark.parametrize("flags", [dict(interp_diag=".false."), dict(interp_hist=".false."), dict(wrf_mod_vars=".false."),
                                   dict(interp_diag=".false.", wrf_mod_vars=".false.")], ids=lambda f: "+".join("%s=%s" % kv for kv in f.items()))
def test_flag_combinations_one_image_two_images_host_arrays(tmp_path, gpu_lib, regional_case, flags):
    """The namelist's switches (program_setup.F90:103-106: only the diag file, only the history file, no WRF post-ops) through the three
    flows of the driver -- device resident, host arrays, two images: each runs, and all three write the same bytes."""
    m, _ = regional_case
    d = str(tmp_path)
    _write_inputs(d, m, 6, 4)
    nml = NAMELIST.format(d=d).replace(".raw", ".nc")
    for k, v in flags.items():
        assert "%s=.true." % k in nml
        nml = nml.replace("%s=.true." % k, "%s=%s" % (k, v))
    open(os.path.join(d, "namelist.dev"), "w").write(nml)
    open(os.path.join(d, "namelist.host"), "w").write(nml.replace("out.nc", "out_host.nc"))
    open(os.path.join(d, "namelist.two"), "w").write(nml.replace("out.nc", "out_two.nc"))
    r = subprocess.run([_driver(), "namelist.dev"], cwd=d, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "FIELDS STAY ON THE DEVICE" in r.stdout, r.stdout + r.stderr
    r = subprocess.run([_driver(), "namelist.host"], cwd=d, capture_output=True, text=True, timeout=300, env=dict(os.environ, MPASSIT_HOST_ARRAYS="1"))
    assert r.returncode == 0, r.stdout + r.stderr
    _run_images(d, "namelist.two", 2)
    one = open(os.path.join(d, "out.nc"), "rb").read()
    assert one == open(os.path.join(d, "out_host.nc"), "rb").read() and one == open(os.path.join(d, "out_two.nc"), "rb").read()
    from mpassit_amd import ncio
    with ncio.Reader(os.path.join(d, "out.nc")) as f:
        assert ("T2" in f.vars) == (flags.get("interp_diag") != ".false.") and ("TSK" in f.vars) == (flags.get("interp_hist") != ".false.")
        assert ("P_TOP" in f.vars) == (flags.get("wrf_mod_vars") != ".false." and flags.get("interp_hist") != ".false.")
