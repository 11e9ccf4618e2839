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


def test_driver_netcdf_in_netcdf_out(tmp_path, gpu_lib, regional_case):
    from mpassit_amd import interp as I, io_nc, ncio, post, regrid as R, synth
    exe = _driver()
    m, g = regional_case
    d = str(tmp_path)
    nz, nsoil = 6, 4
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
                           diag_path=os.path.join(d, "diag.nc"), diag=diag, xtime="2024-08-07_12:00:00", fmt=5)
    for fname, lst in (("diaglist", DIAG), ("histlist_2d", HIST_2D), ("histlist_3d", HIST_3D), ("histlist_soil", SOIL)):
        with open(os.path.join(d, fname), "w") as f:
            f.write("".join("%s\t\t%s\n" % p for p in lst))
    open(os.path.join(d, "namelist.input"), "w").write(NAMELIST.format(d=d).replace(".raw", ".nc"))
    r = subprocess.run([exe, "namelist.input"], cwd=d, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr

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
        assert f.att("MAP_PROJ")[0] == 1 and abs(f.att("TRUELAT1")[0] - 38.5) < 1e-6
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
