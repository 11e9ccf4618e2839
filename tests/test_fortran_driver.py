"""The Fortran driver (reference surface: `mpassit <namelist>`, &config, parm lists) over the ISO_C_BINDING
boundary.  CPU: it builds with amdflang, links the C-ABI library and aborts like the reference's
error_handler (code 999) when no GPU is present.  GPU: full run against the Python mirror of interp.F90."""
import os
import subprocess

import numpy as np
import pytest

from test_fields import HIST_2D, HIST_3D, SOIL

DIAG = [("rainc", "RAINC"), ("u10", "U10"), ("v10", "V10"), ("refl10cm", "REFL_10CM"), ("t2m", "T2")]

NAMELIST = """&config
  grid_file_input_grid="{d}/init.raw"
  hist_file_input_grid="{d}/hist.raw"
  diag_file_input_grid="{d}/diag.raw"
  output_file="{d}/out.raw"
  target_grid_type = 'lambert'
  interp_diag=.true.
  interp_hist=.true.
  wrf_mod_vars=.true.
  esmf_log=.false.
  nx = 151
  ny = 91
  dx = 30000.0
  dy = 30000.0
  ref_lat = 38.50
  ref_lon = -97.50
  truelat1 = 38.5
  truelat2 = 38.5
  stand_lon = -97.5
/
"""


def _driver():
    from mpassit_amd import build
    build.build()
    return build.build_fortran()


def _write_case(d, m, nz=6, nsoil=4):
    from mpassit_amd import rawio, synth
    rng = np.random.default_rng(11)
    hist, diag = {}, {}
    for k, (n, _) in enumerate(HIST_2D):
        if n == "xland":
            hist[n] = synth.category_field(m.nCells, 1, ncat=3)[0]
        elif n in ("snow", "snowh"):
            hist[n] = synth.snow_field(m.latCell, m.lonCell)[0] * (1 + k)
        else:
            hist[n] = synth.analytic_field(m.latCell, m.lonCell, 1, seed=20 + k)[0] + 280
    for k, (n, _) in enumerate(HIST_3D):  # MPAS file order [nCells][nlev]
        hist[n] = synth.analytic_field(m.latCell, m.lonCell, nz + 1 if n in ("zgrid", "w") else nz, seed=40 + k, cell_fast=False)
    for k, (n, _) in enumerate(SOIL):
        hist[n] = rng.uniform(0, 1, (m.nCells, nsoil))
    for k, (n, _) in enumerate(DIAG):
        a = synth.analytic_field(m.latCell, m.lonCell, nz if n == "refl10cm" else 1, seed=70 + k, cell_fast=False)
        diag[n] = a if n == "refl10cm" else a[:, 0].copy()
    ter = synth.analytic_field(m.latCell, m.lonCell, 1, seed=1)[0] * 1000
    rawio.write(os.path.join(d, "init.raw"), dict(latCell=m.latCell, lonCell=m.lonCell, latVertex=m.latVertex, lonVertex=m.lonVertex,
                                                   verticesOnCell=m.verticesOnCell, ter=ter))
    rawio.write(os.path.join(d, "hist.raw"), hist)
    rawio.write(os.path.join(d, "diag.raw"), diag)
    for fname, lst in (("diaglist", DIAG), ("histlist_2d", HIST_2D), ("histlist_3d", HIST_3D), ("histlist_soil", SOIL)):
        with open(os.path.join(d, fname), "w") as f:
            f.write("".join("%s\t\t%s\n" % p for p in lst))
    open(os.path.join(d, "namelist.input"), "w").write(NAMELIST.format(d=d))
    return hist, diag, ter


def _gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_gpu(), reason="checks the no-GPU failure mode")
def test_driver_builds_and_aborts_999_without_gpu(tmp_path, regional_case):
    exe = _driver()
    m, _ = regional_case
    _write_case(str(tmp_path), m)
    r = subprocess.run([exe, "namelist.input"], cwd=str(tmp_path), capture_output=True, text=True)
    assert r.returncode != 0
    assert "FATAL ERROR" in r.stdout and "no CPU fallback" in r.stdout     # error_handler contract, utils.F90:28-31
    # malformed namelist (the reference's own parm/namelist.input is malformed too)
    (tmp_path / "bad.nml").write_text("grid_file_input_grid-'x'\n")
    r = subprocess.run([exe, "bad.nml"], cwd=str(tmp_path), capture_output=True, text=True)
    assert r.returncode != 0 and "READING SETUP NAMELIST" in r.stdout


@pytest.mark.gpu
def test_driver_matches_python_mirror(tmp_path, gpu_lib, regional_case):
    from mpassit_amd import interp as I, rawio, regrid as R
    exe = _driver()
    m, g = regional_case
    d = str(tmp_path)
    hist, diag, ter = _write_case(d, m)
    r = subprocess.run([exe, "namelist.input"], cwd=d, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    got = rawio.read(os.path.join(d, "out.raw"))
    inp = I.InputData(nz=6, nzp1=7, nsoil=4, hgt=ter, layout=R.LAYOUT_LEV_FAST, hist=hist, diag=diag)
    cfg = I.InterpConfig(wrf_mod_vars=True, keep_mass_winds=True, diag_list=DIAG, hist_2d=HIST_2D, hist_3d=HIST_3D, hist_soil=SOIL)
    import copy
    gd = copy.copy(g)            # both hosts evaluate the projection on the device (mpg_grid_create_proj)
    mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_proj(gd)
    want = I.interp_data(mesh, grid, gd, inp, cfg)
    mesh.destroy()
    grid.destroy()
    assert set(want) == set(got)
    for k, w in want.items():
        assert got[k].shape == w.shape, k
        assert np.array_equal(got[k], w), k                      # same kernels on the same grid and inputs: bit exact
