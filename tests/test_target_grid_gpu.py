"""Device-side target grid (mpg_grid_create_proj) against the host mirror / oracle of
define_target_grid_params, which is pinned by the compiled-reference projection goldens (tests/golden/projection_lc.json).
Tolerance: 1e-12 degrees / 1e-13 relative -- same float64 operations, only the device libm's last bit may differ."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "projection_lc.json")))
DEG_TOL = 1e-12


def _lon_close(a, b, tol):
    d = np.abs(a - b)
    return np.minimum(d, np.abs(d - 360.0)).max() < tol


@pytest.mark.parametrize("case", ["readme_lambert", "two_truelats_south", "regional_latlon", "global_latlon", "polar_north", "polar_south",
                                  "mercator"])
def test_device_grid_matches_host_mirror(oracle, gpu_lib, case):
    from mpassit_amd import regrid as R, target_grid as T
    kw = {"readme_lambert": dict(target_grid_type="lambert", nx=1801, ny=1061, dx=3000.0, dy=3000.0, ref_lat=38.5, ref_lon=-97.5,
                                 truelat1=38.5, truelat2=38.5, stand_lon=-97.5),
          "two_truelats_south": dict(target_grid_type="lambert", nx=301, ny=201, dx=12000.0, dy=12000.0, ref_lat=-35.0, ref_lon=140.0,
                                     truelat1=-30.0, truelat2=-60.0, stand_lon=135.0, ref_x=100.0, ref_y=80.0),
          "regional_latlon": dict(target_grid_type="lat-lon", nx=201, ny=201, dx=0.1, dy=0.1, ref_lat=30.0, ref_lon=-110.0, ref_x=1.0,
                                  ref_y=1.0, stand_lon=-110.0),
          "global_latlon": dict(target_grid_type="lat-lon", nx=361, ny=181, stand_lon=0.0, is_regional=False),
          "polar_north": dict(target_grid_type="polar", nx=401, ny=351, dx=15000.0, dy=15000.0, ref_lat=78.0, ref_lon=-40.0, truelat1=70.0,
                              stand_lon=-45.0),                      # the pole lies inside this domain
          "polar_south": dict(target_grid_type="polar", nx=201, ny=221, dx=25000.0, dy=25000.0, ref_lat=-75.0, ref_lon=100.0, truelat1=-71.0,
                              stand_lon=90.0, ref_x=60.0, ref_y=80.0),
          "mercator": dict(target_grid_type="mercator", nx=501, ny=301, dx=12000.0, dy=12000.0, ref_lat=8.0, ref_lon=150.0, truelat1=15.0,
                           stand_lon=150.0)}[case]       # crosses the date line
    host = T.define_target_grid_params(**kw)
    lean = T.define_target_grid_params(arrays=False, **kw)
    assert lean.lat is None and lean.nx == host.nx
    g = R.Grid.from_proj(lean)
    for st, lon_h, lat_h in ((R.STAGGERLOC_CENTER, host.lon, host.lat), (R.STAGGERLOC_EDGE1, host.lon_u, host.lat_u),
                             (R.STAGGERLOC_EDGE2, host.lon_v, host.lat_v), (R.STAGGERLOC_CORNER, host.lon_c, host.lat_c)):
        lon_d, lat_d = g.coords(st)
        assert lon_d.shape == lon_h.shape
        assert np.abs(lat_d - lat_h).max() < DEG_TOL and _lon_close(lon_d, lon_h, DEG_TOL)
    for st, lat_h in ((R.STAGGERLOC_CENTER, host.lat), (R.STAGGERLOC_EDGE1, host.lat_u), (R.STAGGERLOC_EDGE2, host.lat_v)):
        if host.proj.code in (T.PROJ_PS, T.PROJ_MERC):
            op = (oracle.Proj.polar(kw["truelat1"], kw["stand_lon"], host.proj.lat1, host.proj.lon1, host.proj.knowni, host.proj.knownj, kw["dx"])
                  if host.proj.code == T.PROJ_PS else
                  oracle.Proj.mercator(kw["truelat1"], host.proj.lat1, host.proj.lon1, host.proj.knowni, host.proj.knownj, kw["dx"]))
            want = op.map_factor(lat_h)
            np.testing.assert_allclose(T.get_map_factor(host.proj, lat_h), want, rtol=1e-13)
        else:
            want = oracle.map_factor(lat_h, host.proj.code, kw.get("truelat1", 0.0), kw.get("truelat2", 0.0))
        np.testing.assert_allclose(g.mapfac(st), want, rtol=1e-12)
    if host.proj.code == T.PROJ_LC:
        cosa, sina = g.rotang()
        cosa_o, sina_o = oracle.get_rotang(host.lat, host.lon)
        # alpha comes from differences of neighbouring coordinates: 1e-12-degree noise over a ~0.03-degree step
        assert np.abs(cosa - cosa_o).max() < 1e-9 and np.abs(sina - sina_o).max() < 1e-9
        assert np.abs(lean.cosa - host.cosa).max() < 1e-9          # from_proj filled the lean TargetGrid
        assert np.abs(cosa ** 2 + sina ** 2 - 1).max() < 1e-14
    else:
        with pytest.raises(gpu_lib.MpgError):
            g.rotang()
    g.destroy()


def test_device_grid_hits_the_compiled_reference_goldens(gpu_lib):
    """SURVEY App. E values (Fortran reference compiled with -fdefault-real-8) straight from the device arrays."""
    from mpassit_amd import regrid as R, target_grid as T
    nl = GOLD["namelist"]
    lean = T.define_target_grid_params("lambert", nl["nx"], nl["ny"], dx=nl["dx"], dy=nl["dx"], ref_lat=nl["ref_lat"],
                                       ref_lon=nl["ref_lon"], truelat1=nl["truelat1"], truelat2=nl["truelat2"],
                                       stand_lon=nl["stand_lon"], arrays=False)
    g = R.Grid.from_proj(lean)
    lon, lat = g.coords(R.STAGGERLOC_CENTER)
    assert (lat[0, 0], lon[0, 0]) == pytest.approx(tuple(GOLD["mass_1_1"]), rel=1e-13)
    assert (lat[-1, -1], lon[-1, -1]) == pytest.approx(tuple(GOLD["mass_1800_1060"]), rel=1e-13)
    lon_c, lat_c = g.coords(R.STAGGERLOC_CORNER)
    assert (lat_c[0, 0], lon_c[0, 0]) == pytest.approx(tuple(GOLD["corner_1_1"]), rel=1e-13)
    g.destroy()


def test_job_on_device_grid_equals_job_on_host_grid(gpu_lib):
    """The Stores only see unit vectors: weights built on the device-generated grid reproduce the host-grid job."""
    from mpassit_amd import regrid as R, synth, target_grid as T, workloads
    m, host, nlev, _ = workloads.workload("tiny")
    lean = T.define_target_grid_params("lambert", 181, 107, dx=30000.0, dy=30000.0, ref_lat=38.5, ref_lon=-97.5, truelat1=38.5,
                                       truelat2=38.5, stand_lon=-97.5, arrays=False)
    mesh = R.Mesh.from_mpas(m)
    gh, gd = R.Grid.from_target(host), R.Grid.from_proj(lean)
    f = synth.analytic_field(m.latCell, m.lonCell, nlev)
    snow = synth.snow_field(m.latCell, m.lonCell)
    for method, src, nl in ((R.REGRIDMETHOD_BILINEAR, f, nlev), (R.REGRIDMETHOD_CONSERVE, snow, 1)):
        a = R.regrid_store(mesh, gh, method)
        b = R.regrid_store(mesh, gd, method)
        np.testing.assert_allclose(b.regrid(src, nlev=nl), a.regrid(src, nlev=nl), rtol=1e-9, atol=1e-9)
        a.release()
        b.release()
    a, b = R.regrid_store(mesh, gh, R.REGRIDMETHOD_NEAREST_STOD), R.regrid_store(mesh, gd, R.REGRIDMETHOD_NEAREST_STOD)
    assert (a.weights()[0] != b.weights()[0]).mean() < 1e-4         # ties aside, the same nearest cells
    for st in (R.STAGGERLOC_EDGE1, R.STAGGERLOC_EDGE2):
        a, b = R.regrid_store_grid(gh, st), R.regrid_store_grid(gd, st)
        mass = np.random.default_rng(2).normal(size=(3, host.ny, host.nx))
        np.testing.assert_allclose(b.regrid(mass.reshape(-1), nlev=3), a.regrid(mass.reshape(-1), nlev=3), rtol=0, atol=1e-9)
        a.release()
        b.release()
    with pytest.raises(gpu_lib.MpgError):
        gh.coords(R.STAGGERLOC_CENTER)                               # caller-array grids do not echo coordinates back
    for x in (gh, gd, mesh):
        x.destroy()
