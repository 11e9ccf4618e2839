"""target_grid_type = 'file' (define_target_grid_file, model_grid.F90:1203-1888): grid from a WRF geo_em / wrfinput style
file, CORNER stagger from get_cell_corners (:1902-1972) with its bearings and constants exactly as written."""
import numpy as np
import pytest


@pytest.fixture(scope="module", params=[2, 4], ids=["cdf2", "netcdf4"])
def geo_file(tmp_path_factory, request):
    from mpassit_amd import build, ncio, target_grid as T
    build.build_ncio()
    if request.param == 4 and not ncio.has_netcdf4():
        pytest.skip("libmpassit_ncio was built without HDF5")
    nml = dict(dx=30000.0, dy=30000.0, ref_lat=38.5, ref_lon=-97.5, truelat1=38.5, truelat2=38.5, stand_lon=-97.5)
    g = T.define_target_grid_params("lambert", 41, 31, **nml)
    path = tmp_path_factory.mktemp("geo") / "geo_em.d01.nc"
    with ncio.Writer(path, format=request.param) as w:                 # geo_em / wrfinput files come as classic or as NetCDF-4 files
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
                "MAPFAC_M": (np.ones_like(g.lat), "M"), "MAPFAC_U": (np.ones_like(g.lat_u), "U"), "MAPFAC_V": (np.ones_like(g.lat_v), "V"),
                "HGT_M": (np.full_like(g.lat, 123.0), "M")}
        for name, (_, st) in data.items():
            w.def_var(name, ncio.FLOAT, dims[st])
        for name, (a, _) in data.items():
            w.put(name, a, rec=0)
    return path, g


def test_file_target_grid_and_cell_corners(oracle, geo_file):
    from mpassit_amd import target_grid as T
    path, g = geo_file
    t = T.define_target_grid_file(path)
    assert (t.nx, t.ny) == (g.nx, g.ny) and t.proj.code == T.PROJ_LC and t.proj.dx == 30000.0 and t.is_regional
    for a, b in ((t.lat, g.lat), (t.lon, g.lon), (t.lat_u, g.lat_u), (t.lon_v, g.lon_v), (t.cosa, g.cosa), (t.sina, g.sina)):
        assert a.dtype == np.float64 and np.array_equal(a, b.astype(np.float32).astype(np.float64))     # NF90_FLOAT widened
    assert t.extra["hgt"][3, 4] == 123.0 and t.extra["ref_lat"] == 38.5
    latc, lonc = oracle.cell_corners(t.lat, t.lon, 30000.0)
    assert t.lat_c.shape == (g.ny + 1, g.nx + 1)
    np.testing.assert_allclose(t.lat_c, latc, rtol=0, atol=1e-12)
    np.testing.assert_allclose(t.lon_c, lonc, rtol=0, atol=1e-12)
    # the routine as written: bearing 135 puts the (i, j) "sw" corner south-EAST of mass point (i, j); the extra column
    # (bearing 225 from the last column) lies south-west of it, i.e. west of the corner before it
    assert (t.lat_c[:-1, :-1] < t.lat).all() and (t.lon_c[:-1, :-1] > t.lon).all()
    assert (t.lon_c[:-1, -1] < t.lon[:, -1]).all() and (t.lat_c[-1, :-1] > t.lat[-1]).all()
    d = 6370000.0 * np.arccos(np.clip(np.sin(np.deg2rad(t.lat)) * np.sin(np.deg2rad(t.lat_c[:-1, :-1])) +
                                      np.cos(np.deg2rad(t.lat)) * np.cos(np.deg2rad(t.lat_c[:-1, :-1])) *
                                      np.cos(np.deg2rad(t.lon_c[:-1, :-1] - t.lon)), -1, 1))
    np.testing.assert_allclose(d, 30000.0 / np.sqrt(2.0), rtol=1e-6)


@pytest.mark.gpu
def test_regrid_onto_a_file_defined_grid(oracle, gpu_lib, geo_file):
    from conftest import mesh_xyz
    from mpassit_amd import regrid as R, synth, target_grid as T
    path, g = geo_file
    t = T.define_target_grid_file(path)
    m = synth.regional_mesh_for_lambert(g.proj, 41, 31, 4000, margin=0.3)
    mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_target(t)
    cxyz, vxyz = mesh_xyz(oracle, m)
    tri, _ = oracle.dual_triangles(m.verticesOnCell, m.nVertices, cxyz)
    f = synth.analytic_field(m.latCell, m.lonCell, 3)
    rh = R.regrid_store(mesh, grid, R.REGRIDMETHOD_BILINEAR)
    # round 5: the file's MAP_PROJ / TRUELAT / STAND_LON / DX with its own first mass point as the known point is a claim the library
    # checked on the grid's points (float32 coordinates and all): the bilinear Store searched through the inverse projection
    assert t.proj.knowni == 1.0 and t.proj.lat1 == t.lat[0, 0] and rh.store_path == 1
    idx, w = oracle.bilinear_weights(cxyz, tri, oracle.lonlat_deg_to_xyz(t.lon, t.lat))
    np.testing.assert_allclose(rh.regrid(f, nlev=3).reshape(3, -1), oracle.apply_fixed(idx, w, f, 3), rtol=1e-11, atol=1e-11)
    rh.release()
    # conservative onto the corners the reference would build (overlapping / gapped quads and all)
    rc = R.regrid_store(mesh, grid, R.REGRIDMETHOD_CONSERVE)
    # ... while the CORNER stagger of a file-defined grid is get_cell_corners' (a cell east of the projection's, model_grid.F90:1902-1972):
    # it failed the same check, and the conservative Store kept the pyramid -- index boxes around the projection's cells would miss these
    assert rc.store_path == 0
    rp, col, val = oracle.conserve(m.verticesOnCell, vxyz, t.nx, t.ny, oracle.lonlat_deg_to_xyz(t.lon_c, t.lat_c))
    snow = synth.snow_field(m.latCell, m.lonCell) + 1.0
    np.testing.assert_allclose(rc.regrid(snow, nlev=1).reshape(-1), oracle.apply_csr(rp, col, val, snow, 1)[0], rtol=1e-10, atol=1e-12)
    rc.release()
    mesh.destroy()
    grid.destroy()


def _write_geo(path, g, map_proj, nml):
    from mpassit_amd import ncio
    with ncio.Writer(path, format=2) as w:
        for name, n in (("Time", None), ("west_east", g.nx), ("south_north", g.ny), ("west_east_stag", g.nx + 1), ("south_north_stag", g.ny + 1)):
            w.def_dim(name, n)
        for k, v in (("DX", np.float32(nml.get("dx", 0.0))), ("DY", np.float32(nml.get("dy", 0.0))), ("CEN_LAT", np.float32(nml["ref_lat"])),
                     ("CEN_LON", np.float32(nml["ref_lon"])), ("TRUELAT1", np.float32(nml.get("truelat1", 0.0))), ("TRUELAT2", np.float32(nml.get("truelat2", nml.get("truelat1", 0.0)))),
                     ("MOAD_CEN_LAT", np.float32(nml["ref_lat"])), ("STAND_LON", np.float32(nml.get("stand_lon", 0.0))), ("POLE_LAT", np.float32(90.0)),
                     ("POLE_LON", np.float32(0.0)), ("MAP_PROJ", map_proj)):
            w.put_att(k, v)
        dims = {"M": ("Time", "south_north", "west_east"), "U": ("Time", "south_north", "west_east_stag"), "V": ("Time", "south_north_stag", "west_east")}
        data = {"XLAT_M": (g.lat, "M"), "XLONG_M": (g.lon, "M"), "XLAT_U": (g.lat_u, "U"), "XLONG_U": (g.lon_u, "U"), "XLAT_V": (g.lat_v, "V"),
                "XLONG_V": (g.lon_v, "V"), "MAPFAC_M": (np.ones_like(g.lat), "M"), "MAPFAC_U": (np.ones_like(g.lat_u), "U"),
                "MAPFAC_V": (np.ones_like(g.lat_v), "V"), "HGT_M": (np.zeros_like(g.lat), "M")}
        for name, (_, st) in data.items():
            w.def_var(name, ncio.DOUBLE, dims[st])
        for name, (a, _) in data.items():
            w.put(name, a, rec=0)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,map_proj", [("polar", 2), ("mercator", 3), ("lat-lon", 6)])
def test_file_grids_of_the_other_projections_claim_their_inverse(oracle, gpu_lib, tmp_path, kind, map_proj):
    """Round 5: a WRF-style file of a polar stereographic (MAP_PROJ 2), Mercator (3) or lat-lon (6) grid: the file's projection attributes
    with the grid's own first mass point as the known point are a claim the library checks; the bilinear and nearest Stores then search
    through the inverse projection (same weights as the oracle's), the conservative Store keeps the pyramid (the CORNER stagger is
    get_cell_corners', not the projection's)."""
    from conftest import mesh_xyz
    from _parity_helpers import assert_fixed_weights_equal, assert_nearest_equal
    from mpassit_amd import regrid as R, synth, target_grid as T
    nml = {"polar": dict(dx=40000.0, dy=40000.0, ref_lat=80.0, ref_lon=-30.0, truelat1=70.0, stand_lon=-45.0),
           "mercator": dict(dx=50000.0, dy=50000.0, ref_lat=10.0, ref_lon=120.0, truelat1=20.0, stand_lon=120.0),
           "lat-lon": dict(dx=0.5, dy=0.5, ref_lat=-20.0, ref_lon=100.0, ref_x=1.0, ref_y=1.0, stand_lon=0.0)}[kind]
    g = T.define_target_grid_params(kind, 81, 61, **nml)
    path = tmp_path / ("geo_%d.nc" % map_proj)
    _write_geo(path, g, map_proj, dict(nml, dx=nml["dx"] if kind != "lat-lon" else 55000.0))
    t = T.define_target_grid_file(path)
    assert t.proj.code == {2: T.PROJ_PS, 3: T.PROJ_MERC, 6: T.PROJ_LATLON}[map_proj] and t.proj.knowni == 1.0
    m = synth.icosahedral_mesh(6)
    mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_target(t)
    cxyz, _ = mesh_xyz(oracle, m)
    pxyz = oracle.lonlat_deg_to_xyz(t.lon, t.lat)
    rh = R.regrid_store(mesh, grid, R.REGRIDMETHOD_BILINEAR)
    assert rh.store_path == 1, kind
    tri, _ = oracle.dual_triangles(m.verticesOnCell, m.nVertices, cxyz)
    idx_o, w_o = oracle.bilinear_weights(cxyz, tri, pxyz)
    assert_fixed_weights_equal(idx_o, w_o, *rh.weights(), tol=1e-11)
    rh.release()
    rn = R.regrid_store(mesh, grid, R.REGRIDMETHOD_NEAREST_STOD)
    assert rn.store_path >= 1, kind
    assert_nearest_equal(oracle.nearest(cxyz, pxyz), rn.weights()[0][:, 0], pxyz, cxyz, max_ties=2)
    rn.release()
    rc = R.regrid_store(mesh, grid, R.REGRIDMETHOD_CONSERVE)
    assert rc.store_path == 0
    rc.release()
    mesh.destroy()
    grid.destroy()
