"""Randomised end-to-end parity: small random meshes (global Voronoi, variable resolution, icosahedral, regional hex),
random target grids (Lambert / regional lat-lon / global lat-lon, odd sizes down to a few points), random level counts,
both source layouts and all three methods, GPU (through the C-ABI, default kernel selection) against the oracle."""
import numpy as np
import pytest

from conftest import mesh_xyz

pytestmark = pytest.mark.gpu

CASES = list(range(40))       # 32 .. 39 (round 5): polar stereographic and Mercator grids


def _mesh(rng, kind):
    from mpassit_amd import synth, target_grid as T
    if kind == 0:
        return synth.global_voronoi_mesh(int(rng.integers(300, 2500)), seed=int(rng.integers(1 << 30)))
    if kind == 1:
        return synth.variable_resolution_mesh(int(rng.integers(800, 2500)), seed=int(rng.integers(1 << 30)))
    if kind == 2:
        return synth.icosahedral_mesh(int(rng.integers(2, 5)), order=["morton", "native"][int(rng.integers(2))])
    p = T.define_target_grid_params("lambert", 41, 31, dx=60000.0, dy=60000.0, ref_lat=38.5, ref_lon=-97.5, truelat1=38.5,
                                    truelat2=38.5, stand_lon=-97.5, arrays=False).proj
    return synth.regional_mesh_for_lambert(p, 41, 31, int(rng.integers(500, 4000)), margin=float(rng.uniform(-0.1, 0.2)))


def _grid(rng, kind=None):
    from mpassit_amd import target_grid as T
    kind = int(rng.integers(3)) if kind is None else kind
    nx, ny = int(rng.integers(3, 140)), int(rng.integers(3, 90))
    if kind == 3:      # polar stereographic, the pole in or near the grid (program_setup.F90:179-182)
        dx = float(rng.uniform(20e3, 90e3))
        sgn = -1.0 if rng.random() < 0.4 else 1.0
        return T.define_target_grid_params("polar", nx, ny, dx=dx, dy=dx, ref_lat=sgn * float(rng.uniform(60, 90)), ref_lon=float(rng.uniform(-180, 180)),
                                           truelat1=sgn * float(rng.uniform(50, 90)), stand_lon=float(rng.uniform(-180, 180)))
    if kind == 4:      # Mercator (program_setup.F90:174-177)
        dx = float(rng.uniform(20e3, 90e3))
        return T.define_target_grid_params("mercator", nx, ny, dx=dx, dy=dx, ref_lat=float(rng.uniform(-45, 45)), ref_lon=float(rng.uniform(-180, 180)),
                                           truelat1=float(rng.uniform(-40, 40)), stand_lon=0.0)
    if kind == 0:
        dx = float(rng.uniform(20e3, 90e3))
        return T.define_target_grid_params("lambert", nx, ny, dx=dx, dy=dx, ref_lat=float(rng.uniform(25, 50)), ref_lon=float(rng.uniform(-120, -75)),
                                           truelat1=float(rng.uniform(25, 45)), truelat2=float(rng.uniform(30, 60)), stand_lon=-97.5)
    if kind == 1:
        d = float(rng.uniform(0.2, 1.0))
        return T.define_target_grid_params("lat-lon", nx, ny, dx=d, dy=d, ref_lat=float(rng.uniform(-40, 30)), ref_lon=float(rng.uniform(-170, 150)),
                                           ref_x=1.0, ref_y=1.0, stand_lon=0.0)
    return T.define_target_grid_params("lat-lon", nx, ny, stand_lon=float(rng.uniform(-180, 180)), is_regional=False)


@pytest.mark.parametrize("case", CASES)
def test_random_configuration(oracle, gpu_lib, case):
    from mpassit_amd import regrid as R, synth
    rng = np.random.default_rng(1000 + case)
    m = _mesh(rng, case % 4)
    g = _grid(rng, kind=None if case < 32 else 3 + case % 2)
    nlev = int(rng.choice([1, 2, 3, 5, 8, 17, 33, 55, 70]))
    nf = int(rng.integers(1, 4))
    mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_target(g)
    cxyz, vxyz = mesh_xyz(oracle, m)
    pts = oracle.lonlat_deg_to_xyz(g.lon, g.lat)
    src = rng.normal(size=(nf, nlev, m.nCells)) * 10 + 250
    src_lf = np.ascontiguousarray(src.transpose(0, 2, 1))
    # bilinear
    tri, _ = oracle.dual_triangles(m.verticesOnCell, m.nVertices, cxyz)
    idx, w = oracle.bilinear_weights(cxyz, tri, pts)
    rh = R.regrid_store(mesh, grid, R.REGRIDMETHOD_BILINEAR)
    gi, _ = rh.weights()
    assert np.array_equal(gi[:, 0] < 0, idx[:, 0] < 0)
    want = np.stack([oracle.apply_fixed(idx, w, src[f], nlev) for f in range(nf)]).reshape(nf, nlev, g.ny, g.nx)
    got = rh.regrid(src.reshape(-1), nlev=nlev, nfields=nf)
    assert np.abs(got - want).max() <= 1e-10 * np.abs(want).max()
    got_lf = rh.regrid(src_lf.reshape(-1), nlev=nlev, nfields=nf, layout=R.LAYOUT_LEV_FAST)
    assert np.array_equal(got_lf, got)
    got32 = rh.regrid_typed_host(src.astype(np.float32), nlev=nlev, nfields=nf, out_dtype=np.float32)
    assert np.abs(got32 - want).max() <= 2e-6 * np.abs(want).max()
    rh.release()
    # nearest: bit-exact copy of the oracle's choice (distance ties between the two sin/cos implementations aside)
    rn = R.regrid_store(mesh, grid, R.REGRIDMETHOD_NEAREST_STOD)
    near_o = oracle.nearest(cxyz, pts)
    near_g = rn.weights()[0][:, 0]
    diff = near_o != near_g
    if diff.any():
        dg = ((cxyz[near_g[diff]] - pts[diff]) ** 2).sum(1)
        do = ((cxyz[near_o[diff]] - pts[diff]) ** 2).sum(1)
        assert np.abs(dg - do).max() <= 1e-14
    cat = synth.category_field(m.nCells, nlev=nlev)
    assert np.array_equal(rn.regrid(cat, nlev=nlev)[0].reshape(nlev, -1), cat[:, near_g])
    rn.release()
    # conservative
    rp, col, val = oracle.conserve(m.verticesOnCell, vxyz, g.nx, g.ny, oracle.lonlat_deg_to_xyz(g.lon_c, g.lat_c))
    rc = R.regrid_store(mesh, grid, R.REGRIDMETHOD_CONSERVE)
    snow = np.abs(src[0, :1]) + 1.0
    want_c = oracle.apply_csr(rp, col, val, snow, 1)[0]
    got_c = rc.regrid(snow, nlev=1).reshape(-1)
    assert np.abs(got_c - want_c).max() <= 1e-9 * max(1.0, np.abs(want_c).max())
    rc.release()
    mesh.destroy()
    grid.destroy()


def _polar_grid(rng):
    """Fine regional grids that reach the last degrees before a pole: lat-lon (the index boxes' pad follows the latitude, the
    walk takes over beyond 85 degrees) and Lambert close to its pole."""
    from mpassit_amd import target_grid as T
    if rng.random() < 0.6:
        d = float(rng.uniform(0.1, 0.4))
        ny = int(rng.integers(12, 40))
        top = float(rng.uniform(84.0, 89.8))
        lat0 = top - d * (ny - 1)
        if rng.random() < 0.5:
            lat0 = -top
        return T.define_target_grid_params("lat-lon", int(rng.integers(30, 120)), ny, dx=d, dy=d, ref_lat=lat0, ref_lon=float(rng.uniform(-180, 180)),
                                           ref_x=1.0, ref_y=1.0, stand_lon=0.0)
    dx = float(rng.uniform(15e3, 60e3))
    lat = float(rng.uniform(72, 86)) * (-1 if rng.random() < 0.3 else 1)
    t = float(rng.uniform(50, 80)) * (1 if lat > 0 else -1)
    stand = float(rng.uniform(-180, 180))
    return T.define_target_grid_params("lambert", int(rng.integers(20, 90)), int(rng.integers(15, 60)), dx=dx, dy=dx, ref_lat=lat,
                                       ref_lon=stand + float(rng.uniform(-60, 60)), truelat1=t, truelat2=t, stand_lon=stand)


@pytest.mark.parametrize("case", list(range(10)))
def test_random_polar_configuration(oracle, gpu_lib, case):
    """Global meshes under grids near a pole, the three weight sets against the oracle (mapped mask identical, weights to the
    parity tolerances, nearest index identical up to distance ties)."""
    from _parity_helpers import assert_csr_equal, conserve_tol
    from mpassit_amd import regrid as R
    rng = np.random.default_rng(7000 + case)
    m = _mesh(rng, case % 3)                       # global Voronoi / variable resolution / icosahedral
    g = _polar_grid(rng)
    mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_target(g)
    cxyz, vxyz = mesh_xyz(oracle, m)
    pts = oracle.lonlat_deg_to_xyz(g.lon, g.lat)
    tri, _ = oracle.dual_triangles(m.verticesOnCell, m.nVertices, cxyz)
    idx, w = oracle.bilinear_weights(cxyz, tri, pts)
    rh = R.regrid_store(mesh, grid, R.REGRIDMETHOD_BILINEAR)
    gi, gw = rh.weights()
    assert np.array_equal(gi[:, 0] < 0, idx[:, 0] < 0)
    src = rng.normal(size=(1, 3, m.nCells)) * 10 + 250
    want = oracle.apply_fixed(idx, w, src[0], 3).reshape(3, g.ny, g.nx)
    got = rh.regrid(src.reshape(-1), nlev=3, nfields=1)[0]
    assert np.abs(got - want).max() <= 1e-10 * np.abs(want).max()
    rh.release()
    rn = R.regrid_store(mesh, grid, R.REGRIDMETHOD_NEAREST_STOD)
    near_o, near_g = oracle.nearest(cxyz, pts), rn.weights()[0][:, 0]
    diff = near_o != near_g
    if diff.any():
        assert np.abs(((cxyz[near_g[diff]] - pts[diff]) ** 2).sum(1) - ((cxyz[near_o[diff]] - pts[diff]) ** 2).sum(1)).max() <= 1e-14
    rn.release()
    rp, col, val = oracle.conserve(m.verticesOnCell, vxyz, g.nx, g.ny, oracle.lonlat_deg_to_xyz(g.lon_c, g.lat_c))
    rc = R.regrid_store(mesh, grid, R.REGRIDMETHOD_CONSERVE)
    snow = np.abs(src[0, :1]) + 1.0
    want_c = oracle.apply_csr(rp, col, val, snow, 1)[0]
    got_c = rc.regrid(snow, nlev=1).reshape(-1)
    assert np.abs(got_c - want_c).max() <= max(1e-9, 10 * conserve_tol(oracle, g)) * max(1.0, np.abs(want_c).max())
    rc.release()
    mesh.destroy()
    grid.destroy()


@pytest.mark.parametrize("case", list(range(24)))
def test_random_row_block_on_a_cut_mesh(gpu_lib, case):
    """The same random meshes and grids, a random block of target rows: the mesh cut to that block (mpg_mesh_create_window), with
    the index-space searches the block grid's projection allows, must give the weights of the whole mesh on the pyramid / BVH
    searches bit for bit -- bilinear (element- and node-located), nearest, conservative."""
    from mpassit_amd import regrid as R
    rng = np.random.default_rng(5000 + case)
    m = _mesh(rng, case % 4)
    g = _grid(rng)
    j0 = int(rng.integers(0, max(1, g.ny - 2)))
    j1 = int(rng.integers(j0 + 1, g.ny + 1))
    res = []
    for cut in (False, True):
        gpu_lib.tune("store_boxes", 1 if cut else 0)
        try:
            grid = R.Grid.from_target(g, rows=(j0, j1), attach_proj=cut)
            mesh = R.Mesh.from_mpas(m, window_grid=grid if cut else None)
            out = []
            for kw in (dict(regridmethod=R.REGRIDMETHOD_BILINEAR), dict(regridmethod=R.REGRIDMETHOD_NEAREST_STOD),
                       dict(regridmethod=R.REGRIDMETHOD_CONSERVE), dict(regridmethod=R.REGRIDMETHOD_BILINEAR, meshloc=R.MESHLOC_NODE)):
                rh = R.regrid_store(mesh, grid, **kw)
                out.append(rh.csr() if kw["regridmethod"] == R.REGRIDMETHOD_CONSERVE else rh.weights())
                rh.release()
            res.append(out)
            mesh.destroy()
            grid.destroy()
        finally:
            gpu_lib.tune("store_boxes", 1)
    for a, b in zip(*res):
        for x, y in zip(a, b):
            assert np.array_equal(x, y), (case, j0, j1)
