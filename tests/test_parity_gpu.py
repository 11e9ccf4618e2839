"""GPU parity: every HIP kernel family, called through the C-ABI, against the CPU oracle on the same
seeded inputs.  Tolerances: bit-exact for indices / nearest-neighbour copies / wind rotation;
<= 1e-6 relative (BASELINE.json north_star) for bilinear and conservative values -- in practice the
two implementations agree to ~1e-12 and the tests assert that tighter bound where it holds."""
import numpy as np
import pytest

from _parity_helpers import assert_csr_equal, assert_fixed_weights_equal, assert_nearest_equal
from conftest import mesh_xyz

pytestmark = pytest.mark.gpu

RTOL = 1e-6  # north_star tolerance for float bilinear / conservative


def rel_err(a, b):
    scale = max(np.abs(b).max(), 1e-300)
    return np.abs(a - b).max() / scale


@pytest.fixture(scope="module")
def glob(gpu_lib, oracle, global_mesh, conus_grid_30km):
    from mpassit_amd import regrid as R
    m, g = global_mesh, conus_grid_30km
    mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_target(g)
    cxyz, vxyz = mesh_xyz(oracle, m)
    yield dict(m=m, g=g, mesh=mesh, grid=grid, cxyz=cxyz, vxyz=vxyz)
    mesh.destroy()
    grid.destroy()


@pytest.fixture(scope="module")
def regi(gpu_lib, oracle, regional_case):
    from mpassit_amd import regrid as R
    m, g = regional_case
    mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_target(g)
    cxyz, vxyz = mesh_xyz(oracle, m)
    yield dict(m=m, g=g, mesh=mesh, grid=grid, cxyz=cxyz, vxyz=vxyz)
    mesh.destroy()
    grid.destroy()


@pytest.mark.parametrize("case", ["glob", "regi"])
def test_dual_triangles_identical(case, request, oracle):
    c = request.getfixturevalue(case)
    tri_o, nvalid = oracle.dual_triangles(c["m"].verticesOnCell, c["m"].nVertices, c["cxyz"])
    tri_g = c["mesh"].triangles()
    assert np.array_equal(tri_g, tri_o)
    assert nvalid == (tri_g[:, 0] >= 0).sum()


@pytest.mark.parametrize("case", ["glob", "regi"])
def test_bilinear_store_and_regrid(case, request, oracle):
    from mpassit_amd import regrid as R, synth
    c = request.getfixturevalue(case)
    m, g = c["m"], c["g"]
    tri_o, _ = oracle.dual_triangles(m.verticesOnCell, m.nVertices, c["cxyz"])
    pxyz = oracle.lonlat_deg_to_xyz(g.lon, g.lat)
    idx_o, w_o = oracle.bilinear_weights(c["cxyz"], tri_o, pxyz)
    rh = R.regrid_store(c["mesh"], c["grid"], R.REGRIDMETHOD_BILINEAR)
    idx_g, w_g = rh.weights()
    mapped_o, mapped_g = idx_o[:, 0] >= 0, idx_g[:, 0] >= 0
    # mapped mask identical (points within 1e-9 of the hull edge would be implementation-defined; none here)
    assert np.array_equal(mapped_o, mapped_g)
    if case == "regi":
        assert (~mapped_g).sum() > 0, "regional case must exercise unmapped points"
    # identical rows within 1e-11; any other point is examined and must be a shared-edge tie (both triangles give
    # the same sparse weight vector)
    n_ties = assert_fixed_weights_equal(idx_o, w_o, idx_g, w_g, tol=1e-11)
    assert n_ties <= 0.001 * idx_o.shape[0]
    assert np.abs(w_g[mapped_g].sum(1) - 1).max() < 1e-13
    nlev = 5
    src = synth.analytic_field(m.latCell, m.lonCell, nlev)
    want = oracle.apply_fixed(idx_o, w_o, src, nlev)
    got = rh.regrid(src, nlev=nlev).reshape(nlev, -1)
    assert rel_err(got, want) < 1e-12 < RTOL
    assert (got[:, ~mapped_g] == 0.0).all()  # unmapped => exactly 0.0 (zero-filled destination)
    # level-fastest (MPAS file order) source gives the same answer
    got_lf = rh.regrid(np.ascontiguousarray(src.T), nlev=nlev, layout=R.LAYOUT_LEV_FAST).reshape(nlev, -1)
    assert rel_err(got_lf, want) < 1e-12
    # bundle of 3 fields == 3 single regrids
    src3 = np.stack([src, 2 * src + 1, -src])
    got3 = rh.regrid(src3, nlev=nlev, nfields=3).reshape(3, nlev, -1)
    assert np.array_equal(got3[0], got) and rel_err(got3[1], 2 * want + mapped_g * 1.0) < 1e-12
    # handle cache: same 5-tuple => same handle (reference recomputes 13x)
    rh2 = R.regrid_store(c["mesh"], c["grid"], R.REGRIDMETHOD_BILINEAR)
    assert rh2._h.value == rh._h.value
    rh2.release()
    rh.release()


@pytest.mark.parametrize("case", ["glob", "regi"])
def test_nearest_bit_exact(case, request, oracle):
    from mpassit_amd import regrid as R, synth
    c = request.getfixturevalue(case)
    m, g = c["m"], c["g"]
    pxyz = oracle.lonlat_deg_to_xyz(g.lon, g.lat)
    idx_o = oracle.nearest(c["cxyz"], pxyz)
    rh = R.regrid_store(c["mesh"], c["grid"], R.REGRIDMETHOD_NEAREST_STOD)
    idx_g, _ = rh.weights()
    idx_g = idx_g[:, 0]
    assert (idx_g >= 0).all()  # every destination point is mapped, also outside the mesh footprint
    assert_nearest_equal(idx_o, idx_g, pxyz, c["cxyz"], max_ties=1)  # only ties within rounding may differ
    cat = synth.category_field(m.nCells, nlev=4)
    got = rh.regrid(cat, nlev=4).reshape(4, -1)
    want = oracle.apply_nearest(idx_g, cat, 4)
    assert np.array_equal(got, want)  # bit-exact copy
    assert np.array_equal(got, np.floor(got))
    rh.release()


@pytest.mark.parametrize("case", ["glob", "regi"])
def test_conservative(case, request, oracle):
    from mpassit_amd import regrid as R, synth
    c = request.getfixturevalue(case)
    m, g = c["m"], c["g"]
    pc = oracle.lonlat_deg_to_xyz(g.lon_c, g.lat_c)
    rp_o, col_o, val_o = oracle.conserve(m.verticesOnCell, c["vxyz"], g.nx, g.ny, pc)
    rh = R.regrid_store(c["mesh"], c["grid"], R.REGRIDMETHOD_CONSERVE)
    rp_g, col_g, val_g = rh.csr()
    assert (val_g > 0).all()
    P = g.nx * g.ny
    rows_g = np.repeat(np.arange(P), np.diff(rp_g))
    rowsum = np.bincount(rows_g, weights=val_g, minlength=P)
    assert rowsum.max() < 1 + 1e-10
    if case == "glob":
        assert np.abs(rowsum - 1).max() < 1e-10  # fully covered destination cells
    else:
        assert (rowsum < 0.5).sum() > 0  # uncovered / partially covered rim cells exist
    # the two matrices as sets of (row, col) entries, unconditionally: common entries within 1e-11, entries on one
    # side only must be slivers at the drop threshold (1e-14 of the destination cell)
    n_common, only_o, only_g = assert_csr_equal(rp_o, col_o, val_o, rp_g, col_g, val_g, m.nCells)
    assert n_common > 0.999 * max(col_o.size, col_g.size)
    snow = synth.snow_field(m.latCell, m.lonCell)
    want = oracle.apply_csr(rp_o, col_o, val_o, snow, 1)
    got = rh.regrid(snow, nlev=1).reshape(1, -1)
    assert rel_err(got, want) < 1e-11 < RTOL
    const = np.full((1, m.nCells), 3.25)
    gotc = rh.regrid(const, nlev=1).reshape(-1)
    full = rowsum > 1 - 1e-10
    assert np.abs(gotc[full] - 3.25).max() < 1e-9
    rh.release()


def test_grid_to_grid_destagger(glob, oracle):
    from mpassit_amd import regrid as R
    g = glob["g"]
    cxyz = oracle.lonlat_deg_to_xyz(g.lon, g.lat)
    rng = np.random.default_rng(3)
    umass = rng.standard_normal((3, g.ny, g.nx))
    for stagger_o, stagger_g, lon, lat in ((1, R.STAGGERLOC_EDGE1, g.lon_u, g.lat_u), (2, R.STAGGERLOC_EDGE2, g.lon_v, g.lat_v)):
        dxyz = oracle.lonlat_deg_to_xyz(lon, lat)
        idx_o, w_o = oracle.grid_bilinear(g.nx, g.ny, cxyz, stagger_o, dxyz)
        rh = R.regrid_store_grid(glob["grid"], stagger_g)
        idx_g, w_g = rh.weights()
        mo, mg = idx_o[:, 0] >= 0, idx_g[:, 0] >= 0
        # the outer half-cell ring is outside the hull of the centres => unmapped (App. A4)
        ring = np.zeros(lon.shape, bool)
        if stagger_o == 1:
            ring[:, 0] = ring[:, -1] = True
        else:
            ring[0, :] = ring[-1, :] = True
        assert not mg.reshape(lon.shape)[ring].any()
        # Every point on which the mapped masks differ is examined: it must sit on the rim of the stagger array
        # (first/last row for U, first/last column for V), where the stagger point lies within rounding of the hull
        # edge of the centres (App. A4: implementation-defined); nowhere else may the two disagree.
        dis = (mo != mg).reshape(lon.shape)
        rim = np.zeros(lon.shape, bool)
        if stagger_o == 1:
            rim[0, :] = rim[-1, :] = True
        else:
            rim[:, 0] = rim[:, -1] = True
        assert not (dis & ~rim).any(), "mapped masks differ away from the hull edge at %s" % np.argwhere(dis & ~rim)[:5]
        assert dis.sum() <= rim.sum()
        want = oracle.apply_fixed(idx_o, w_o, umass.reshape(3, -1), 3)
        got = rh.regrid(umass.reshape(3, -1), nlev=3).reshape(3, -1)
        both = mo & mg
        assert np.abs(got[:, both] - want[:, both]).max() < 1e-9
        # interior stagger values are (within the sphere-vs-plane offset) the 2-point average
        interior = both.reshape(lon.shape).copy()
        interior[0, :] = interior[-1, :] = False
        interior[:, 0] = interior[:, -1] = False
        if stagger_o == 1:
            avg = 0.5 * (umass[:, :, :-1] + umass[:, :, 1:])
            sel = interior[:, 1:-1]
            assert np.abs(got.reshape(3, *lon.shape)[:, :, 1:-1][:, sel] - avg[:, sel]).max() < 1e-3
        rh.release()


def test_rotate_winds_bit_exact(gpu_lib, oracle, conus_grid_30km):
    from mpassit_amd import regrid as R
    g = conus_grid_30km
    rng = np.random.default_rng(5)
    u, v = rng.standard_normal((4, g.ny, g.nx)), rng.standard_normal((4, g.ny, g.nx))
    uo, vo = oracle.rotate_winds(g.cosa, g.sina, u, v)
    ug, vg = u.copy(), v.copy()
    R.rotate_winds_cgrid(g.cosa, g.sina, ug, vg)
    assert np.array_equal(ug, uo) and np.array_equal(vg, vo)
    # algebraic identity of interp.F90:741-742
    assert np.abs(ug - (u * g.cosa + v * g.sina)).max() < 1e-12
    assert np.abs(vg - (v * g.cosa - u * g.sina)).max() < 1e-12


def test_localize_halo(regi, oracle):
    """Compact-source (halo) form used by the multi-GPU path gives the same result."""
    from mpassit_amd import regrid as R, synth
    m = regi["m"]
    rh = R.regrid_store(regi["mesh"], regi["grid"], R.REGRIDMETHOD_BILINEAR)
    src = synth.analytic_field(m.latCell, m.lonCell, 3)
    full = rh.regrid(src, nlev=3)
    idx, _ = rh.weights()
    ids = rh.localize()
    assert np.array_equal(ids, np.unique(idx[idx >= 0]))
    assert rh.n_src == ids.size
    compact = rh.regrid(np.ascontiguousarray(src[:, ids]), nlev=3)
    assert np.array_equal(compact, full)
    rh.release()


def test_errors_are_loud(gpu_lib):
    from mpassit_amd import regrid as R
    from mpassit_amd._lib import MpgError
    with pytest.raises(MpgError):
        R.Mesh(np.zeros(4), np.zeros(4), np.zeros(4), np.zeros(4), np.zeros((4, 2), np.int32))  # maxEdges < 3


def test_global_latlon_target(gpu_lib, oracle, global_mesh):
    """BASELINE config 5 in miniature: global mesh -> global lat-lon grid (is_regional=.false.), poles and the
    date line included; bilinear, nearest and conservative against the oracle, plus global conservation."""
    from mpassit_amd import regrid as R, synth, target_grid as tg
    m = global_mesh
    g = tg.define_target_grid_params("lat-lon", 73, 37, stand_lon=-180.0, is_regional=False)   # 72 x 36, 5 degrees
    assert g.lat_c[0, 0] == -90.0 and g.lat_c[-1, 0] == 90.0
    mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_target(g)
    cxyz, vxyz = mesh_xyz(oracle, m)
    tri, _ = oracle.dual_triangles(m.verticesOnCell, m.nVertices, cxyz)
    pxyz = oracle.lonlat_deg_to_xyz(g.lon, g.lat)
    src = synth.analytic_field(m.latCell, m.lonCell, 3)
    # bilinear: the closed sphere has no hull -> every point mapped
    idx_o, w_o = oracle.bilinear_weights(cxyz, tri, pxyz)
    rh = R.regrid_store(mesh, grid, R.REGRIDMETHOD_BILINEAR)
    idx_g, w_g = rh.weights()
    assert (idx_g >= 0).all() and (idx_o >= 0).all()
    assert rel_err(rh.regrid(src, nlev=3).reshape(3, -1), oracle.apply_fixed(idx_o, w_o, src, 3)) < 1e-12
    rh.release()
    rh = R.regrid_store(mesh, grid, R.REGRIDMETHOD_NEAREST_STOD)
    assert np.array_equal(rh.weights()[0][:, 0], oracle.nearest(cxyz, pxyz))
    rh.release()
    # conservative: 5-degree cells hold ~dozens of source cells each; polar cells degenerate to triangles
    pc = oracle.lonlat_deg_to_xyz(g.lon_c, g.lat_c)
    rp_o, col_o, val_o = oracle.conserve(m.verticesOnCell, vxyz, g.nx, g.ny, pc)
    rh = R.regrid_store(mesh, grid, R.REGRIDMETHOD_CONSERVE)
    rp_g, col_g, val_g = rh.csr()
    P = g.nx * g.ny
    rows = np.repeat(np.arange(P), np.diff(rp_g))
    assert np.abs(np.bincount(rows, weights=val_g, minlength=P) - 1).max() < 1e-9
    snow = synth.snow_field(m.latCell, m.lonCell)
    got = rh.regrid(snow, nlev=1).reshape(-1)
    assert rel_err(got, oracle.apply_csr(rp_o, col_o, val_o, snow, 1)[0]) < 1e-10
    # global -> global: sum_j A_j w_ij = A_i for every source cell, hence the global integral is conserved
    lat_e = np.deg2rad(g.lat_c[:, 0])
    A_dst = np.repeat(np.deg2rad(5.0) * (np.sin(lat_e[1:]) - np.sin(lat_e[:-1])), g.nx)
    assert abs(A_dst.sum() - 4 * np.pi) < 1e-12
    A_src = np.bincount(col_g, weights=val_g * A_dst[rows], minlength=m.nCells)
    assert abs(A_src.sum() - 4 * np.pi) < 1e-9
    assert abs((got * A_dst).sum() - (snow[0] * A_src).sum()) < 1e-12
    rh.release()
    mesh.destroy()
    grid.destroy()


@pytest.mark.parametrize("case", ["glob", "regi"])
def test_node_located_bilinear(case, request, oracle):
    """vorticity path (interp.F90:350-366): MESHLOC_NODE bilinear with the documented fan triangulation."""
    from mpassit_amd import regrid as R
    c = request.getfixturevalue(case)
    m, g = c["m"], c["g"]
    ftri, _ = oracle.fan_triangles(m.verticesOnCell, c["vxyz"])
    pxyz = oracle.lonlat_deg_to_xyz(g.lon, g.lat)
    idx_o, w_o = oracle.bilinear_weights(c["vxyz"], ftri, pxyz)
    rh = R.regrid_store(c["mesh"], c["grid"], R.REGRIDMETHOD_BILINEAR, meshloc=R.MESHLOC_NODE)
    assert rh.n_src == m.nVertices
    idx_g, w_g = rh.weights()
    n_ties = assert_fixed_weights_equal(idx_o, w_o, idx_g, w_g, tol=1e-11)
    assert n_ties <= 0.001 * idx_o.shape[0]
    vort = np.random.default_rng(9).standard_normal((4, m.nVertices))
    got = rh.regrid(vort, nlev=4).reshape(4, -1)
    assert rel_err(got, oracle.apply_fixed(idx_o, w_o, vort, 4)) < 1e-12
    rh.release()
