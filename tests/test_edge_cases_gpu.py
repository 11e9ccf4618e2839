"""Edge cases of the Regrid path: a target that misses the mesh entirely, the smallest grids, ragged level counts against the
kernels' level chunks (4 / 8 / 16), one-point rows, bad arguments.  All through the C-ABI, checked against the oracle."""
import numpy as np
import pytest

from conftest import mesh_xyz

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))


def test_target_that_misses_the_mesh(gpu_lib, oracle, regional_case):
    """Mesh over CONUS, target grid over the southern Indian Ocean: bilinear and conservative leave every point unmapped
    (zero-filled, ESMF_UNMAPPEDACTION_IGNORE + zeroregion TOTAL), nearest still maps every point (it always has a
    nearest source), and all kernel variants cope with tiles that reference no cell at all."""
    from mpassit_amd import regrid as R, synth, target_grid as T
    m, _ = regional_case
    far = T.define_target_grid_params("lat-lon", 41, 31, dx=0.25, dy=0.25, ref_lat=-40.0, ref_lon=80.0, ref_x=1.0, ref_y=1.0, stand_lon=80.0)
    mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_target(far)
    src = synth.analytic_field(m.latCell, m.lonCell, 6) + 5.0
    for method in (R.REGRIDMETHOD_BILINEAR, R.REGRIDMETHOD_CONSERVE):
        rh = R.regrid_store(mesh, grid, method)
        for layout, s in ((R.LAYOUT_CELL_FAST, src), (R.LAYOUT_LEV_FAST, np.ascontiguousarray(src.T))):
            out = rh.regrid(s, nlev=6, layout=layout)
            assert out.shape == (1, 6, far.ny, far.nx) and not out.any()
        assert not rh.regrid(src[0].astype(np.float32), nlev=1).any()
        rh.release()
    rh = R.regrid_store(mesh, grid, R.REGRIDMETHOD_NEAREST_STOD)
    cxyz, _ = mesh_xyz(oracle, m)
    want = oracle.nearest(cxyz, oracle.lonlat_deg_to_xyz(far.lon, far.lat))
    idx, _ = rh.weights()
    assert np.array_equal(idx[:, 0], want) and np.array_equal(rh.regrid(src[2], nlev=1).reshape(-1), src[2][want])
    rh.release()
    mesh.destroy()
    grid.destroy()


@pytest.mark.parametrize("nxy", [(2, 2), (2, 5), (65, 2), (3, 300)])
def test_smallest_and_thinnest_grids(gpu_lib, oracle, global_mesh, nxy):
    """Namelist nx, ny down to 2 (one mass point, two U points): tiles are mostly padding, rows shorter than a wavefront."""
    from mpassit_amd import regrid as R, synth, target_grid as T
    nx, ny = nxy
    g = T.define_target_grid_params("lat-lon", nx, ny, dx=0.5, dy=0.5, ref_lat=20.0, ref_lon=-100.0, ref_x=1.0, ref_y=1.0, stand_lon=-100.0)
    m = global_mesh
    cxyz, _ = mesh_xyz(oracle, m)
    tri, _ = oracle.dual_triangles(m.verticesOnCell, m.nVertices, cxyz)
    idx_o, w_o = oracle.bilinear_weights(cxyz, tri, oracle.lonlat_deg_to_xyz(g.lon, g.lat))
    mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_target(g)
    rh = R.regrid_store(mesh, grid, R.REGRIDMETHOD_BILINEAR)
    nlev = 7
    src = synth.analytic_field(m.latCell, m.lonCell, nlev)
    want = oracle.apply_fixed(idx_o, w_o, src, nlev)
    assert (idx_o[:, 0] >= 0).all()
    for layout, s in ((R.LAYOUT_CELL_FAST, src), (R.LAYOUT_LEV_FAST, np.ascontiguousarray(src.T))):
        got = rh.regrid(s, nlev=nlev, layout=layout).reshape(nlev, -1)
        assert got.shape[1] == (nx - 1) * (ny - 1) and _rel(got, want) < 1e-12
    # the staggered companions exist; a constant comes back as the constant where the point lies inside the hull of the
    # mass points and as exactly 0 elsewhere -- everywhere when a dimension has a single mass point (no bilinear cell)
    for stag, shape in ((R.STAGGERLOC_EDGE1, (ny - 1, nx)), (R.STAGGERLOC_EDGE2, (ny, nx - 1))):
        rs = R.regrid_store_grid(grid, stag)
        out = rs.regrid(np.full((2, ny - 1, nx - 1), 3.25).reshape(-1), nlev=2)
        assert out.shape == (1, 2) + shape
        mapped = out != 0.0
        assert np.abs(out[mapped] - 3.25).max(initial=0.0) < 1e-13 and np.array_equal(mapped[0, 0], mapped[0, 1])
        if min(nx - 1, ny - 1) < 2:
            assert not mapped.any()
        else:
            assert mapped.any()
        rs.release()
    rh.release()
    mesh.destroy()
    grid.destroy()


@pytest.mark.parametrize("nlev", [1, 2, 3, 5, 9, 17, 33, 64, 65, 129])
def test_ragged_level_counts(gpu_lib, oracle, regional_case, nlev):
    """Level counts that are not multiples of the kernels' level chunks, both layouts, float64 and float32 sources, and a
    two-field bundle: the last, partial chunk must neither read nor write past its levels."""
    import torch

    from mpassit_amd import regrid as R, synth
    m, g = regional_case
    cxyz, _ = mesh_xyz(oracle, m)
    tri, _ = oracle.dual_triangles(m.verticesOnCell, m.nVertices, cxyz)
    idx_o, w_o = oracle.bilinear_weights(cxyz, tri, oracle.lonlat_deg_to_xyz(g.lon, g.lat))
    mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_target(g)
    rh = R.regrid_store(mesh, grid, R.REGRIDMETHOD_BILINEAR)
    src = synth.analytic_field(m.latCell, m.lonCell, nlev)
    want = oracle.apply_fixed(idx_o, w_o, src, nlev)
    P = g.nx * g.ny
    for layout, s in ((R.LAYOUT_CELL_FAST, src), (R.LAYOUT_LEV_FAST, np.ascontiguousarray(src.T))):
        # guard band behind the destination: must stay untouched
        out = torch.full((2 * nlev * P + 4096,), 7.5, dtype=torch.float64, device="cuda")
        s2 = torch.as_tensor(np.stack([s, -2.0 * s]), device="cuda").reshape(-1)
        rh.regrid(s2, nlev=nlev, nfields=2, layout=layout, out=out[:2 * nlev * P].view(2, nlev, g.ny, g.nx))
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        assert (got[2 * nlev * P:] == 7.5).all()
        assert _rel(got[:nlev * P].reshape(nlev, P), want) < 1e-12 and _rel(got[nlev * P:2 * nlev * P].reshape(nlev, P), -2.0 * want) < 1e-12
        s32 = s.astype(np.float32)
        want32 = oracle.apply_fixed(idx_o, w_o, (s32 if layout == R.LAYOUT_CELL_FAST else s32.T).astype(np.float64), nlev)
        got32 = rh.regrid(s32, nlev=nlev, layout=layout).reshape(nlev, P)
        assert _rel(got32, want32) < 1e-12
        # float32 in, float32 out (the file-to-file path: linear 64-point tiles, two levels per lane; 64 / 65 / 129 levels
        # take more than one pass over the lanes), guard band behind the float32 destination
        o32 = torch.full((nlev * P + 4096,), 7.5, dtype=torch.float32, device="cuda")
        rh.regrid_typed(torch.as_tensor(s32, device="cuda").reshape(-1), nlev=nlev, layout=layout, out_dtype=torch.float32, offset=-1.5,
                        out=o32[:nlev * P].view(1, nlev, g.ny, g.nx))
        torch.cuda.synchronize()
        g32 = o32.cpu().numpy()
        assert (g32[nlev * P:] == 7.5).all()
        ref = (want32 - 1.5).astype(np.float32)
        ne = g32[:nlev * P].reshape(nlev, P) != ref
        assert ne.mean() < 1e-4 and (np.abs(g32[:nlev * P].reshape(nlev, P)[ne].astype(np.float64) - ref[ne]) <= np.spacing(np.abs(ref[ne]))).all()
    rh.release()
    mesh.destroy()
    grid.destroy()


def test_bad_arguments_are_refused(gpu_lib, regional_case):
    import ctypes as C

    from mpassit_amd import regrid as R
    from mpassit_amd._lib import MpgError
    m, g = regional_case
    mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_target(g)
    rh = R.regrid_store(mesh, grid, R.REGRIDMETHOD_BILINEAR)
    lib = gpu_lib.load()
    src = np.zeros(m.nCells)
    dst = np.zeros(g.nx * g.ny)
    p = lambda a: a.ctypes.data_as(C.c_void_p)          # noqa: E731
    assert lib.mpg_regrid(rh._h, p(src), C.c_int(0), C.c_int(0), C.c_int(1), p(dst)) == 2 and b"nlev" in lib.mpg_last_error()
    assert lib.mpg_regrid(rh._h, p(src), C.c_int(0), C.c_int(1), C.c_int(0), p(dst)) == 2
    assert lib.mpg_regrid(rh._h, p(src), C.c_int(7), C.c_int(1), C.c_int(1), p(dst)) == 2 and b"layout" in lib.mpg_last_error()
    assert lib.mpg_regrid(rh._h, None, C.c_int(0), C.c_int(1), C.c_int(1), p(dst)) == 2
    assert lib.mpg_regrid_typed(rh._h, p(src), C.c_int(4), C.c_int(0), C.c_int(1), C.c_int(1), p(dst), C.c_int(0), C.c_double(1.0),
                                C.c_double(0.0)) == 2
    with pytest.raises(ValueError):
        rh.regrid(np.zeros(m.nCells + 1), nlev=1)                           # wrong element count is caught before the call
    with pytest.raises(MpgError):
        R.regrid_store(mesh, grid, 17)                                      # unknown regrid method
    rh.release()
    mesh.destroy()
    grid.destroy()


def test_shared_handle_is_not_reindexed_in_place(gpu_lib, regional_case):
    """mpg_handle_localize / mpg_handle_rebase re-index a handle in place; a handle the Store cache has handed out twice
    (same 5-tuple -> same pointer, refcount 2) must be refused, or the other holder's next Regrid would read wrong cells."""
    from mpassit_amd import regrid as R
    from mpassit_amd._lib import MpgError
    m, g = regional_case
    mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_target(g)
    a = R.regrid_store(mesh, grid, R.REGRIDMETHOD_BILINEAR)
    b = R.regrid_store(mesh, grid, R.REGRIDMETHOD_BILINEAR)
    assert a._h.value == b._h.value
    with pytest.raises(MpgError, match="shared"):
        a.localize()
    with pytest.raises(MpgError, match="shared"):
        b.rebase(0, m.nCells)
    b.release()
    ids = a.localize()                                   # sole owner now: allowed
    assert a.n_src == ids.size
    a.release()
    mesh.destroy()
    grid.destroy()


def test_reinit_on_another_device_is_refused(gpu_lib):
    """mpg_init twice with the same device is a no-op; a different index while initialised is refused (streams and pinned
    staging belong to the first device)."""
    import ctypes as C
    lib = gpu_lib.load()
    assert lib.mpg_init(C.c_int(0)) == 0
    assert lib.mpg_init(C.c_int(1)) != 0                 # out of range on a one-GPU box, "already initialised" on a node


@pytest.mark.parametrize("level", [0, 1])
def test_conservative_store_of_a_very_coarse_mesh(gpu_lib, oracle, level):
    """12 / 42 source cells under a 300 x 250 grid: every source cell has tens of thousands of candidate destination cells, far
    beyond the fixed-size candidate lists -- all of them take the cooperative count / list passes (and outgrow the 2048-node
    breadth-first queue of those passes), every destination cell lies inside one to three source cells."""
    from _parity_helpers import assert_csr_equal, conserve_tol
    from mpassit_amd import regrid as R, synth, target_grid as T
    m = synth.icosahedral_mesh(level)
    g = T.define_target_grid_params("lat-lon", 301, 251, dx=0.2, dy=0.2, ref_lat=-25.0, ref_lon=100.0, ref_x=1.0, ref_y=1.0, stand_lon=100.0)
    cxyz, vxyz = mesh_xyz(oracle, m)
    rp_o, col_o, val_o = oracle.conserve(m.verticesOnCell, vxyz, g.nx, g.ny, oracle.lonlat_deg_to_xyz(g.lon_c, g.lat_c))
    mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_target(g)
    rh = R.regrid_store(mesh, grid, R.REGRIDMETHOD_CONSERVE)
    rp_g, col_g, val_g = rh.csr()
    n_common, only_o, only_g = assert_csr_equal(rp_o, col_o, val_o, rp_g, col_g, val_g, m.nCells, tol=conserve_tol(oracle, g))
    assert n_common >= g.nx * g.ny and n_common > 0.999 * max(col_o.size, col_g.size)
    P = g.nx * g.ny
    rowsum = np.bincount(np.repeat(np.arange(P), np.diff(rp_g)), weights=val_g, minlength=P)
    assert np.abs(rowsum - 1).max() < 1e-9                       # the mesh covers the sphere: every destination cell fully covered
    out = rh.regrid(np.arange(m.nCells, dtype=np.float64) + 1.0, nlev=1).reshape(-1)
    want = np.bincount(np.repeat(np.arange(P), np.diff(rp_o)), weights=val_o * (col_o + 1.0), minlength=P)
    assert _rel(out, want) < 1e-9
    rh.release()
    mesh.destroy()
    grid.destroy()


def test_vertices_on_cell_as_wide_as_mpas_writes_it(gpu_lib, regional_case):
    """MPAS files carry verticesOnCell ten columns wide (maxEdges) whatever the cells have -- six or seven vertices, the rest
    padded with 0.  The Stores size their per-cell work by the vertices present (the conservative clip its LDS polygons), not
    by the array's width: the same weights, bit for bit, from the array as the synthetic meshes hold it and from a padded one."""
    from mpassit_amd import regrid as R, synth
    m, g = regional_case
    wide = np.zeros((m.nCells, 10), np.int32)
    wide[:, :m.maxEdges] = m.verticesOnCell
    mw = synth.MpasMesh(m.latCell, m.lonCell, m.latVertex, m.lonVertex, wide)
    res = []
    for mm in (m, mw):
        mesh, grid = R.Mesh.from_mpas(mm), R.Grid.from_target(g)
        out = []
        for kw in (dict(regridmethod=R.REGRIDMETHOD_CONSERVE), dict(regridmethod=R.REGRIDMETHOD_BILINEAR),
                   dict(regridmethod=R.REGRIDMETHOD_BILINEAR, meshloc=R.MESHLOC_NODE)):
            rh = R.regrid_store(mesh, grid, **kw)
            out.append(rh.csr() if kw["regridmethod"] == R.REGRIDMETHOD_CONSERVE else rh.weights())
            rh.release()
        res.append(out)
        mesh.destroy()
        grid.destroy()
    for a, b in zip(*res):
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
    assert res[0][0][1].size > 0


@pytest.mark.parametrize("windowed", [False, True])
def test_a_corrupt_mesh_is_refused_at_creation(gpu_lib, regional_case, windowed):
    """What a damaged file hands over must be caught before any geometry kernel uses it as an index or an angle: a vertex number beyond
    nVertices (would be read as a coordinate index), a NaN coordinate, latitudes in degrees where radians are expected.  With and
    without a geometry window; the mesh as it is still builds afterwards."""
    from mpassit_amd import regrid as R
    from mpassit_amd._lib import MpgError
    m, g = regional_case
    grid = R.Grid.from_target(g)
    wg = grid if windowed else None

    def make(**over):
        a = dict(latCell=m.latCell, lonCell=m.lonCell, latVertex=m.latVertex, lonVertex=m.lonVertex, verticesOnCell=m.verticesOnCell)
        a.update(over)
        return R.Mesh(a["latCell"], a["lonCell"], a["latVertex"], a["lonVertex"], a["verticesOnCell"], window_grid=wg)

    mid = m.nCells // 2
    voc = m.verticesOnCell.copy()
    voc[mid, 1] = m.nVertices + 5
    with pytest.raises(MpgError, match="verticesOnCell refers to vertex %d of %d" % (m.nVertices + 5, m.nVertices)):
        make(verticesOnCell=voc)
    lat = m.latCell.copy()
    lat[mid] = np.nan
    with pytest.raises(MpgError, match="point %d has latitude nan.*RADIANS" % mid):
        make(latCell=lat)
    lon = m.lonVertex.copy()
    vmid = int(m.verticesOnCell[mid, 0]) - 1
    lon[vmid] = np.inf
    with pytest.raises(MpgError, match="%d has latitude .* longitude inf.*RADIANS" % vmid):
        make(lonVertex=lon)
    with pytest.raises(MpgError, match="RADIANS"):
        make(latCell=np.degrees(m.latCell), lonCell=np.degrees(m.lonCell))
    mesh = make()
    rh = R.regrid_store(mesh, grid, R.REGRIDMETHOD_BILINEAR)
    assert rh.nnz > 0
    rh.release()
    mesh.destroy()
    grid.destroy()


def test_a_grid_with_coordinates_that_are_no_angles_is_refused(gpu_lib, regional_case):
    from mpassit_amd import regrid as R
    from mpassit_amd._lib import MpgError
    _, g = regional_case
    lat = g.lat.copy()
    lat[3, 4] = np.nan
    with pytest.raises(MpgError, match="point %d has latitude nan.*DEGREES" % (3 * g.nx + 4)):
        R.Grid(g.lon, lat, g.lon_c, g.lat_c, g.lon_u, g.lat_u, g.lon_v, g.lat_v)
    lat_c = g.lat_c.copy()
    lat_c[0, 0] = 1e30
    with pytest.raises(MpgError, match="DEGREES"):
        R.Grid(g.lon, g.lat, g.lon_c, lat_c, g.lon_u, g.lat_u, g.lon_v, g.lat_v)
