"""mpg_mesh_create_window: a rank brings only the part of the mesh its row block of the target grid can see to the device
(where the reference hands every rank 1/N of the cells, model_grid.F90:423-438).  The bar: every RegridStore of the windowed
mesh onto its grid gives the weights of the whole mesh BIT FOR BIT -- bilinear (element- and node-located), nearest,
conservative -- on row blocks of a banded regional mesh, on a block that sticks out of the mesh (nearest falls back to all
centres), on a global mesh with scattered numbering (the window degrades to the whole mesh) and on a coarse mesh under a
fine grid (the margin has to grow before the closure check passes)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _weights(R, mesh, grid, with_node=True):
    out = {}
    for name, kw in (("bilinear", dict(regridmethod=R.REGRIDMETHOD_BILINEAR)), ("nearest", dict(regridmethod=R.REGRIDMETHOD_NEAREST_STOD)),
                     ("conserve", dict(regridmethod=R.REGRIDMETHOD_CONSERVE)),
                     ("node", dict(regridmethod=R.REGRIDMETHOD_BILINEAR, meshloc=R.MESHLOC_NODE))):
        if name == "node" and not with_node:
            continue
        rh = R.regrid_store(mesh, grid, **kw)
        out[name] = rh.csr() if name == "conserve" else rh.weights()
        out[name + "_n_src"] = rh.n_src
        rh.release()
    return out


def _same(a, b):
    assert a.keys() == b.keys()
    for k in a:
        if k.endswith("_n_src"):
            assert a[k] == b[k], k
        else:
            for x, y in zip(a[k], b[k]):
                assert np.array_equal(x, y), k


@pytest.mark.parametrize("rows", [(0, 23), (23, 46), (40, 90)])
def test_row_blocks_of_a_banded_regional_mesh(gpu_lib, regional_case, rows):
    from mpassit_amd import regrid as R
    m, g = regional_case
    whole = R.Mesh.from_mpas(m)
    grid = R.Grid.from_target(g, rows=rows)
    want = _weights(R, whole, grid)
    whole.destroy()
    mesh = R.Mesh.from_mpas(m, window_grid=grid)
    c0, cn, v0, vn, margin = mesh.window_info()
    assert 0 < cn < 0.75 * m.nCells and 0 < vn < 0.75 * m.nVertices and 0 < margin < 0.5     # a proper part of the mesh
    _same(_weights(R, mesh, grid), want)
    tri = mesh.triangles()
    assert (tri[:v0] == -1).all() and (tri[v0 + vn:] == -1).all() and (tri[v0:v0 + vn] >= 0).any()
    # the mesh serves the grid it was cut to and no other
    other = R.Grid.from_target(g, rows=(rows[0], rows[1] - 1))
    with pytest.raises(gpu_lib.MpgError, match="cut to another grid"):
        R.regrid_store(mesh, other, R.REGRIDMETHOD_BILINEAR)
    other.destroy()
    mesh.destroy()
    grid.destroy()


def test_windowed_fields_regrid_like_the_whole_mesh(gpu_lib, regional_case):
    """Source windows (which cells of a FIELD a rank holds) on top of a geometry window: the Regrid of the slab equals the
    whole-mesh Regrid."""
    from mpassit_amd import regrid as R
    m, g = regional_case
    rows = (30, 60)
    grid = R.Grid.from_target(g, rows=rows)
    whole = R.Mesh.from_mpas(m)
    src = np.random.default_rng(5).normal(size=(4, m.nCells))
    rh = R.regrid_store(whole, grid, R.REGRIDMETHOD_BILINEAR)
    want = rh.regrid(src.reshape(-1), nlev=4)
    rh.release()
    whole.destroy()
    mesh = R.Mesh.from_mpas(m, window_grid=grid)
    rh = R.regrid_store(mesh, grid, R.REGRIDMETHOD_BILINEAR)
    a, b = rh.source_range()
    c0, cn, _, _, _ = mesh.window_info()
    assert c0 <= a < b <= c0 + cn                                  # what the handle references lies inside the resident rows
    mesh.set_source_window(a, b - a)
    rh._refresh()
    assert np.array_equal(rh.regrid(np.ascontiguousarray(src[:, a:b]).reshape(-1), nlev=4), want)
    rh.release()
    mesh.destroy()
    grid.destroy()


def test_grid_that_sticks_out_of_the_mesh_and_one_off_the_mesh(gpu_lib, regional_case):
    """The target of `regional_case` is larger than the mesh: its first rows lie outside the footprint.  Bilinear / conservative
    leave them unmapped either way; nearest must still find the true nearest cell (further than the window's margin for the
    outermost points: the search goes back to all centres)."""
    from mpassit_amd import regrid as R, target_grid as tg
    m, g = regional_case
    for rows in ((0, 6), (0, 30)):
        grid = R.Grid.from_target(g, rows=rows)
        whole = R.Mesh.from_mpas(m)
        want = _weights(R, whole, grid, with_node=False)
        whole.destroy()
        mesh = R.Mesh.from_mpas(m, window_grid=grid)
        _same(_weights(R, mesh, grid, with_node=False), want)
        mesh.destroy()
        grid.destroy()
    # a grid on the other side of the globe: no cell at all in the window
    far = tg.define_target_grid_params("lat-lon", 21, 11, dx=0.5, dy=0.5, ref_lat=-40.0, ref_lon=80.0, ref_x=1.0, ref_y=1.0, stand_lon=80.0)
    grid = R.Grid.from_target(far)
    whole = R.Mesh.from_mpas(m)
    want = _weights(R, whole, grid, with_node=False)
    whole.destroy()
    mesh = R.Mesh.from_mpas(m, window_grid=grid)
    assert mesh.window_info()[1] == 0
    got = _weights(R, mesh, grid, with_node=False)
    _same(got, want)
    assert (got["bilinear"][0] == -1).all() and got["conserve"][1].size == 0
    mesh.destroy()
    grid.destroy()


def test_scattered_numbering_and_a_coarse_mesh_under_a_fine_grid(gpu_lib, global_mesh, conus_grid_30km):
    from mpassit_amd import regrid as R, synth, target_grid as tg
    # a global Voronoi mesh (scipy numbering: no bands) under a regional grid: the covering id range is (nearly) everything
    grid = R.Grid.from_target(conus_grid_30km, rows=(10, 60))
    whole = R.Mesh.from_mpas(global_mesh)
    want = _weights(R, whole, grid)
    whole.destroy()
    mesh = R.Mesh.from_mpas(global_mesh, window_grid=grid)
    _same(_weights(R, mesh, grid), want)
    mesh.destroy()
    grid.destroy()
    # 480-km cells (642-cell icosahedral mesh, Morton-numbered) under a 10-km grid: every triangle is far larger than the margin the
    # first guess starts from
    coarse = synth.icosahedral_mesh(3)
    fine = tg.define_target_grid_params("lambert", 121, 81, dx=10000.0, dy=10000.0, ref_lat=38.5, ref_lon=-97.5, truelat1=38.5, truelat2=38.5,
                                        stand_lon=-97.5)
    grid = R.Grid.from_target(fine)
    whole = R.Mesh.from_mpas(coarse)
    want = _weights(R, whole, grid)
    whole.destroy()
    mesh = R.Mesh.from_mpas(coarse, window_grid=grid)
    _same(_weights(R, mesh, grid), want)
    assert (want["bilinear"][0] >= 0).all()                        # every point of the fine grid sits in a (huge) triangle
    mesh.destroy()
    grid.destroy()


@pytest.mark.parametrize("level", [0, 1, 2])
def test_a_small_grid_inside_one_cell_of_a_very_coarse_mesh(gpu_lib, level):
    """12 / 42 / 162 cells on the globe (7 000 / 3 700 / 1 900-km spacing) and a 30 x 20-point grid of 5-km cells placed between
    cell centres: every centre is further from the grid than any margin a window starts from -- the grid must not be taken to
    lie off the mesh (an empty window, empty weights) but be served by the cells around it, as on the whole mesh."""
    from mpassit_amd import regrid as R, synth, target_grid as tg
    m = synth.icosahedral_mesh(level)
    for ref_lat, ref_lon in ((31.7, -58.3), (-12.0, 77.0), (64.0, 10.0)):
        g = tg.define_target_grid_params("lambert", 31, 21, dx=5000.0, dy=5000.0, ref_lat=ref_lat, ref_lon=ref_lon, truelat1=ref_lat, truelat2=ref_lat,
                                         stand_lon=ref_lon)
        grid = R.Grid.from_target(g)
        whole = R.Mesh.from_mpas(m)
        want = _weights(R, whole, grid)
        whole.destroy()
        mesh = R.Mesh.from_mpas(m, window_grid=grid)
        _same(_weights(R, mesh, grid), want)
        assert (want["bilinear"][0] >= 0).all() and (want["nearest"][0] >= 0).all() and want["conserve"][1].size > 0
        mesh.destroy()
        grid.destroy()


def test_one_row_of_a_fine_polar_grid_under_a_coarse_mesh(gpu_lib):
    """Found by tools/fuzz_soak.py: one row of a 0.11-degree lat-lon grid at 83 N (points 1.5 km apart along the row) under a
    162-cell mesh (1 900-km spacing).  The window's first margin -- six grid lengths -- is four hundred times below the mesh's
    spacing; the margin must keep growing (to half a radian) until it finds cells instead of giving up after three steps and
    declaring the grid off the mesh (an empty window, all points unmapped)."""
    from mpassit_amd import regrid as R, synth, target_grid as tg
    for level, nx, ny, d, lat0 in ((2, 188, 27, 0.11, 81.36), (3, 48, 27, 0.117, 84.38)):
        m = synth.icosahedral_mesh(level)
        g = tg.define_target_grid_params("lat-lon", nx + 1, ny + 1, dx=d, dy=d, ref_lat=lat0, ref_lon=-40.0, ref_x=1.0, ref_y=1.0, stand_lon=0.0)
        for rows in ((14, 15), (ny - 1, ny)):
            grid = R.Grid.from_target(g, rows=rows)
            whole = R.Mesh.from_mpas(m)
            want = _weights(R, whole, grid)
            whole.destroy()
            mesh = R.Mesh.from_mpas(m, window_grid=grid)
            _same(_weights(R, mesh, grid), want)
            assert (want["bilinear"][0] >= 0).all() and want["conserve"][1].size > 0
            mesh.destroy()
            grid.destroy()


def _banded(m):
    """Cells AND vertices renumbered by latitude (then longitude): the spatially banded numbering windows rely on."""
    from mpassit_amd import synth
    pc = np.lexsort((m.lonCell, np.round(m.latCell, 3)))
    pv = np.lexsort((m.lonVertex, np.round(m.latVertex, 3)))
    inv = np.empty(m.nVertices + 1, np.int32)
    inv[0] = 0
    inv[pv + 1] = np.arange(1, m.nVertices + 1, dtype=np.int32)
    return synth.MpasMesh(m.latCell[pc], m.lonCell[pc], m.latVertex[pv], m.lonVertex[pv], inv[m.verticesOnCell[pc]])


@pytest.mark.parametrize("rows", [(0, 30), (30, 61), (61, 90)])
def test_variable_resolution_mesh_with_coarse_cells_beside_the_grid(gpu_lib, rows):
    """A 60-3 km style mesh (cells 16 times larger outside a cap than inside it), banded numbering, under a fine grid that
    straddles the cap's edge: the margin is sized from the FINE cells the grid lies on, while the rows' id range also holds
    coarse cells many margins wide.  The closure check measures such a cell's distance for real (it lies beyond the distance pass's
    cap but can reach the grid) -- the weights of the cut mesh are those of the whole mesh, bit for bit."""
    from mpassit_amd import regrid as R, synth, target_grid as tg
    m = _banded(synth.variable_resolution_mesh(14000, lat0_deg=40.0, lon0_deg=-100.0, ratio=16.0, radius_deg=6.0, seed=5))
    g = tg.define_target_grid_params("lambert", 151, 91, dx=12000.0, dy=12000.0, ref_lat=44.5, ref_lon=-100.0, truelat1=40.0, truelat2=40.0, stand_lon=-100.0)
    grid = R.Grid.from_target(g, rows=rows)
    whole = R.Mesh.from_mpas(m)
    want = _weights(R, whole, grid)
    whole.destroy()
    mesh = R.Mesh.from_mpas(m, window_grid=grid)
    c0, cn, v0, vn, margin = mesh.window_info()
    _same(_weights(R, mesh, grid), want)
    assert (want["bilinear"][0][:, 0] >= 0).all()                  # a global mesh: every point mapped
    print("rows", rows, "window cells", cn, "of", m.nCells, "margin", margin)
    mesh.destroy()
    grid.destroy()
