"""O(1) candidate boxes of the Stores on projection-built grids (mpg_grid_create_proj): a source triangle's / polygon's target
points are found through the inverse projection instead of the box pyramid.  Same per-point / per-cell tests on a superset
of the candidates, so the owners, the weights and the conservative matrix must be IDENTICAL to the pyramid walk's
("store_boxes" 0) -- on Lambert and lat-lon grids, CENTER and staggered points, element- and node-located sources, a mesh
much coarser than the grid (every triangle exceeds the box limit and takes the walk), a global grid with its seam and the
caps poleward of 75 degrees (walk) -- and to the same grid built from coordinate ARRAYS, which has no inverse at all."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _stores(gpu_lib, R, mesh, grid, boxes, staggers=(0,), node=True, conserve=True):
    gpu_lib.tune("store_boxes", boxes)
    out = {}
    try:
        for st in staggers:
            rh = R.regrid_store(mesh, grid, R.REGRIDMETHOD_BILINEAR, staggerloc=st)
            out["bil%d" % st] = rh.weights()
            rh.release()
        if node:
            rh = R.regrid_store(mesh, grid, R.REGRIDMETHOD_BILINEAR, meshloc=R.MESHLOC_NODE)
            out["node"] = rh.weights()
            rh.release()
        if conserve:
            rh = R.regrid_store(mesh, grid, R.REGRIDMETHOD_CONSERVE)
            out["cons"] = rh.csr()
            rh.release()
    finally:
        gpu_lib.tune("store_boxes", 1)
    return out


def _same(a, b):
    assert a.keys() == b.keys()
    for k in a:
        for x, y in zip(a[k], b[k]):
            assert np.array_equal(x, y), k


def _case(gpu_lib, m, tgt, **kw):
    from mpassit_amd import regrid as R
    res = []
    for boxes in (1, 0):
        mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_proj(tgt, fill_target=False)      # fresh objects: nothing from the handle cache
        res.append(_stores(gpu_lib, R, mesh, grid, boxes, **kw))
        mesh.destroy()
        grid.destroy()
    _same(res[0], res[1])
    return res[0]


def test_lambert_grid_all_staggers_and_locations(gpu_lib):
    from mpassit_amd import workloads
    m, g, _, _ = workloads.workload("tiny")
    got = _case(gpu_lib, m, g, staggers=(0, 1, 2))
    assert (got["bil0"][0] >= 0).mean() > 0.9 and got["cons"][1].size > 0
    # the same grid from coordinate arrays (no inverse: pyramid only) holds the same CENTER weights
    from mpassit_amd import regrid as R
    mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_target(g)
    ref = _stores(gpu_lib, R, mesh, grid, 1, staggers=(0,), node=False)
    mesh.destroy()
    grid.destroy()
    assert np.array_equal(ref["bil0"][0], got["bil0"][0])          # owners; the weights agree to the last bits of the coordinates
    assert np.abs(ref["bil0"][1] - got["bil0"][1]).max() < 1e-9


def test_regional_mesh_smaller_than_the_grid(gpu_lib, regional_case):
    m, g = regional_case                                             # rim + unmapped strip: boxes clipped at the grid's edge
    got = _case(gpu_lib, m, g)
    assert (got["bil0"][0][:, 0] < 0).any()


def test_global_latlon_grid_seam_and_polar_caps(gpu_lib):
    from mpassit_amd import workloads
    m, g, _, _ = workloads.workload("c5_small")                      # 40 962 cells -> 360 x 180 global lat-lon, periodic, poles
    got = _case(gpu_lib, m, g, node=False)
    assert (got["bil0"][0] >= 0).all()


def test_mesh_much_coarser_than_the_grid(gpu_lib):
    from mpassit_amd import synth, target_grid as tg
    fine = tg.define_target_grid_params("lambert", 121, 81, dx=10000.0, dy=10000.0, ref_lat=38.5, ref_lon=-97.5, truelat1=38.5, truelat2=38.5,
                                        stand_lon=-97.5)
    got = _case(gpu_lib, synth.icosahedral_mesh(3), fine, node=False)   # 480-km cells: every triangle spans > 6 index units
    assert (got["bil0"][0] >= 0).all()
    got = _case(gpu_lib, synth.icosahedral_mesh(6), fine, node=False)   # 60-km cells: boxes of up to 8 x 8 points, some over the limit
    assert (got["bil0"][0] >= 0).all()


def test_southern_lambert_and_two_true_latitudes(gpu_lib, global_mesh):
    from mpassit_amd import target_grid as tg
    for kw in (dict(ref_lat=-35.0, ref_lon=140.0, truelat1=-30.0, truelat2=-60.0, stand_lon=140.0),
               dict(ref_lat=60.0, ref_lon=10.0, truelat1=30.0, truelat2=60.0, stand_lon=0.0)):
        g = tg.define_target_grid_params("lambert", 101, 91, dx=40000.0, dy=40000.0, **kw)
        got = _case(gpu_lib, global_mesh, g, node=False)
        assert (got["bil0"][0] >= 0).all()


def test_row_block_from_arrays_with_its_projection_attached(gpu_lib, regional_case):
    """mpg_grid_attach_proj: a rank's block of target rows, made from coordinate arrays, is told its projection and row offset --
    the Stores then search through the inverse.  Same weights as the pyramid search; a projection that does not fit the
    arrays (wrong row offset, wrong grid length) is refused and changes nothing."""
    from mpassit_amd import regrid as R, target_grid as tg
    m, g = regional_case
    rows = (17, 49)
    res = []
    for attach in (True, False):
        mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_target(g, rows=rows, attach_proj=attach)
        res.append(_stores(gpu_lib, R, mesh, grid, 1, staggers=(0, 1, 2)))
        mesh.destroy()
        grid.destroy()
    _same(res[0], res[1])
    grid = R.Grid.from_target(g, rows=rows, attach_proj=False)
    with pytest.raises(gpu_lib.MpgError, match="does not reproduce"):
        grid.attach_proj(g.proj, rows[0] + 1)                       # one row off
    other = tg.define_target_grid_params("lambert", g.nx + 1, g.ny + 1, dx=31000.0, dy=31000.0, ref_lat=38.5, ref_lon=-97.5, truelat1=38.5,
                                         truelat2=38.5, stand_lon=-97.5)
    with pytest.raises(gpu_lib.MpgError, match="does not reproduce"):
        grid.attach_proj(other.proj, rows[0])                       # another grid length
    grid.attach_proj(g.proj, rows[0])                               # the right one is accepted
    mesh = R.Mesh.from_mpas(m)
    _same(_stores(gpu_lib, R, mesh, grid, 1), {k: v for k, v in res[1].items() if k in ("bil0", "node", "cons")})
    mesh.destroy()
    grid.destroy()


def _nearest(gpu_lib, R, m, grid_fn, boxes, staggers=(0,), window=False, paths=None):
    gpu_lib.tune("store_boxes", boxes)
    try:
        grid = grid_fn()
        mesh = R.Mesh.from_mpas(m, window_grid=grid if window else None)
        out = []
        for st in staggers:
            rh = R.regrid_store(mesh, grid, R.REGRIDMETHOD_NEAREST_STOD, staggerloc=st)
            out.append(rh.weights()[0][:, 0].copy())
            if paths is not None:
                paths.append(rh.store_path)
            rh.release()
        mesh.destroy()
        grid.destroy()
    finally:
        gpu_lib.tune("store_boxes", 1)
    return out


def test_nearest_through_index_bins_equals_the_bvh_search(gpu_lib, oracle, regional_case):
    """Nearest-neighbour Store on a grid that knows its projection: cells binned in the grid's index space, a few bins around
    every target point, exactness from a lower bound of the chord length of one index unit -- must give the BVH search's (= the
    oracle's brute-force) indices, ties to the lowest id included: a regional mesh under a Lambert grid that sticks out of it
    (the far points cannot be settled within the rings: the whole Store goes back to the BVH), a row block on a windowed mesh,
    staggered points, a fine mesh under a coarse grid (many cells per bin), a mesh coarser than the grid (empty bins)."""
    from mpassit_amd import regrid as R, synth, target_grid as tg, workloads
    from conftest import mesh_xyz
    m, g = regional_case
    for grid_fn, kw in ((lambda: R.Grid.from_target(g), {}), (lambda: R.Grid.from_target(g, rows=(25, 60)), dict(window=True)),
                        (lambda: R.Grid.from_target(g, rows=(30, 50)), dict(staggers=(0, 1, 2)))):
        a, b = _nearest(gpu_lib, R, m, grid_fn, 1, **kw), _nearest(gpu_lib, R, m, grid_fn, 0, **kw)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
    # the tiny workload: grid inside the mesh -> the bins settle every point; against the oracle's brute force
    mt, gt, _, _ = workloads.workload("tiny")
    got = _nearest(gpu_lib, R, mt, lambda: R.Grid.from_proj(gt, fill_target=False), 1)[0]
    cxyz, _ = mesh_xyz(oracle, mt)
    want = oracle.nearest(cxyz, oracle.lonlat_deg_to_xyz(gt.lon, gt.lat))
    diff = got != want
    if diff.any():                                                # distance ties between the two sin / cos implementations only
        pts = oracle.lonlat_deg_to_xyz(gt.lon, gt.lat).reshape(-1, 3)
        assert np.abs(((cxyz[got[diff]] - pts[diff]) ** 2).sum(1) - ((cxyz[want[diff]] - pts[diff]) ** 2).sum(1)).max() <= 1e-14
    assert diff.mean() < 1e-3
    # a GLOBAL lat-lon grid (periodic index, polar rows left to the masked BVH search, cells within a degree of the poles without
    # an index) and a global mesh under a Lambert grid (cells beyond 60 S have no Lambert index: capped, never binned)
    mg, gg, _, _ = workloads.workload("c5_small")
    for st in ((0,), (0, 2)):
        a = _nearest(gpu_lib, R, mg, lambda: R.Grid.from_proj(gg, fill_target=False), 1, staggers=st)
        b = _nearest(gpu_lib, R, mg, lambda: R.Grid.from_proj(gg, fill_target=False), 0, staggers=st)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
    a = _nearest(gpu_lib, R, mg, lambda: R.Grid.from_proj(gt, fill_target=False), 1)[0]
    b = _nearest(gpu_lib, R, mg, lambda: R.Grid.from_proj(gt, fill_target=False), 0)[0]
    assert np.array_equal(a, b)
    # constructed exact ties: two cells mirrored about a row of target points would need an exactly symmetric mesh; the lattice of
    # `tiny` has many near-ties instead, which the comparison above covers.  Coarse and fine extremes:
    coarse_grid = tg.define_target_grid_params("lambert", 31, 21, dx=150000.0, dy=150000.0, ref_lat=38.5, ref_lon=-97.5, truelat1=38.5, truelat2=38.5,
                                               stand_lon=-97.5)
    for mesh_, grid_ in ((mt, coarse_grid), (synth.regional_mesh_for_lambert(gt.proj, 181, 107, 1500), gt)):
        a = _nearest(gpu_lib, R, mesh_, lambda: R.Grid.from_target(grid_), 1)[0]
        b = _nearest(gpu_lib, R, mesh_, lambda: R.Grid.from_target(grid_), 0)[0]
        assert np.array_equal(a, b)


def test_nearest_bins_at_the_lambert_cut_and_on_wide_regional_latlon_grids(gpu_lib):
    """Places where a cell near a point on the sphere sits far from it in index space: the cut of a Lambert projection (the
    meridian opposite its standard longitude, here crossed by a grid centred 150 degrees away from it), and the index seam of
    a regional lat-lon grid (indices unwrapped about the middle column; a grid too wide for that uses the tree).  A global
    mesh under each; the bins' answer equals the tree's."""
    from mpassit_amd import regrid as R, target_grid as tg, workloads
    mg, _, _, _ = workloads.workload("c5_small")
    # (the east edges of both Lambert grids come within a degree or two of the cut at 82 E without crossing it: a grid ACROSS the cut maps
    # a wedge of the plane to no place on the sphere and is no grid at all)
    grids = [tg.define_target_grid_params("lambert", 61, 41, dx=40000.0, dy=40000.0, ref_lat=45.0, ref_lon=63.5, truelat1=30.0, truelat2=60.0,
                                          stand_lon=-98.0),
             tg.define_target_grid_params("lambert", 51, 51, dx=30000.0, dy=30000.0, ref_lat=50.0, ref_lon=64.0, truelat1=50.0, truelat2=50.0,
                                          stand_lon=-98.0),
             tg.define_target_grid_params("lat-lon", 281, 61, dx=0.5, dy=0.5, ref_lat=10.0, ref_lon=170.0),     # 140 degrees wide, across 180 E
             tg.define_target_grid_params("lat-lon", 331, 41, dx=1.0, dy=1.0, ref_lat=-20.0, ref_lon=10.0)]    # 330 degrees wide: too wide to unwrap
    seen = []
    for g, want in zip(grids, ((1, 2), (1, 2), (1,), (0,))):
        pa, pb = [], []
        a = _nearest(gpu_lib, R, mg, lambda: R.Grid.from_target(g), 1, staggers=(0, 1), paths=pa)
        b = _nearest(gpu_lib, R, mg, lambda: R.Grid.from_target(g), 0, staggers=(0, 1), paths=pb)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
        assert pb == [0, 0] and pa[0] in want and pa[1] in want, (pa, pb)      # the search under test did run (and stood aside where it must)
        seen += pa
    assert 2 in seen                                              # ... and the points next to the cut were handed to the tree


def test_polar_stereographic_and_mercator_grids(gpu_lib):
    """Round 5: the inverses of the other two projections of the namelist (llij_ps, module_map_utils.F90:718-760; llij_merc,
    :1320-1341).  A polar stereographic grid with the NORTH POLE inside it (a regular point of that map: no exclusion zone, no
    cut), one in the southern hemisphere far from its pole, a Mercator grid across the date line with its known point in a corner
    (the map's cut is taken opposite the grid's middle column, not opposite the known point): owners, weights and the conservative
    matrix identical to the pyramid walk's, and the Stores did go through the index space."""
    from mpassit_amd import regrid as R, synth, target_grid as tg
    m = synth.icosahedral_mesh(6)
    grids = [tg.define_target_grid_params("polar", 141, 121, dx=50000.0, dy=50000.0, ref_lat=89.0, ref_lon=25.0, truelat1=75.0, stand_lon=-100.0),
             tg.define_target_grid_params("polar", 121, 101, dx=40000.0, dy=40000.0, ref_lat=-55.0, ref_lon=140.0, truelat1=-71.0, stand_lon=150.0),
             tg.define_target_grid_params("mercator", 161, 91, dx=70000.0, dy=70000.0, ref_lat=-8.0, ref_lon=179.0, truelat1=-15.0, stand_lon=179.0),
             tg.define_target_grid_params("mercator", 201, 81, dx=60000.0, dy=60000.0, ref_lat=20.0, ref_lon=100.0, ref_x=1.0, ref_y=1.0, truelat1=30.0, stand_lon=0.0)]
    for g in grids:
        got = _case(gpu_lib, m, g, staggers=(0, 1, 2))
        assert (got["bil0"][0] >= 0).all() and got["cons"][1].size > 0
        mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_proj(g, fill_target=False)
        for kw in (dict(regridmethod=R.REGRIDMETHOD_BILINEAR), dict(regridmethod=R.REGRIDMETHOD_CONSERVE), dict(regridmethod=R.REGRIDMETHOD_NEAREST_STOD)):
            rh = R.regrid_store(mesh, grid, **kw)
            assert rh.store_path >= 1, (g.proj.code, kw)
            rh.release()
        mesh.destroy()
        grid.destroy()


def test_nearest_on_the_two_projections_equals_the_tree(gpu_lib):
    """The nearest Store through index bins on polar stereographic / Mercator grids against the tree search ("store_boxes" 0):
    identical indices, CENTER and EDGE1 points; meshes finer and coarser than the grid."""
    from mpassit_amd import regrid as R, synth, target_grid as tg
    grids = [tg.define_target_grid_params("polar", 141, 121, dx=50000.0, dy=50000.0, ref_lat=89.0, ref_lon=25.0, truelat1=75.0, stand_lon=-100.0),
             tg.define_target_grid_params("mercator", 161, 91, dx=70000.0, dy=70000.0, ref_lat=-8.0, ref_lon=179.0, truelat1=-15.0, stand_lon=179.0)]
    for level in (4, 7):
        m = synth.icosahedral_mesh(level)
        for g in grids:
            res = []
            for boxes in (1, 0):
                gpu_lib.tune("store_boxes", boxes)
                try:
                    mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_proj(g, fill_target=False)
                    out = []
                    for st in (0, 1):
                        rh = R.regrid_store(mesh, grid, R.REGRIDMETHOD_NEAREST_STOD, staggerloc=st)
                        out.append((rh.weights()[0], rh.store_path))
                        rh.release()
                    res.append(out)
                    mesh.destroy()
                    grid.destroy()
                finally:
                    gpu_lib.tune("store_boxes", 1)
            for (a, pa), (b, pb) in zip(*res):
                assert np.array_equal(a, b) and pa >= 1 and pb == 0
