"""Collisions / exact ties: destination points that coincide with source cell centres (a vertex shared by ~6 dual
triangles: lowest triangle id must win, weights exactly (1,0,0) up to order), exactly equidistant nearest-neighbour
candidates (lowest cell id wins, App. A6), and argument limits."""
import numpy as np
import pytest

from conftest import mesh_xyz

pytestmark = pytest.mark.gpu


def test_points_on_cell_centres(gpu_lib, oracle, global_mesh):
    from mpassit_amd import regrid as R
    m = global_mesh
    lon_d, lat_d = oracle.mesh_coords_deg(m.lonCell, m.latCell)          # degrees exactly as the library derives them
    sel = np.arange(0, 20000, 7)[:2800].reshape(40, 70)                   # a 70 x 40 "grid" made of cell centres
    grid = R.Grid(lon_d[sel], lat_d[sel])
    mesh = R.Mesh.from_mpas(m)
    cxyz, _ = mesh_xyz(oracle, m)
    tri, _ = oracle.dual_triangles(m.verticesOnCell, m.nVertices, cxyz)
    pxyz = oracle.lonlat_deg_to_xyz(lon_d[sel], lat_d[sel])
    idx_o, w_o = oracle.bilinear_weights(cxyz, tri, pxyz)
    rh = R.regrid_store(mesh, grid, R.REGRIDMETHOD_BILINEAR)
    idx_g, w_g = rh.weights()
    assert (idx_g[:, 0] >= 0).all()
    # the coincident cell carries all the weight, whichever of its triangles was chosen
    for idx, w in ((idx_o, w_o), (idx_g, w_g)):
        k = np.argmax(w, axis=1)
        assert np.array_equal(idx[np.arange(len(idx)), k], sel.ravel())
        assert np.abs(w.max(1) - 1).max() < 1e-9 and np.abs(np.sort(w, axis=1)[:, :2]).max() < 1e-9
    # both sides resolve the 5..7-fold tie the same way (lowest triangle id) for the overwhelming majority
    assert (np.sort(idx_o, 1) == np.sort(idx_g, 1)).all(1).mean() > 0.98
    src = np.random.default_rng(0).standard_normal((1, m.nCells))
    out = rh.regrid(src, nlev=1).ravel()
    assert np.abs(out - src[0, sel.ravel()]).max() < 1e-8               # interpolation reproduces the cell value
    # nearest neighbour of a cell centre is that cell, bit-exact copy
    rn = R.regrid_store(mesh, grid, R.REGRIDMETHOD_NEAREST_STOD)
    assert np.array_equal(rn.weights()[0][:, 0], sel.ravel())
    assert np.array_equal(rn.regrid(src, nlev=1).ravel(), src[0, sel.ravel()])
    rn.release()
    rh.release()
    mesh.destroy()
    grid.destroy()


def test_equidistant_nearest_lowest_id(gpu_lib):
    """Sites mirrored about the meridian plane lon = 0: queries on lon = 0 are exactly equidistant (the x and z
    arithmetic is identical, y differs only in sign) -> the lower cell id must win; swapping ids swaps the answer."""
    from mpassit_amd import regrid as R
    lat = np.deg2rad(np.array([10.0, 10.0, -60.0, 40.0, 40.0]))
    lon = np.deg2rad(np.array([20.0, 340.0, 180.0, 5.0, 355.0]))          # cells 0/1 and 3/4 are mirror pairs
    voc = np.array([[1, 2, 3]] * 5, np.int32)
    vlat, vlon = np.zeros(3), np.deg2rad(np.array([0.0, 120.0, 240.0]))
    qlon = np.zeros((1, 4))
    qlat = np.array([[10.0, 12.0, 38.0, 41.0]])
    grid = R.Grid(qlon, qlat)
    for order in ([0, 1, 2, 3, 4], [1, 0, 2, 4, 3]):
        mesh = R.Mesh(lat[order], lon[order], vlat, vlon, voc)
        rh = R.regrid_store(mesh, grid, R.REGRIDMETHOD_NEAREST_STOD)
        got = rh.weights()[0][:, 0]
        assert list(got) == [0, 0, 3, 3]                                   # the lower id of each mirror pair
        rh.release()
        mesh.destroy()
    grid.destroy()


def test_argument_limits(gpu_lib):
    import ctypes as C
    from mpassit_amd import _lib
    lib = _lib.load()
    h = C.c_void_p()
    one = (C.c_double * 1)(0.0)
    voc = (C.c_int32 * 3)(1, 1, 1)
    assert lib.mpg_mesh_create(C.c_int64(2 ** 31), C.c_int64(10), C.c_int(3), one, one, one, one, voc, C.byref(h)) == 2   # INVALID_ARG
    assert b"int32" in lib.mpg_last_error()
    assert lib.mpg_mesh_create(C.c_int64(0), C.c_int64(10), C.c_int(3), one, one, one, one, voc, C.byref(h)) == 2
    assert lib.mpg_grid_create(C.c_int(50000), C.c_int(50000), C.c_int(0), one, one, None, None, None, None, None, None, C.byref(h)) == 2
    assert lib.mpg_regrid_store(None, C.c_int(0), None, C.c_int(0), C.c_int(0), C.byref(h)) == 2
    assert lib.mpg_handle_release(None) == 0                              # releasing nothing is fine (like ESMF on a null handle)


@pytest.mark.parametrize("case", ["global", "regional", "latlon_global"])
def test_wave_cooperative_search_equals_per_thread_search(gpu_lib, global_mesh, conus_grid_30km, regional_case, case):
    """nn_variant 1 (default: one wavefront walks the BVH for a patch of 8 x 8 points, shared stack, ballot-pruned) against
    nn_variant 0 (one thread per point, private stack): the same comparisons in every lane -> the same indices, ties
    included, on ragged patches (grid sizes not multiples of 8), outside the mesh footprint and across the date line."""
    from mpassit_amd import regrid as R, synth, target_grid as tg
    if case == "global":
        m, g = global_mesh, conus_grid_30km
    elif case == "regional":
        m, g = regional_case
    else:
        m, g = synth.icosahedral_mesh(5), tg.define_target_grid_params("lat-lon", 73, 37, stand_lon=-180.0, is_regional=False)
    got = {}
    try:
        for v in (0, 1):
            gpu_lib.tune("nn_variant", v)
            mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_target(g)     # fresh objects: no cached handle
            rh = R.regrid_store(mesh, grid, R.REGRIDMETHOD_NEAREST_STOD)
            got[v] = rh.weights()[0][:, 0].copy()
            rh.release()
            mesh.destroy()
            grid.destroy()
    finally:
        gpu_lib.tune("nn_variant", 1)
    assert got[0].min() >= 0 and np.array_equal(got[0], got[1])
