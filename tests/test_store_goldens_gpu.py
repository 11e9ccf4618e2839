"""The library's whole RegridStores, through the C-ABI, against tests/golden/store_hp.json (brute force at 50 digits, see
tests/test_store_goldens.py): bilinear, nearest and conservative on a regional mesh with rim and unmapped points, across the date line,
and with the North Pole inside the grid -- with and without the projection's inverse attached (index-space and hierarchical candidate
searches), and on a mesh cut to the grid's window.  Not the oracle: an answer that shares no code with either side."""
import numpy as np
import pytest

from test_store_goldens import cases, check_bilinear, check_conserve, check_nearest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("windowed", [False, True], ids=["whole_mesh", "mesh_window"])
@pytest.mark.parametrize("case", cases(), ids=lambda c: c.name)
def test_library_whole_stores_equal_the_brute_force_goldens(gpu_lib, case, windowed):
    from mpassit_amd import regrid as R
    grid = R.Grid(case.lon, case.lat, case.lon_c, case.lat_c)
    mesh = R.Mesh.from_mpas(case.mesh, window_grid=grid if windowed else None)
    rb = R.regrid_store(mesh, grid, R.REGRIDMETHOD_BILINEAR)
    idx, w = rb.weights()
    wb = check_bilinear(case, idx, w)
    rn = R.regrid_store(mesh, grid, R.REGRIDMETHOD_NEAREST_STOD)
    check_nearest(case, rn.weights()[0][:, 0])
    rc = R.regrid_store(mesh, grid, R.REGRIDMETHOD_CONSERVE)
    wc = check_conserve(case, *rc.csr())
    print("%s: library vs 50-digit brute force: bilinear %.1e, conservative %.1e" % (case.name, wb, wc))
    for h in (rb, rn, rc):
        h.release()
    mesh.destroy()
    grid.destroy()


def test_index_space_search_gives_the_golden_too(gpu_lib):
    """The same answer when the candidate search goes through the projection's inverse: case 1's grid made by the library from its
    namelist (mpg_grid_attach_proj claimed and accepted), the Stores report the index-space path."""
    from mpassit_amd import regrid as R, target_grid as tg
    case = cases()[0]
    g = tg.define_target_grid_params("lambert", 18, 14, dx=120000.0, dy=120000.0, ref_lat=38.5, ref_lon=-97.5, truelat1=38.5, truelat2=38.5,
                                     stand_lon=-97.5)
    assert np.array_equal(g.lon, case.lon) and np.array_equal(g.lat_c, case.lat_c)        # the fixture's grid, bit for bit
    grid, mesh = R.Grid.from_target(g), R.Mesh.from_mpas(case.mesh)
    rb = R.regrid_store(mesh, grid, R.REGRIDMETHOD_BILINEAR)
    rn = R.regrid_store(mesh, grid, R.REGRIDMETHOD_NEAREST_STOD)
    rc = R.regrid_store(mesh, grid, R.REGRIDMETHOD_CONSERVE)
    assert all(h.store_path in (1, 2) for h in (rb, rn, rc)), [h.store_path for h in (rb, rn, rc)]
    check_bilinear(case, *rb.weights())
    check_nearest(case, rn.weights()[0][:, 0])
    check_conserve(case, *rc.csr())
    for h in (rb, rn, rc):
        h.release()
    mesh.destroy()
    grid.destroy()


@pytest.mark.parametrize("case", cases(), ids=lambda c: c.name)
def test_library_destaggering_stores_equal_the_brute_force_goldens(gpu_lib, case):
    """CENTER -> EDGE1 / EDGE2 (the U / V destaggering, interp.F90:298,316) through the C-ABI against the 40-digit findroot goldens: the same
    points mapped, the same weights; points on a border of two quads (every V point of a lat-lon grid) or on the hull are judged as such."""
    from mpassit_amd import regrid as R
    from test_store_goldens import check_grid_to_grid, tol_for
    grid = R.Grid(case.lon, case.lat, case.lon_c, case.lat_c, case.lon_u, case.lat_u, case.lon_v, case.lat_v)
    for key, stag in (("edge1", R.STAGGERLOC_EDGE1), ("edge2", R.STAGGERLOC_EDGE2)):
        rh = R.regrid_store_grid(grid, stag)
        idx, w = rh.weights()
        worst, one_sided = check_grid_to_grid(case, key, idx, w)
        assert worst < 10 * tol_for(case) and not one_sided, (case.name, key, worst, one_sided[:10])
        rh.release()
    grid.destroy()
