"""The data-dependent branches of the Store kernels, each FORCED by a case built for it, its taking ASSERTED through
mpg_handle_store_stats, and the weights compared with the ORACLE (not with the other search of the library):
  conservative -- polygons that spill past their 24-entry list from their index box, polygons whose index box a wavefront
                  enumerates (> 128 box cells), polygons the cooperative passes walk the pyramid for, lists copied from the spill area;
  bilinear     -- triangles without a usable index next to the poles of a lat-lon grid: the wave-per-triangle rasteriser;
  nearest      -- the two ends of the bin sizing (a 12-cell mesh: 16-point bins; a 2.6 M-cell mesh: 2-point bins), the tree finishing
                  what the bins cannot vouch for, a grid of a few points.
Reference: the candidate searches stand in for ESMF's own (interp.F90:368-434 Store calls); the weights must not depend on them."""
import numpy as np
import pytest

from _parity_helpers import assert_csr_equal, assert_fixed_weights_equal, assert_nearest_equal, conserve_tol
from conftest import mesh_xyz

pytestmark = pytest.mark.gpu


def _lambert(nx, ny, dx, **kw):
    from mpassit_amd import target_grid as tg
    p = dict(ref_lat=38.5, ref_lon=-97.5, truelat1=38.5, truelat2=38.5, stand_lon=-97.5)
    p.update(kw)
    return tg.define_target_grid_params("lambert", nx, ny, dx=dx, dy=dx, **p)


def _cons_vs_oracle(oracle, m, g, grid=None):
    from mpassit_amd import regrid as R
    mesh = R.Mesh.from_mpas(m)
    grid = grid or R.Grid.from_proj(g)
    rh = R.regrid_store(mesh, grid, R.REGRIDMETHOD_CONSERVE)
    st, path = rh.store_stats, rh.store_path
    rp_g, col_g, val_g = rh.csr()
    rh.release()
    mesh.destroy()
    grid.destroy()
    _, vxyz = mesh_xyz(oracle, m)
    rp_o, col_o, val_o = oracle.conserve(m.verticesOnCell, vxyz, g.nx, g.ny, oracle.lonlat_deg_to_xyz(g.lon_c, g.lat_c))
    n_common, only_o, only_g = assert_csr_equal(rp_o, col_o, val_o, rp_g, col_g, val_g, m.nCells, tol=conserve_tol(oracle, g))
    assert n_common > 0
    return st, path, n_common, only_o, only_g


@pytest.mark.parametrize("dx_km", [40, 45, 50])
def test_conservative_spill_and_wave_enumeration_from_index_boxes(gpu_lib, oracle, dx_km):
    """450-km cells (the 2 562-cell icosahedral mesh; 520 km from vertex to vertex) under a 40 / 45 / 50-km Lambert grid: a polygon spans
    10-13 index units (sixteen is the limit of the boxes) and under five degrees, its box holds more than 128 cells (a wavefront
    enumerates it) and 70-110 of them are candidates (past the 24-entry list, into the spill area, copied by the list pass); at most a few
    walk the pyramid."""
    from mpassit_amd import synth
    m = synth.icosahedral_mesh(4)
    g = _lambert(121, 91, dx_km * 1000.0)
    st, path, *_ = _cons_vs_oracle(oracle, m, g)
    assert path == 1
    pairs, spilled, wave, walked, copied, slots = st[1:7]
    assert wave > 0 and spilled > 0 and copied > 0, st          # MODE 7 ran, lists outgrew 24 entries, spill areas were copied
    assert walked <= spilled // 4 and copied + walked == spilled, st   # ... nearly every count exact from the boxes (a few polygons span > 16 units)
    assert pairs > 24 * spilled and slots >= 9


def test_conservative_lane_spill_without_wave_enumeration(gpu_lib, oracle):
    """220-km cells under a 28-km grid: boxes of about 10 x 10 cells (at most 128: the lane enumerates them itself) with 40-70
    candidates -- the lane-form spill."""
    from mpassit_amd import synth
    m = synth.icosahedral_mesh(5)
    g = _lambert(141, 101, 28000.0)
    st, path, *_ = _cons_vs_oracle(oracle, m, g)
    assert path == 1 and st[2] > 0 and st[5] > 0 and st[4] == 0, st
    assert st[3] < st[2], st                                     # most (the pentagons' neighbours aside: all) spilled from the lane form


def test_conservative_cooperative_walk_for_polygons_without_a_box(gpu_lib, oracle):
    """960-km cells (the 162-cell icosahedral mesh) under a 25-km grid: a polygon spans far more than sixteen index units, so it has
    no box; the candidate pass walks, overflows, and the cooperative count and list passes walk the pyramid for it."""
    from mpassit_amd import synth
    m = synth.icosahedral_mesh(2)
    g = _lambert(161, 121, 25000.0)
    st, path, *_ = _cons_vs_oracle(oracle, m, g)
    assert st[4] > 0 and st[2] >= st[4], st                      # walked by a workgroup each
    assert st[1] >= 160 * 120                                    # every destination cell lies in some polygon


def test_conservative_polar_caps_of_a_global_grid(gpu_lib, oracle):
    """A global lat-lon grid: polygons poleward of 85 degrees have no usable index and go straight to the cooperative passes;
    the others through their boxes."""
    from mpassit_amd import workloads
    m, g, _, _ = workloads.workload("c5_small")
    st, path, *_ = _cons_vs_oracle(oracle, m, g)
    assert path == 1 and st[4] > 0 and st[2] >= st[4] and st[1] > 3 * g.nx * g.ny, st     # the caps' polygons walked, the rest through boxes


def test_bilinear_wave_kernel_next_to_the_poles(gpu_lib, oracle):
    """Global lat-lon grid under a global mesh: the dual triangles of the last degrees before the poles have no usable index and a
    wavefront rasterises each (k_tri_raster_big); owners and weights against the oracle."""
    from mpassit_amd import regrid as R, workloads
    m, g, _, _ = workloads.workload("c5_small")
    mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_proj(g)
    rh = R.regrid_store(mesh, grid, R.REGRIDMETHOD_BILINEAR)
    st = rh.store_stats
    idx_g, w_g = rh.weights()
    rh.release()
    mesh.destroy()
    grid.destroy()
    assert st[0] == 1 and 0 < st[1] < st[2], st                  # some, not all, triangles went to the wave kernel
    cxyz, _ = mesh_xyz(oracle, m)
    tri, _ = oracle.dual_triangles(m.verticesOnCell, m.nVertices, cxyz)
    idx_o, w_o = oracle.bilinear_weights(cxyz, tri, oracle.lonlat_deg_to_xyz(g.lon, g.lat))
    assert_fixed_weights_equal(idx_o, w_o, idx_g, w_g, tol=1e-10)


def _nearest_vs_oracle(oracle, m, g, brute):
    from mpassit_amd import regrid as R
    mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_proj(g)
    rh = R.regrid_store(mesh, grid, R.REGRIDMETHOD_NEAREST_STOD)
    st = rh.store_stats
    idx_g, _ = rh.weights()
    rh.release()
    mesh.destroy()
    grid.destroy()
    cxyz, _ = mesh_xyz(oracle, m)
    pxyz = oracle.lonlat_deg_to_xyz(g.lon, g.lat)
    idx_o = oracle.nearest(cxyz, pxyz, brute=brute)
    assert_nearest_equal(idx_o, idx_g[:, 0], pxyz, cxyz, max_ties=4)
    return st


def test_nearest_bins_on_a_twelve_cell_mesh(gpu_lib, oracle):
    """12 cells (the icosahedron's vertices) under a 60-km Lambert grid: about no cell per bin whatever the bin -- the largest bins
    (16 grid points), four rings of them settle nothing far from a cell, the tree finishes those points."""
    from mpassit_amd import synth
    m = synth.icosahedral_mesh(0)
    assert m.nCells == 12
    g = _lambert(61, 41, 60000.0)
    st = _nearest_vs_oracle(oracle, m, g, brute=True)
    assert st[0] in (1, 2) and st[1] == 16, st
    if st[0] == 2:
        assert st[3] > 0


def test_nearest_bins_on_a_fine_mesh(gpu_lib, oracle):
    """2.6 M cells (7.5 km) under a 6-km Lambert grid: the mesh is as fine as the grid -- 2-point bins, every point settled by them."""
    from mpassit_amd import synth
    m = synth.icosahedral_mesh(9)
    g = _lambert(241, 161, 6000.0)
    st = _nearest_vs_oracle(oracle, m, g, brute=False)
    assert st[0] == 1 and st[1] == 2 and st[3] == 0, st
    assert st[2] == ((240 + 16 + 1) // 2) * ((160 + 16 + 1) // 2)      # bins of the grid plus four rings of margin


def test_nearest_bin_side_follows_the_mesh(gpu_lib, oracle):
    """30-km cells under a 3-km grid (configuration 2's ratio): about ten grid points per cell -- a bin side in between."""
    from mpassit_amd import synth
    m = synth.icosahedral_mesh(8)
    g = _lambert(201, 151, 3000.0)
    st = _nearest_vs_oracle(oracle, m, g, brute=False)
    assert 6 <= st[1] <= 12 and st[0] in (1, 2), st


def test_nearest_on_a_grid_of_a_few_points(gpu_lib, oracle):
    """3 x 2 mass points: the bin grid is almost all margin (the counters are sized for the smallest bins and that size is never
    exceeded by larger ones)."""
    from mpassit_amd import synth
    g = _lambert(4, 3, 50000.0)
    for level in (2, 5):
        st = _nearest_vs_oracle(oracle, synth.icosahedral_mesh(level), g, brute=True)
        assert st[1] >= 2 and st[2] > 0, st


def test_nearest_tree_finishes_what_the_bins_leave(gpu_lib, oracle, regional_case):
    """A grid that sticks out of a regional mesh: points further than four rings from any cell are left at -1 by the bins and the
    tree search finishes exactly those."""
    m, g = regional_case
    st = _nearest_vs_oracle(oracle, m, g, brute=True)
    assert st[0] == 2 and 0 < st[3] < g.nx * g.ny, st


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 255, 256, 257, 4095, 4096, 4097, 8191, 8192, 12288, 1 << 20, (1 << 20) + 1, 256 * 4096, 256 * 4096 + 1,
                               3 * 256 * 4096 - 1, 5_000_003])
def test_the_scan_and_sum_primitives_at_awkward_sizes(gpu_lib, n):
    """csrc/k_prims.hip (the library's own exclusive scan and 64-bit sum, in rocPRIM's place since round 5) asked directly: sizes around
    the wavefront, the workgroup, the 4096-entry block and the 256-block walk of the block sums; out of place and in place; sums beyond
    32 bits."""
    import ctypes as C
    rng = np.random.default_rng(n)
    a = rng.integers(0, 400, n).astype(np.int32)
    out = np.empty(n, np.int32)
    tot = C.c_longlong()
    gpu_lib.check(gpu_lib.load().mpg_debug_scan_i32(a.ctypes.data_as(C.c_void_p), C.c_int64(n), out.ctypes.data_as(C.c_void_p), C.byref(tot)))
    want = np.concatenate([[0], np.cumsum(a[:-1], dtype=np.int64)])
    assert want[-1] < 2 ** 31 and np.array_equal(out, want.astype(np.int32)) and tot.value == int(a.sum(dtype=np.int64))
    big = np.full(min(n, 70000), 2 ** 31 - 1, np.int32)                 # the 64-bit sum does not wrap where the int32 scan would
    gpu_lib.check(gpu_lib.load().mpg_debug_scan_i32(big.ctypes.data_as(C.c_void_p), C.c_int64(big.size), None, C.byref(tot)))
    assert tot.value == int(big.size) * (2 ** 31 - 1)
    neg = -a                                                              # negative entries subtract (two's complement)
    gpu_lib.check(gpu_lib.load().mpg_debug_scan_i32(neg.ctypes.data_as(C.c_void_p), C.c_int64(n), out.ctypes.data_as(C.c_void_p), C.byref(tot)))
    assert np.array_equal(out, (-want).astype(np.int32)) and tot.value == -int(a.sum(dtype=np.int64))
