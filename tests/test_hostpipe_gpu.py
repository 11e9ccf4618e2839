"""mpg_regrid_typed: host buffers, typed I/O, chunked full-duplex pipeline.  Must equal the plain host entry point
(mpg_regrid) bit for bit for float64 I/O and the device-side typed Regrid for every other type combination, for all
handle kinds and both source layouts, also when the field is cut into many chunks."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("method", ["BILINEAR", "NEAREST_STOD", "CONSERVE"])
def test_host_pipeline_equals_plain_paths(gpu_lib, regional_case, method):
    import torch

    from mpassit_amd import regrid as R
    m, g = regional_case
    mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_target(g)
    rh = R.regrid_store(mesh, grid, getattr(R, "REGRIDMETHOD_" + method))
    rng = np.random.default_rng(4)
    nlev, nf = 7, 3
    src = rng.normal(280.0, 30.0, (nf, nlev, m.nCells))
    want = rh.regrid(src.reshape(-1), nlev=nlev, nfields=nf)
    got = rh.regrid_typed_host(src, nlev=nlev, nfields=nf)
    assert got.dtype == np.float64 and np.array_equal(got, want)
    src_lf = np.ascontiguousarray(src.transpose(0, 2, 1))
    assert np.array_equal(rh.regrid_typed_host(src_lf, nlev=nlev, nfields=nf, layout=R.LAYOUT_LEV_FAST), want)
    # float32 in / float32 out with the writer's epilogue == the device-side typed Regrid on the same bits
    s32 = src.astype(np.float32)
    dev = rh.regrid_typed(torch.from_numpy(s32).cuda().reshape(-1), nlev=nlev, nfields=nf, out_dtype=torch.float32, scale=9.81,
                          offset=-300.0).cpu().numpy()
    host = rh.regrid_typed_host(s32, nlev=nlev, nfields=nf, out_dtype=np.float32, scale=9.81, offset=-300.0)
    assert host.dtype == np.float32 and np.array_equal(host, dev)
    mixed = rh.regrid_typed_host(s32, nlev=nlev, nfields=nf, out_dtype=np.float64)
    assert np.array_equal(mixed, rh.regrid(s32.astype(np.float64).reshape(-1), nlev=nlev, nfields=nf))
    rh.release()
    mesh.destroy()
    grid.destroy()


def test_many_chunks_and_destagger_handle(gpu_lib):
    """A field large enough to be cut into several level chunks, and the 4-point + pole handle of a periodic grid."""
    from mpassit_amd import regrid as R, synth, target_grid as T
    g = T.define_target_grid_params("lat-lon", nx=721, ny=361, stand_lon=0.0, is_regional=False)     # 720 x 360
    m = synth.icosahedral_mesh(7)                                                                     # 163 842 cells
    mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_target(g)
    rh = R.regrid_store(mesh, grid, R.REGRIDMETHOD_BILINEAR)
    nlev = 200                                                     # 131 MB of float32 source -> 2+ chunks
    src = np.random.default_rng(8).normal(size=(nlev, m.nCells)).astype(np.float32)
    got = rh.regrid_typed_host(src, nlev=nlev, out_dtype=np.float32)
    want = rh.regrid(src.astype(np.float64).reshape(-1), nlev=nlev)
    assert np.array_equal(got, want.astype(np.float32))
    rv = R.regrid_store_grid(grid, R.STAGGERLOC_EDGE2)
    mass = np.random.default_rng(9).normal(size=(5, g.ny, g.nx))
    assert np.array_equal(rv.regrid_typed_host(mass, nlev=5), rv.regrid(mass.reshape(-1), nlev=5))
    rv.release()
    rh.release()
    mesh.destroy()
    grid.destroy()


def test_only_the_referenced_cells_cross_the_link(gpu_lib, conus_grid_30km):
    """A regional grid under a global mesh references a small id range of the cells: the host path uploads that range only
    (each level's run in the cell-fast layout, one block of rows in file order).  The result must not depend on anything outside
    the range -- NaNs there -- and must equal the device path; a second call with other values must not see stale ones."""
    import torch

    from mpassit_amd import regrid as R, synth
    mesh, grid = R.Mesh.from_mpas(synth.icosahedral_mesh(6)), R.Grid.from_target(conus_grid_30km)    # 40 962 cells, Morton-numbered
    for method in (R.REGRIDMETHOD_BILINEAR, R.REGRIDMETHOD_NEAREST_STOD, R.REGRIDMETHOD_CONSERVE):
        rh = R.regrid_store(mesh, grid, method)
        a, b = rh.source_range()
        assert 0 <= a < b <= rh.n_src and (b - a) < 0.6 * rh.n_src       # a range worth trimming
        rng = np.random.default_rng(11)
        nlev, nf = 9, 2
        for trial in range(2):
            src = rng.normal(280.0, 30.0, (nf, nlev, rh.n_src))
            want = rh.regrid(src.reshape(-1), nlev=nlev, nfields=nf)
            holed = src.copy()
            holed[:, :, :a] = np.nan
            holed[:, :, b:] = np.nan
            assert np.array_equal(rh.regrid_typed_host(holed, nlev=nlev, nfields=nf), want)
            holed_lf = np.ascontiguousarray(holed.transpose(0, 2, 1))
            assert np.array_equal(rh.regrid_typed_host(holed_lf, nlev=nlev, nfields=nf, layout=R.LAYOUT_LEV_FAST), want)
            h32 = holed.astype(np.float32)
            dev = rh.regrid_typed(torch.from_numpy(np.nan_to_num(h32)).cuda().reshape(-1), nlev=nlev, nfields=nf, out_dtype=torch.float32).cpu().numpy()
            assert np.array_equal(rh.regrid_typed_host(h32, nlev=nlev, nfields=nf, out_dtype=np.float32), dev)
        rh.release()
    mesh.destroy()
    grid.destroy()


def test_bundle_of_separate_host_arrays_equals_single_calls(gpu_lib, regional_case):
    """mpg_regrid_bundle_typed: the fields of a bundle as separate host arrays through one pipeline == one mpg_regrid_typed per
    field, bit for bit: both layouts, float32 / float64 on either side, per-field offsets, 2-D fields."""
    from mpassit_amd import regrid as R
    m, g = regional_case
    mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_target(g)
    rng = np.random.default_rng(3)
    for method in (R.REGRIDMETHOD_BILINEAR, R.REGRIDMETHOD_CONSERVE):
        rh = R.regrid_store(mesh, grid, method)
        for nlev, nf in ((9, 4), (1, 3), (40, 2)):
            for layout in (R.LAYOUT_CELL_FAST, R.LAYOUT_LEV_FAST):
                for sdt, ddt in ((np.float32, np.float32), (np.float64, np.float64), (np.float32, np.float64), (np.float64, np.float32)):
                    shape = (rh.n_src, nlev) if layout == R.LAYOUT_LEV_FAST else (nlev, rh.n_src)
                    srcs = [rng.normal(280.0, 30.0, shape).astype(sdt) for _ in range(nf)]
                    offs = [-300.0 * (f % 2) for f in range(nf)]
                    got = rh.regrid_bundle_host(srcs, nlev=nlev, layout=layout, out_dtype=ddt, scale=1.5, offsets=offs)
                    for f in range(nf):
                        want = rh.regrid_typed_host(srcs[f], nlev=nlev, nfields=1, layout=layout, out_dtype=ddt, scale=1.5, offset=offs[f])[0]
                        assert got[f].dtype == ddt and np.array_equal(got[f], want), (method, nlev, nf, layout, sdt, ddt, f)
        rh.release()
    mesh.destroy()
    grid.destroy()
