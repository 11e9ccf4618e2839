"""LDS-staged Regrid kernels (k_apply_lfu.hip): every variant must reproduce the cell-fast kernel bit for bit
(same wsum3 arithmetic, only the data path differs) on meshes with unmapped rims, global meshes with Morton or
bisection numbering, randomly renumbered cells, ragged tile edges and several fields."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LFU = [0, 1, 2]                  # "lf_variant": row gather on linear tiles, staged in level chunks, row gather on grid-row tiles
A3 = [-2, 0, 1, 2]               # "a3_staged": lane gather, staged on 64x8 / 64x16 (256 threads) / 64x16 (512 threads) tiles
A3_STAGED_DEFAULT = -1           # library default of the "a3_staged" knob (per-handle choice)


def _check(R, gpu_lib, m, g, nlev, nfields=2, seed=0):
    mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_target(g)
    rh = R.regrid_store(mesh, grid, R.REGRIDMETHOD_BILINEAR)
    rng = np.random.default_rng(seed)
    src = rng.normal(size=(nfields, nlev, m.nCells))
    want = rh.regrid(src.reshape(-1), nlev=nlev, nfields=nfields)
    src_lf = np.ascontiguousarray(src.transpose(0, 2, 1))
    try:
        for v in LFU:
            gpu_lib.tune("lf_variant", v)
            got = rh.regrid(src_lf.reshape(-1), nlev=nlev, nfields=nfields, layout=R.LAYOUT_LEV_FAST)
            assert np.array_equal(got, want), "lf_variant %d differs" % v
        gpu_lib.tune("lf_variant", -1)                      # the default: per-handle choice
        got = rh.regrid(src_lf.reshape(-1), nlev=nlev, nfields=nfields, layout=R.LAYOUT_LEV_FAST)
        assert np.array_equal(got, want)
        rh.rebase(0, m.nCells)                               # re-indexing drops the tile lists; they are rebuilt on demand
        got = rh.regrid(src_lf.reshape(-1), nlev=nlev, nfields=nfields, layout=R.LAYOUT_LEV_FAST)
        assert np.array_equal(got, want)
        # cell-fast staged variants and the per-handle choice (-1) against the lane-gather kernel (-2)
        for v in A3 + [-1]:
            gpu_lib.tune("a3_staged", v)
            got = rh.regrid(src.reshape(-1), nlev=nlev, nfields=nfields)
            assert np.array_equal(got, want), "a3_staged %d differs" % v
    finally:
        gpu_lib.tune("lf_variant", -1)
        gpu_lib.tune("a3_staged", A3_STAGED_DEFAULT)
    unmapped = int((rh.weights()[0][:, 0] < 0).sum())
    rh.release()
    mesh.destroy()
    grid.destroy()
    return unmapped


def test_regional_mesh_with_unmapped_rim(gpu_lib, regional_case):
    from mpassit_amd import regrid as R
    m, g = regional_case
    assert _check(R, gpu_lib, m, g, nlev=13) > 0            # 13 levels: ragged last level chunk for every LC


def test_global_meshes_and_numberings(gpu_lib):
    from mpassit_amd import regrid as R, synth, target_grid as T
    g = T.define_target_grid_params("lat-lon", nx=331, ny=167, stand_lon=0.0, is_regional=False)   # 330 x 166: ragged tiles
    for m in (synth.icosahedral_mesh(5), synth.icosahedral_mesh(5, order="native"),
              synth.shuffle_cells(synth.icosahedral_mesh(4), seed=5)):
        assert _check(R, gpu_lib, m, g, nlev=55, nfields=1) == 0


def test_tiny_workload_many_fields(gpu_lib):
    from mpassit_amd import regrid as R, workloads
    m, g, nlev, _ = workloads.workload("tiny")
    _check(R, gpu_lib, m, g, nlev=nlev, nfields=5)


def test_fine_mesh_under_a_coarse_grid_falls_back(gpu_lib):
    """Downscaling direction (30 k-cell mesh -> 18 x 10 grid): neighbouring target points share no cells, a tile's list
    holds up to three cells per point and exceeds what the staged kernel keeps in registers -> the library serves the
    handle with the lane-gather kernel; results are the same either way."""
    from mpassit_amd import regrid as R, target_grid as T, workloads
    m, _, nlev, _ = workloads.workload("tiny")
    g = T.define_target_grid_params("lambert", 19, 11, dx=300000.0, dy=300000.0, ref_lat=38.5, ref_lon=-97.5, truelat1=38.5,
                                    truelat2=38.5, stand_lon=-97.5)
    assert _check(R, gpu_lib, m, g, nlev=nlev, nfields=2) >= 0


def test_typed_staged_paths_agree(gpu_lib):
    """float32 / float64 I/O with the affine epilogue: staged level-fast (file order), staged cell-fast and the lane- /
    row-gather typed kernels all give the same bits on a high-reuse case (5 target points per cell)."""
    import torch

    from mpassit_amd import regrid as R, synth, target_grid as T
    g = T.define_target_grid_params("lat-lon", nx=331, ny=167, stand_lon=0.0, is_regional=False)
    m = synth.icosahedral_mesh(5)
    mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_target(g)
    rh = R.regrid_store(mesh, grid, R.REGRIDMETHOD_BILINEAR)
    nlev, nf = 55, 2
    s64 = torch.randn((nf, nlev, m.nCells), dtype=torch.float64, device="cuda") * 30 + 280
    try:
        for sdt, ddt in ((torch.float32, torch.float32), (torch.float32, torch.float64), (torch.float64, torch.float32)):
            src = s64.to(sdt)
            src_lf = src.permute(0, 2, 1).contiguous()
            outs = []
            for staged, lfv in ((1, -1), (-2, 2), (0, 0), (2, 1)):
                gpu_lib.tune("a3_staged", staged)
                gpu_lib.tune("lf_variant", lfv)
                outs.append(rh.regrid_typed(src.reshape(-1), nlev=nlev, nfields=nf, out_dtype=ddt, scale=9.81, offset=-300.0))
                outs.append(rh.regrid_typed(src_lf.reshape(-1), nlev=nlev, nfields=nf, layout=R.LAYOUT_LEV_FAST, out_dtype=ddt,
                                            scale=9.81, offset=-300.0))
            for o in outs[1:]:
                assert torch.equal(o, outs[0])
    finally:
        gpu_lib.tune("lf_variant", -1)
        gpu_lib.tune("a3_staged", A3_STAGED_DEFAULT)
    rh.release()
    mesh.destroy()
    grid.destroy()


@pytest.mark.parametrize("nlev", [2, 7, 8, 15, 16, 17, 31, 32, 33, 48, 63, 64, 65, 80])
def test_staged_level_fast_kernel_at_every_chunk_boundary(gpu_lib, nlev):
    """k_apply3_lfu walks the levels in chunks of 16 with the last chunk moved back to nlev - 16 and a clamped level index
    below 16 levels: every remainder (and the bundles too short to be staged, which fall to the gather) must give the row
    gather's bits, in float32 and float64, host and big-endian byte order."""
    import torch

    from mpassit_amd import regrid as R, synth, target_grid as T
    g = T.define_target_grid_params("lat-lon", nx=181, ny=91, stand_lon=0.0, is_regional=False)
    m = synth.icosahedral_mesh(4)
    mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_target(g)
    rh = R.regrid_store(mesh, grid, R.REGRIDMETHOD_BILINEAR)
    s64 = torch.randn((2, m.nCells, nlev), dtype=torch.float64, device="cuda") * 30 + 280
    try:
        for dt in (torch.float32, torch.float64):
            src = s64.to(dt).contiguous()
            for be in (False, True):
                outs = []
                for v in (0, 1):
                    gpu_lib.tune("lf_variant", v)
                    outs.append(rh.regrid_typed(src.view(-1), nlev=nlev, nfields=2, layout=R.LAYOUT_LEV_FAST, out_dtype=dt, scale=9.81,
                                                offset=-300.0, src_be=be, dst_be=be))
                assert torch.equal(outs[0].view(torch.uint8), outs[1].view(torch.uint8)), (nlev, dt, be)
        gpu_lib.tune("lf_variant", 1)
        a = rh.regrid(s64.view(-1), nlev=nlev, nfields=2, layout=R.LAYOUT_LEV_FAST)
        gpu_lib.tune("lf_variant", 0)
        b = rh.regrid(s64.view(-1), nlev=nlev, nfields=2, layout=R.LAYOUT_LEV_FAST)
        assert torch.equal(torch.as_tensor(a), torch.as_tensor(b))
        if nlev >= 8:   # the staged kernel did run: 64 x 8-point tile lists are in use and fit its 512 rows
            st = rh.tile_stats()
            assert st is not None and st[:2] == (64, 8) and 0 < rh.kernel_choice()[2] <= 512
    finally:
        gpu_lib.tune("lf_variant", -1)
    rh.release()
    mesh.destroy()
    grid.destroy()


@pytest.mark.parametrize("n_cells,band", [(1500, (1, 64)), (4000, (65, 128)), (9000, (129, 256)), (20000, (257, 512))])
def test_every_row_slot_variant_of_the_staged_level_fast_kernel(gpu_lib, n_cells, band):
    """Round 5: the staged level-fast kernel's row slots per thread follow the handle's longest tile list (2 / 4 / 8 / 16: slabs of 64 / 128 /
    256 / 512 rows).  Meshes of four densities under one 241 x 161 grid put that list into each band; the staged result equals the row
    gather's and the cell-fast kernel's bit for bit -- with the automatic choice, with every larger forced value, and with a forced value too
    small for the handle (which the launcher must not take)."""
    from mpassit_amd import regrid as R, synth, target_grid as T
    g = T.define_target_grid_params("lambert", 242, 162, dx=12000.0, dy=12000.0, ref_lat=38.5, ref_lon=-97.5, truelat1=38.5, truelat2=38.5, stand_lon=-97.5)
    m = synth.regional_mesh_for_lambert(g.proj, 242, 162, n_cells, margin=0.02)
    mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_target(g)
    rh = R.regrid_store(mesh, grid, R.REGRIDMETHOD_BILINEAR)
    nlev, nf = 21, 2
    rng = np.random.default_rng(n_cells)
    src = rng.normal(size=(nf, nlev, m.nCells)).astype(np.float32)
    src_lf = np.ascontiguousarray(src.transpose(0, 2, 1))
    want = rh.regrid_typed_host(src.reshape(-1), nlev=nlev, nfields=nf)                       # cell-fast kernels
    try:
        gpu_lib.tune("lf_variant", 0)
        rows = rh.regrid_typed_host(src_lf.reshape(-1), nlev=nlev, nfields=nf, layout=R.LAYOUT_LEV_FAST)
        assert np.array_equal(rows, want)
        gpu_lib.tune("lf_variant", 1)                                                        # staged, whatever the reuse statistic says
        for npf in (0, 2, 4, 8, 16):
            gpu_lib.tune("lfu_npf", npf)
            got = rh.regrid_typed_host(src_lf.reshape(-1), nlev=nlev, nfields=nf, layout=R.LAYOUT_LEV_FAST)
            assert np.array_equal(got, want), "lfu_npf %d" % npf
        ut_max = rh.kernel_choice()[2]
        assert band[0] <= ut_max <= band[1], (ut_max, band)                                   # the case is in the band it was built for
    finally:
        gpu_lib.tune("lf_variant", -1)
        gpu_lib.tune("lfu_npf", 0)
    rh.release()
    mesh.destroy()
    grid.destroy()
