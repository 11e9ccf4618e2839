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
