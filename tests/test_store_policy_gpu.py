"""Streaming stores of the Regrid kernels (csrc/geom.h: stream_store_lane / stream_nt): which lanes store non-temporal and which
write-back depends on where a level's plane starts inside a 128-byte line -- never WHAT is stored.  On a grid with an odd number
of points per level (every plane of the result starts somewhere else in a line) and with result arrays that themselves start 4 .. 100
bytes into a line, every policy of the A/B knobs gives the bits of the default, the default the bits of the float64 reference path,
and nothing is written outside the result (canary bands either side)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _case(R, gpu_lib, m, g, nlev, nfields, io, out, layout, shift_bytes, knobs, seed):
    import torch
    mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_target(g)
    rh = R.regrid_store(mesh, grid, R.REGRIDMETHOD_BILINEAR)
    rng = np.random.default_rng(seed)
    src64 = rng.normal(size=(nfields, nlev, m.nCells))
    lev_fast = layout == R.LAYOUT_LEV_FAST
    sdt = torch.float32 if io == "f32" else torch.float64
    odt = torch.float32 if out == "f32" else torch.float64
    src = torch.from_numpy(np.ascontiguousarray(src64.transpose(0, 2, 1) if lev_fast else src64)).to(sdt).cuda()
    # the reference: the float64 cell-fast path on the source as the kernel sees it (float32 sources are widened exactly), narrowed once
    ref_src = src.to(torch.float64).cpu().numpy()
    ref_src = np.ascontiguousarray(ref_src.transpose(0, 2, 1)) if lev_fast else ref_src
    want = torch.from_numpy(rh.regrid(ref_src.reshape(-1), nlev=nlev, nfields=nfields)).to(odt)
    n = nfields * nlev * g.ny * g.nx
    es = odt.itemsize
    pad = 256 // es
    canary = 12345.0
    typed = not (io == "f64" and out == "f64")
    try:
        for key, values in knobs:
            for v in values:
                gpu_lib.tune(key, v)
                raw = torch.full((n + 2 * pad + 32,), canary, dtype=odt, device="cuda")
                assert raw.data_ptr() % 256 == 0
                o0 = pad + shift_bytes // es
                res = raw[o0:o0 + n].view(nfields, nlev, g.ny, g.nx)
                assert res.data_ptr() % 128 == shift_bytes % 128
                if typed:
                    rh.regrid_typed(src.view(-1), nlev=nlev, nfields=nfields, layout=layout, out=res)
                else:
                    rh.regrid(src.view(-1), nlev=nlev, nfields=nfields, layout=layout, out=res)
                torch.cuda.synchronize()
                assert torch.equal(res.cpu().view(torch.int32 if es == 4 else torch.int64), want.view(-1).view(torch.int32 if es == 4 else torch.int64).view(res.shape)), \
                    "%s=%d: %s -> %s, layout %d, result %d bytes into a line" % (key, v, io, out, layout, shift_bytes)
                assert bool((raw[:o0] == canary).all()) and bool((raw[o0 + n:] == canary).all()), "wrote outside the result"
            gpu_lib.tune(key, values[0])
    finally:
        gpu_lib.tune("lf_rows_store", 0)
        gpu_lib.tune("staged_store", 0)
        gpu_lib.tune("a3_staged", -1)
        gpu_lib.tune("lf_variant", -1)
    rh.release()
    mesh.destroy()
    grid.destroy()


@pytest.mark.parametrize("io,out", [("f32", "f32"), ("f32", "f64"), ("f64", "f32"), ("f64", "f64")])
@pytest.mark.parametrize("shift", [0, 8, 40, 100])
def test_odd_planes_and_shifted_results_every_policy(gpu_lib, regional_case, io, out, shift):
    from mpassit_amd import regrid as R
    m, g = regional_case                                   # 150 x 90 mass points; nlev = 9 -> every staged kernel is eligible
    shift -= shift % (4 if out == "f32" else 8)            # whole elements: 100 -> 100 (float32) / 96 (float64)
    # file order: the row gather (per level / per lane / plain / non-temporal), the grid-row-tile fallback, the staged kernel
    _case(R, gpu_lib, m, g, 9, 2, io, out, R.LAYOUT_LEV_FAST, shift, [("lf_rows_store", [0, 1, 2, 3]), ("lf_variant", [-1, 0, 1, 2])], seed=shift)
    # cell-fast: the staged kernel per lane / every lane non-temporal, the lane gather, every staged variant
    _case(R, gpu_lib, m, g, 9, 2, io, out, R.LAYOUT_CELL_FAST, shift, [("staged_store", [0, 2]), ("a3_staged", [-1, -2, 0, 1, 2])], seed=shift + 1)


def test_an_odd_number_of_points_per_level(gpu_lib, global_mesh):
    """181 x 107 namelist values -> 180 x 106 is even; 182 x 108 -> 181 x 107 mass points: 19 367 points per level, plane k starts
    k * 19 367 * 4 (* 8) bytes behind plane 0 -- every residue of 4 (8) mod 128 occurs among 55 levels."""
    from mpassit_amd import regrid as R, target_grid as tg
    from conftest import LAMBERT
    g = tg.define_target_grid_params("lambert", 182, 108, dx=30000.0, dy=30000.0, **LAMBERT)
    assert (g.nx * g.ny) % 2 == 1
    for io, out in (("f32", "f32"), ("f64", "f64"), ("f32", "f64")):
        _case(R, gpu_lib, global_mesh, g, 55, 1, io, out, R.LAYOUT_LEV_FAST, 0, [("lf_rows_store", [0, 2, 3])], seed=7)
        _case(R, gpu_lib, global_mesh, g, 55, 1, io, out, R.LAYOUT_CELL_FAST, 0, [("staged_store", [0, 2]), ("a3_staged", [-1, -2])], seed=8)


def test_knob_ranges(gpu_lib):
    from mpassit_amd import _lib as L
    for key, bad in (("lf_rows_store", 4), ("lf_rows_store", -1), ("staged_store", 1), ("staged_store", 3)):
        with pytest.raises(L.MpgError):
            gpu_lib.tune(key, bad)
    gpu_lib.tune("lf_rows_store", 0)
    gpu_lib.tune("staged_store", 0)
