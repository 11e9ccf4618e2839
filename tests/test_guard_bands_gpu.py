"""Out-of-range accesses of the Regrid kernels, looked for where they would hide: every kernel family (lane gather, row gather,
LDS-staged cell-fast and level-fast; float32 / float64; either byte order) runs with its source slab embedded between two
bands of NaNs and its destination between two bands of a canary pattern.  A load outside the slab that reaches a result
shows as a NaN, a store outside the destination as a broken canary; the results must equal the run on plain buffers bit for
bit.  Boundaries on purpose: level counts around the 16-level chunks and the 64-level row pieces, a grid whose shifted tiles
hang over both edges, the last cell of the mesh referenced, unmapped rim points, a windowed source.
(Round 3 left one unexplained GPU memory fault in an experiment build -- profiles/r04_lin_fault.md; this is the standing
check that the shipped kernels keep to their buffers.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
PAD = 8192


def _embedded(torch, a, fill):
    big = torch.full((a.numel() + 2 * PAD,), fill, dtype=a.dtype, device="cuda")
    big[PAD:PAD + a.numel()].copy_(a.reshape(-1))
    return big, big[PAD:PAD + a.numel()]


@pytest.mark.parametrize("nlev", [2, 15, 16, 17, 55, 64, 65])
def test_kernels_keep_to_their_buffers(gpu_lib, regional_case, nlev):
    import torch
    from mpassit_amd import regrid as R
    m, g = regional_case
    mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_target(g)
    rh = R.regrid_store(mesh, grid, R.REGRIDMETHOD_BILINEAR)
    a, b = rh.source_range()
    assert (rh.weights()[0][:, 0] < 0).any()                     # unmapped rim points take the masked paths
    nf = 3
    gen = torch.Generator(device="cuda")
    gen.manual_seed(nlev)
    canary = -7.0e33
    try:
        for windowed in (False, True):
            if windowed:
                mesh.set_source_window(a, b - a)
                rh._refresh()
            n_src = rh.n_src
            for dt in (torch.float32, torch.float64):
                src = (torch.rand((nf, nlev, n_src), dtype=torch.float64, device="cuda", generator=gen) * 50.0 + 250.0).to(dt)
                for layout in (R.LAYOUT_CELL_FAST, R.LAYOUT_LEV_FAST):
                    s = src if layout == R.LAYOUT_CELL_FAST else src.permute(0, 2, 1).contiguous()
                    for a3, lfv in ((-1, -1), (-2, 2), (0, 0), (1, 1), (2, 1)):
                        gpu_lib.tune("a3_staged", a3)
                        gpu_lib.tune("lf_variant", lfv)
                        for be in (False, True):
                            want = rh.regrid_typed(s.reshape(-1), nlev=nlev, nfields=nf, layout=layout, out_dtype=dt, scale=1.5, offset=-3.0,
                                                   src_be=be, dst_be=be)
                            sbig, sview = _embedded(torch, s, float("nan"))
                            out = torch.empty((nf, nlev, g.ny, g.nx), dtype=dt, device="cuda")
                            obig, oview = _embedded(torch, out, canary)
                            rh.regrid_typed(sview, nlev=nlev, nfields=nf, layout=layout, out=oview.view(nf, nlev, g.ny, g.nx), scale=1.5, offset=-3.0,
                                            src_be=be, dst_be=be)
                            torch.cuda.synchronize()
                            what = (nlev, windowed, dt, layout, a3, lfv, be)
                            it = torch.int32 if dt == torch.float32 else torch.int64
                            assert torch.equal(oview.view(it), want.reshape(-1).view(it)), what        # bit patterns (big-endian results)
                            assert bool((obig[:PAD] == canary).all()) and bool((obig[-PAD:] == canary).all()), what
                            assert bool(torch.isnan(sbig[:PAD]).all()) and bool(torch.isnan(sbig[-PAD:]).all()), what
    finally:
        gpu_lib.tune("a3_staged", -1)
        gpu_lib.tune("lf_variant", -1)
        mesh.set_source_window(0, m.nCells)
    rh.release()
    mesh.destroy()
    grid.destroy()
