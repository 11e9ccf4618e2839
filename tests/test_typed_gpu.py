"""Fused ingest/egress Regrid (float32 in the file -> float64 arithmetic -> float32 in the file): results must be
bit-identical to what the reference's reader/writer chain produces (widen at read, regrid in float64, post-op,
narrow at write: input_data.F90:630-655, write_data.F90:779,1343,1418)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("method", ["bilinear", "nearest", "conserve"])
def test_typed_regrid_equals_reference_chain(gpu_lib, regional_case, method):
    import torch
    from mpassit_amd import regrid as R, synth
    m, g = regional_case
    mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_target(g)
    rh = R.regrid_store(mesh, grid, {"bilinear": R.REGRIDMETHOD_BILINEAR, "nearest": R.REGRIDMETHOD_NEAREST_STOD,
                                     "conserve": R.REGRIDMETHOD_CONSERVE}[method])
    nlev = 7
    src32 = torch.as_tensor(synth.analytic_field(m.latCell, m.lonCell, nlev, dtype=np.float32) + 300, device="cuda")   # file: float32
    src64 = src32.double()                                      # what the reference holds after nf90_get_var
    ref64 = rh.regrid(src64.view(-1), nlev=nlev)                # float64 regrid (the parity-tested path)
    for layout, s32, s64 in ((R.LAYOUT_CELL_FAST, src32, src64),
                             (R.LAYOUT_LEV_FAST, src32.t().contiguous(), src64.t().contiguous())):
        # theta -> T: regrid, subtract 300 in float64, narrow to float32 at write
        want = (ref64 - 300.0).float()
        got = rh.regrid_typed(s32.view(-1), nlev=nlev, layout=layout, out_dtype=torch.float32, offset=-300.0)
        assert got.dtype == torch.float32 and torch.equal(got, want)
        # zgrid -> PHB: * 9.81
        assert torch.equal(rh.regrid_typed(s32.view(-1), nlev=nlev, layout=layout, out_dtype=torch.float32, scale=9.81), (ref64 * 9.81).float())
        # mixed types and the identity epilogue reproduce the float64 path exactly
        assert torch.equal(rh.regrid_typed(s64.view(-1), nlev=nlev, layout=layout, out_dtype=torch.float64), ref64)
        assert torch.equal(rh.regrid_typed(s32.view(-1), nlev=nlev, layout=layout, out_dtype=torch.float64), ref64)
        assert torch.equal(rh.regrid_typed(s64.view(-1), nlev=nlev, layout=layout, out_dtype=torch.float32), ref64.float())
    # a 2-field bundle
    two = torch.stack([src32, 2 * src32]).contiguous()
    got2 = rh.regrid_typed(two.view(-1), nlev=nlev, nfields=2, out_dtype=torch.float32)
    assert torch.equal(got2[0], ref64[0].float())
    rh.release()
    mesh.destroy()
    grid.destroy()
