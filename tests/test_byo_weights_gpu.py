"""Bring-your-own weights (ESMF factorList / factorIndexList form): export -> import round trips reproduce the
original Regrid; a hand-made weight matrix gives the hand-computed answer."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_export_import_round_trip(gpu_lib, regional_case):
    from mpassit_amd import regrid as R, synth
    m, g = regional_case
    mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_target(g)
    src = synth.analytic_field(m.latCell, m.lonCell, 3)
    for method in (R.REGRIDMETHOD_BILINEAR, R.REGRIDMETHOD_NEAREST_STOD, R.REGRIDMETHOD_CONSERVE):
        rh = R.regrid_store(mesh, grid, method)
        want = rh.regrid(src, nlev=3)
        row, col, S = rh.to_esmf_weights()
        assert row.min() >= 1 and row.max() <= g.nx * g.ny and col.min() >= 1 and col.max() <= m.nCells
        perm = np.random.default_rng(1).permutation(row.size)          # ESMF gives no ordering guarantee
        if method == R.REGRIDMETHOD_CONSERVE:
            perm = np.argsort(row, kind="stable")                       # keep the in-row order -> identical summation order
        rh2 = R.RouteHandle.from_weights(m.nCells, g.nx, g.ny, row[perm], col[perm], S[perm])
        got = rh2.regrid(src, nlev=3)
        if method == R.REGRIDMETHOD_BILINEAR:
            assert rh2.nnz_per_row == 3                                 # fast fixed-3 layout recognised
            assert np.abs(got - want).max() <= 1e-14 * np.abs(want).max()
        else:
            assert np.array_equal(got, want)
        rh2.release()
        rh.release()
    mesh.destroy()
    grid.destroy()


def test_hand_made_matrix(gpu_lib):
    from mpassit_amd import regrid as R
    from mpassit_amd._lib import MpgError
    # 2 x 2 destination, 3 sources: dst(1,1) = 0.5 s1 + 0.5 s3 ; dst(2,1) = s2 ; dst(1,2) unmapped ; dst(2,2) = 2 s3 - s1
    row = [1, 1, 2, 4, 4]
    col = [1, 3, 2, 3, 1]
    S = [0.5, 0.5, 1.0, 2.0, -1.0]
    rh = R.RouteHandle.from_weights(3, 2, 2, row, col, S)
    src = np.array([[1.0, 10.0, 100.0], [2.0, 20.0, 200.0]])           # 2 levels x 3 sources
    out = rh.regrid(src, nlev=2)[0]
    assert np.array_equal(out[0], [[50.5, 10.0], [0.0, 199.0]])
    assert np.array_equal(out[1], [[101.0, 20.0], [0.0, 398.0]])
    rh.release()
    with pytest.raises(MpgError):
        R.RouteHandle.from_weights(3, 2, 2, [5], [1], [1.0])              # row out of range
