"""The RCCL path of the C-ABI with a world of one (the GPU box has one card; RCCL refuses two ranks on one device): librccl is
loaded, a communicator made, the all-gather, the halo build + exchange (range form on a banded mesh, compact form on a
shuffled one) and mpg_gather_rows run through it.  What a single rank can check: the re-indexed handle on the exchanged
local slab gives the bits of the global Regrid.  Data crossing xGMI needs the driver's multi-GPU node."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _one(gpu_lib, m, g, expect_mode, rows=None):
    import torch

    from mpassit_amd import comm as MC, regrid as R
    mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_target(g, rows=rows)
    nlev, nf = 6, 2
    src = torch.randn((nf * nlev, m.nCells), dtype=torch.float64, device="cuda")
    rh0 = R.regrid_store(mesh, grid, R.REGRIDMETHOD_BILINEAR)
    want = rh0.regrid(src.view(-1), nlev=nlev, nfields=nf).clone()
    rh0.release()
    mesh.destroy()                                               # drops the parked handle: the next Store builds its own
    mesh = R.Mesh.from_mpas(m)
    rh = R.regrid_store(mesh, grid, R.REGRIDMETHOD_BILINEAR)
    c = MC.Comm(0, 1)
    assert np.array_equal(c.allgather(np.arange(5, dtype=np.int64)), np.arange(5, dtype=np.int64)[None])
    h = MC.Halo(c, rh, m.nCells)
    assert h.mode == expect_mode and rh.n_src == h.n_local and h.sent_per_row == 0 and h.received_per_row == 0
    for dt in (torch.float64, torch.float32):
        s = src.to(dt)
        local = torch.full((nf * nlev, h.n_local), float("nan"), dtype=dt, device="cuda")
        if h.mode == "range":                                    # own block in place inside the local slab
            own = local[:, h.own_pos[0]:h.own_pos[1]]
            own.copy_(s[:, h.own[0]:h.own[1]])
            assert h.base + h.own_pos[0] == h.own[0]
        else:
            own = s[:, h.own[0]:h.own[1]].contiguous()
        h.exchange(own, local)
        torch.cuda.synchronize()
        assert not torch.isnan(local).any()
        if h.mode == "range":
            # the own block held in a SEPARATE buffer (what INTEGRATION.md's Fortran sequence passes): it is copied to its
            # place in the local slab; round 3 left those columns untouched and the Regrid read uninitialised memory
            own2 = s[:, h.own[0]:h.own[1]].contiguous()
            local2 = torch.full_like(local, float("nan"))
            h.exchange(own2, local2)
            torch.cuda.synchronize()
            assert torch.equal(local2, local)
            if h.own[1] - h.own[0] > 2 and nf * nlev > 1:        # a view into the slab that is NOT the in-place one: refused
                from mpassit_amd import _lib
                with pytest.raises(_lib.MpgError, match="overlaps"):
                    h.exchange(local2[:, h.own_pos[0] + 1:h.own_pos[1] + 1], local2)
        got = rh.regrid_typed(local.view(-1), nlev=nlev, nfields=nf, out_dtype=torch.float64)
        ref = want if dt == torch.float64 else None
        if ref is None:                                          # float32 sources: against the same values widened on the host side
            mesh2 = R.Mesh.from_mpas(m)
            r2 = R.regrid_store(mesh2, grid, R.REGRIDMETHOD_BILINEAR)
            ref = r2.regrid_typed(s.view(-1), nlev=nlev, nfields=nf, out_dtype=torch.float64)
            r2.release()
            mesh2.destroy()
        assert torch.equal(got, ref)
    # ESMF_FieldGather with one rank: the block IS the field
    rows = torch.randn((nlev, g.ny, g.nx), dtype=torch.float32, device="cuda")
    assert torch.equal(c.gather_rows(rows, 0, g.ny, g.ny), rows)
    h.destroy()
    c.destroy()
    rh.release()
    mesh.destroy()
    grid.destroy()


def test_range_form_on_a_banded_mesh(gpu_lib, regional_case):
    m, g = regional_case
    _one(gpu_lib, m, g, "range")


def test_compact_form_on_a_shuffled_mesh(gpu_lib, regional_case):
    from mpassit_amd import synth
    m, g = regional_case
    _one(gpu_lib, synth.shuffle_cells(m, seed=11), g, "compact", rows=(20, 55))   # a row block: 40 % of the (scattered) ids
