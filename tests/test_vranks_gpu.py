"""The RCCL path of the C-ABI REHEARSED ON ONE GPU with virtual ranks (mpg_comm_virtual): V host threads, one per virtual rank,
make the calls V real ranks make -- mpg_halo_build (votes and counts through ncclAllGather on the set-up stream), then
mpg_halo_exchange_dev and mpg_gather_rows, whose transfers go out as ONE ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd
to self with each rank's own device pointers (sendbuf + soff * nrows * es, recvbuf + roff * nrows * es, the recv_a pitches of the
unpack), ordered against each rank's stream by events.  Checked: every rank's local slab holds the whole field's bytes at every id its rows reference,
the Regrid of the slabs equals the single-GPU Regrid bit for bit, the gathered field equals it too, the schedules equal
mpg_halo_plan_host's (itself checked against dist.HaloSchedule on the CPU, tests/test_comm_plan.py), and the group's counters say
the sends and receives were really issued.  What this cannot show: bytes crossing xGMI (the box has one card)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _case(rows=97):
    from mpassit_amd import synth, target_grid as tg
    g = tg.define_target_grid_params("lambert", 151, rows, dx=30000.0, dy=30000.0, ref_lat=38.5, ref_lon=-97.5, truelat1=38.5, truelat2=38.5,
                                     stand_lon=-97.5)
    m = synth.regional_mesh_for_lambert(g.proj, 141, rows - 10, 24000, margin=0.0)
    return m, g


def _rehearse(m, g, V, expect_mode, dtype, lev_fast, own_streams, ownership="aligned"):
    import torch

    from mpassit_amd import comm as MC, dist as mdist, regrid as R
    nlev, nf = 6, 2
    layout = R.LAYOUT_LEV_FAST if lev_fast else R.LAYOUT_CELL_FAST
    rng = np.random.default_rng(5)
    full = torch.as_tensor(rng.standard_normal((nf, nlev, m.nCells)), device="cuda").to(dtype)            # [F][L][n]
    src = full.permute(0, 2, 1).contiguous() if lev_fast else full.contiguous()
    # the single-GPU answer
    mesh0, grid0 = R.Mesh.from_mpas(m), R.Grid.from_target(g)
    rh0 = R.regrid_store(mesh0, grid0, R.REGRIDMETHOD_BILINEAR)
    want = rh0.regrid_typed(src.view(-1), nlev=nlev, nfields=nf, layout=layout).clone()                   # [F][L][ny][nx], the sources' type
    rh0.release()
    mesh0.destroy()
    grid0.destroy()
    # every virtual rank's objects (made on this thread, one after the other: Stores are not part of the rehearsal)
    real = MC.Comm(0, 1)
    vcs = [real.virtual(r, V) for r in range(V)]
    blocks = [mdist.row_block(g.ny, V, r) for r in range(V)]
    grids = [R.Grid.from_target(g, rows=b) for b in blocks]
    meshes = [R.Mesh.from_mpas(m, window_grid=grids[r]) for r in range(V)]
    rhs = [R.regrid_store(meshes[r], grids[r], R.REGRIDMETHOD_BILINEAR) for r in range(V)]
    needed = [rh.unique_sources() for rh in rhs]
    torch.cuda.synchronize()

    def body(r):
        torch.cuda.set_device(0)
        st = torch.cuda.Stream() if own_streams else torch.cuda.current_stream()
        h = MC.Halo(vcs[r], rhs[r], m.nCells, ownership)
        with torch.cuda.stream(st):
            shape = (nf, h.n_local, nlev) if lev_fast else (nf * nlev, h.n_local)
            local = torch.full(shape, float("nan"), dtype=dtype, device="cuda")
            c0, c1 = h.own
            mine = src[:, c0:c1] if lev_fast else src.view(nf * nlev, -1)[:, c0:c1]
            if h.mode == "range" and r % 2 == 0:                     # own block in place inside the slab ...
                own = local[:, h.own_pos[0]:h.own_pos[1]]
                own.copy_(mine)
            else:                                                    # ... or in a buffer of its own
                own = mine.contiguous()
            h.exchange(own, local, stream=st.cuda_stream)
            out = rhs[r].regrid_typed(local.view(-1), nlev=nlev, nfields=nf, layout=layout)
            gathered = [vcs[r].gather_rows(out[f], blocks[r][0], blocks[r][1], g.ny, root=V - 1, stream=st.cuda_stream) for f in range(nf)]
        st.synchronize()
        info = dict(mode=h.mode, n_local=h.n_local, own=h.own, base=h.base, own_pos=h.own_pos, sent=h.sent_per_row, received=h.received_per_row)
        h.destroy()
        return info, local, out, gathered

    res = MC.run_virtual_ranks(V, body)
    groups, sends, recvs, allgathers = real.virtual_stats()
    for r, (info, local, out, gathered) in enumerate(res):
        assert expect_mode is None or info["mode"] == expect_mode
        expect = info["mode"]
        plan = MC.plan_host(r, needed, m.nCells, ownership)          # the pure schedule (== dist.HaloSchedule, tests/test_comm_plan.py)
        assert (plan["mode"], plan["n_local"], tuple(plan["own"]), plan["base"], tuple(plan["own_pos"])) == (
            info["mode"], info["n_local"], tuple(info["own"]), info["base"], tuple(info["own_pos"]))
        ids_local = np.arange(info["base"], info["base"] + info["n_local"]) if expect == "range" else needed[r]
        pos = torch.as_tensor(np.searchsorted(ids_local, needed[r]), device="cuda")
        ids = torch.as_tensor(needed[r].astype(np.int64), device="cuda")
        if lev_fast:
            assert torch.equal(local[:, pos], src[:, ids])
        else:
            assert torch.equal(local[:, pos], src.view(nf * nlev, -1)[:, ids])
        j0, j1 = blocks[r]
        assert torch.equal(out, want[:, :, j0:j1])                   # Regrid of the exchanged slab == single-GPU Regrid, bit for bit
        if r == V - 1:
            for f in range(nf):
                assert torch.equal(gathered[f], want[f])             # ESMF_FieldGather (write_data.F90:1006) through the same group calls
        else:
            assert all(x is None for x in gathered)
    sent_total = sum(i["sent"] for i, *_ in res)
    assert sent_total == sum(i["received"] for i, *_ in res) and (sent_total > 0 or expect_mode is None)
    # what really went through RCCL: the all-gathers of the V halo builds and nf gathers, one group per exchange and per gather
    assert allgathers >= 2 + nf and groups in (nf, 1 + nf) and sends == recvs and (expect_mode is None or sends >= (V - 1) + nf * (V - 1) * nlev)
    for rh, me, gr, vc in zip(rhs, meshes, grids, vcs):
        rh.release()
        me.destroy()
        gr.destroy()
        vc.destroy()
    real.destroy()
    return dict(groups=groups, sends=sends, allgathers=allgathers)


@pytest.mark.parametrize("V", [2, 3, 8])
def test_range_form_float64_cell_fast(gpu_lib, V):
    import torch
    m, g = _case()
    _rehearse(m, g, V, "range", torch.float64, lev_fast=False, own_streams=False)


@pytest.mark.parametrize("V,own_streams", [(2, False), (3, True), (8, True)])
def test_range_form_float32_file_order(gpu_lib, V, own_streams):
    """What the shipped driver holds: float32 rows in MPAS file order -- a neighbour's strip is one byte range per field."""
    import torch
    m, g = _case()
    _rehearse(m, g, V, "range", torch.float32, lev_fast=True, own_streams=own_streams)


@pytest.mark.parametrize("V,dtype,lev_fast", [(2, "float64", False), (3, "float32", False), (3, "float32", True), (5, "float64", True)])
def test_compact_form_on_a_shuffled_mesh(gpu_lib, V, dtype, lev_fast):
    import torch

    from mpassit_amd import synth
    m, g = _case()
    _rehearse(synth.shuffle_cells(m, seed=11), g, V, "compact", getattr(torch, dtype), lev_fast=lev_fast, own_streams=True)


@pytest.mark.parametrize("V,partition,dtype,lev_fast", [(3, "need", "float64", False), (5, "need", "float32", True), (4, "random", "float32", False),
                                                        (3, "random", "float64", True)])
def test_owned_form_the_callers_own_partition(gpu_lib, V, partition, dtype, lev_fast):
    """mpg_halo_build_owned through the real RCCL group calls: the cells of a SHUFFLED mesh partitioned by the caller -- `need`: every
    cell to the lowest rank whose rows reference it (what bench.py does for meshes without banded numbering); `random`: an arbitrary
    assignment, as a coupled model's own decomposition looks to this library.  Every rank's slab holds the field at its needed ids,
    its Regrid equals the single-GPU Regrid bit for bit, and with `need` only what neighbouring row blocks share has travelled."""
    import torch

    from mpassit_amd import synth
    m0, g = _case()
    _rehearse_owned(synth.shuffle_cells(m0, seed=11), g, V, partition, getattr(torch, dtype), lev_fast)


def _rehearse_owned(m, g, V, partition, dt, lev_fast, nlev=6, nf=2, strict=True):
    import torch

    from mpassit_amd import comm as MC, dist as mdist, regrid as R
    layout = R.LAYOUT_LEV_FAST if lev_fast else R.LAYOUT_CELL_FAST
    full = torch.as_tensor(np.random.default_rng(5).standard_normal((nf, nlev, m.nCells)), device="cuda").to(dt)
    src = full.permute(0, 2, 1).contiguous() if lev_fast else full.contiguous().view(nf * nlev, -1)
    mesh0, grid0 = R.Mesh.from_mpas(m), R.Grid.from_target(g)
    rh0 = R.regrid_store(mesh0, grid0, R.REGRIDMETHOD_BILINEAR)
    want = rh0.regrid_typed(src.view(-1), nlev=nlev, nfields=nf, layout=layout).clone()
    rh0.release()
    mesh0.destroy()
    grid0.destroy()
    real = MC.Comm(0, 1)
    vcs = [real.virtual(r, V) for r in range(V)]
    blocks = [mdist.row_block(g.ny, V, r) for r in range(V)]
    grids = [R.Grid.from_target(g, rows=b) for b in blocks]
    meshes = [R.Mesh.from_mpas(m, window_grid=grids[r]) for r in range(V)]
    rhs = [R.regrid_store(meshes[r], grids[r], R.REGRIDMETHOD_BILINEAR) for r in range(V)]
    needed = [rh.unique_sources() for rh in rhs]
    if partition == "need":
        owned, seen = [], np.zeros(m.nCells, bool)
        for r in range(V):
            mine = needed[r][~seen[needed[r]]]
            seen[mine] = True
            owned.append(mine.astype(np.int32))
    else:
        owner = np.random.default_rng(3).integers(0, V, m.nCells)
        owned = [np.nonzero(owner == r)[0].astype(np.int32) for r in range(V)]
    torch.cuda.synchronize()

    def body(r):
        torch.cuda.set_device(0)
        st = torch.cuda.Stream()
        h = MC.Halo(vcs[r], rhs[r], m.nCells, owned_ids=owned[r])
        assert h.mode == "owned" and h.n_local == needed[r].size and h.own == (0, owned[r].size)
        with torch.cuda.stream(st):
            ids = torch.as_tensor(owned[r].astype(np.int64), device="cuda")
            own = src[:, ids].contiguous()
            local = torch.full((nf, h.n_local, nlev) if lev_fast else (nf * nlev, h.n_local), float("nan"), dtype=dt, device="cuda")
            h.exchange(own, local, stream=st.cuda_stream)
            out = rhs[r].regrid_typed(local.view(-1), nlev=nlev, nfields=nf, layout=layout)
        st.synchronize()
        res = (h.sent_per_row, h.received_per_row, local, out)
        h.destroy()
        return res

    res = MC.run_virtual_ranks(V, body)
    for r, (sent, recv, local, out) in enumerate(res):
        plan = MC.plan_owned_host(r, needed, owned)
        assert recv == sum(x.size for q, x in enumerate(plan["recv_ids"]) if q != r) and sent == sum(x.size for q, x in enumerate(plan["send_ids"]) if q != r)
        assert torch.equal(local, src[:, torch.as_tensor(needed[r].astype(np.int64), device="cuda")])
        assert torch.equal(out, want[:, :, blocks[r][0]:blocks[r][1]])
    total_sent, total_needed = sum(x[0] for x in res), sum(x.size for x in needed)
    assert total_sent == sum(x[1] for x in res) and (total_sent > 0 or not strict)
    if partition == "need":
        assert total_sent == total_needed - np.unique(np.concatenate(needed)).size and (total_sent < 0.2 * total_needed or not strict)
    groups, sends, recvs, allgathers = real.virtual_stats()
    assert groups in (0, 1) and sends == recvs and (sends > 0 or not strict) and allgathers == 2
    for rh, me, gr, vc in zip(rhs, meshes, grids, vcs):
        rh.release()
        me.destroy()
        gr.destroy()
        vc.destroy()
    real.destroy()


def test_owned_form_refuses_a_partition_that_is_none(gpu_lib, regional_case, monkeypatch):
    """A cell owned twice, a referenced cell owned by nobody, an unsorted list: refused with the cell's number."""
    from mpassit_amd import _lib, comm as MC, regrid as R
    monkeypatch.setenv("MPG_COMM_TIMEOUT_S", "20")
    m, g = regional_case
    mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_target(g)
    rh = R.regrid_store(mesh, grid, R.REGRIDMETHOD_BILINEAR)
    need = rh.unique_sources()
    c = MC.Comm(0, 1)
    with pytest.raises(_lib.MpgError, match="no rank owns"):
        MC.Halo(c, rh, m.nCells, owned_ids=need[1:])
    with pytest.raises(_lib.MpgError, match="sorted"):
        MC.Halo(c, rh, m.nCells, owned_ids=need[::-1])
    h = MC.Halo(c, rh, m.nCells, owned_ids=np.arange(m.nCells, dtype=np.int32))           # one rank owning everything: fine
    assert h.mode == "owned" and h.n_local == need.size and h.sent_per_row == 0
    h.destroy()
    c.destroy()
    rh.release()
    mesh.destroy()
    grid.destroy()


def test_every_rank_refuses_together_when_one_ranks_cell_has_no_owner(gpu_lib, monkeypatch):
    """Two virtual ranks; a cell only rank 1's rows reference is left out of both owned lists.  BOTH ranks come back with the same error at
    once (every rank checks every rank's needs): none goes on alone to wait in an exchange for a peer that has given up."""
    import time

    from mpassit_amd import _lib, comm as MC, dist as mdist, regrid as R
    monkeypatch.setenv("MPG_COMM_TIMEOUT_S", "60")
    m, g = _case(rows=41)
    V = 2
    real = MC.Comm(0, 1)
    vcs = [real.virtual(r, V) for r in range(V)]
    blocks = [mdist.row_block(g.ny, V, r) for r in range(V)]
    grids = [R.Grid.from_target(g, rows=b) for b in blocks]
    meshes = [R.Mesh.from_mpas(m, window_grid=grids[r]) for r in range(V)]
    rhs = [R.regrid_store(meshes[r], grids[r], R.REGRIDMETHOD_BILINEAR) for r in range(V)]
    needed = [rh.unique_sources() for rh in rhs]
    only1 = np.setdiff1d(needed[1], needed[0])
    lost = int(only1[only1.size // 2])
    owned = [needed[0].astype(np.int32), np.setdiff1d(only1, [lost]).astype(np.int32)]
    errs = [None] * V

    def body(r):
        try:
            MC.Halo(vcs[r], rhs[r], m.nCells, owned_ids=owned[r])
        except _lib.MpgError as e:
            errs[r] = str(e)
        return True

    t0 = time.perf_counter()
    MC.run_virtual_ranks(V, body, timeout=120.0)
    assert time.perf_counter() - t0 < 20.0
    assert all(e is not None and "rank 1's rows reference cell %d, which no rank owns" % lost in e for e in errs), errs
    for rh, me, gr, vc in zip(rhs, meshes, grids, vcs):
        rh.release()
        me.destroy()
        gr.destroy()
        vc.destroy()
    real.destroy()


@pytest.mark.parametrize("lev_fast", [False, True])
def test_a_rank_that_needs_nothing_still_serves_its_cells(gpu_lib, lev_fast):
    """The mesh covers the lower half of the grid only and its cells are numbered at random: with equal id blocks (para_range) the rank
    of the top rows references no cell at all -- an EMPTY local space (NULL) -- yet owns a third of the cells the others need and takes
    part in the exchange.  (Found by tools/vranks_soak.py: mpg_halo_exchange_dev refused the NULL.)"""
    import torch

    from mpassit_amd import synth, target_grid as tg
    g = tg.define_target_grid_params("lambert", 121, 91, dx=30000.0, dy=30000.0, ref_lat=38.5, ref_lon=-97.5, truelat1=38.5, truelat2=38.5, stand_lon=-97.5)
    m = synth.shuffle_cells(synth.regional_mesh_for_lambert(g.proj, 121, 50, 12000, margin=0.0), seed=2)
    _rehearse(m, g, 3, "compact", torch.float32, lev_fast=lev_fast, own_streams=True, ownership="para_range")


def test_para_range_ownership(gpu_lib):
    import torch
    m, g = _case()
    _rehearse(m, g, 3, "range", torch.float32, lev_fast=False, own_streams=True, ownership="para_range")


def test_file_order_float64_on_separate_streams(gpu_lib):
    """Every virtual rank on a stream of its own: the group runs on rank 0's stream with event edges from and to the others'."""
    import torch
    m, g = _case()
    _rehearse(m, g, 3, "range", torch.float64, lev_fast=True, own_streams=True)


def test_disagreeing_ranks_fail_the_step(gpu_lib, monkeypatch):
    """Two virtual ranks that disagree about a transfer (the root expects two level segments from its peer, the peer sends three):
    the rendezvous reports it to both instead of letting RCCL match a send with the wrong receive."""
    import torch

    from mpassit_amd import _lib, comm as MC
    monkeypatch.setenv("MPG_COMM_TIMEOUT_S", "20")
    real = MC.Comm(0, 1)
    vcs = [real.virtual(r, 2) for r in range(2)]

    def body(r):
        torch.cuda.set_device(0)
        rows = torch.zeros((2 + r, 4, 8), dtype=torch.float32, device="cuda")       # rank 0: 2 levels, rank 1: 3 levels of the same field
        try:
            vcs[r].gather_rows(rows, 4 * r, 4 * r + 4, 8, root=0)
        except _lib.MpgError as e:
            return str(e)
        return None

    msgs = MC.run_virtual_ranks(2, body)
    assert all(m is not None and "sends" in m for m in msgs), msgs
    for vc in vcs:
        vc.destroy()
    real.destroy()


def test_untagged_launch_of_several_ranks_is_refused(gpu_lib, monkeypatch, tmp_path):
    """mpg_comm_init with nranks > 1 and no MPASSIT_RUN_ID: refused before anything is written or waited for (a file left by
    a crashed launch could not be told from this launch's)."""
    from mpassit_amd import _lib, comm as MC
    monkeypatch.delenv("MPASSIT_RUN_ID", raising=False)
    monkeypatch.delenv("MPG_COMM_ALLOW_UNTAGGED", raising=False)
    with pytest.raises(_lib.MpgError, match="MPASSIT_RUN_ID"):
        MC.Comm(1, 2, str(tmp_path / "x.rcclid"))
