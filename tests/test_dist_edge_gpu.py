"""Row-sharded path, edge case: a rank whose target rows lie entirely outside the mesh footprint (nothing mapped,
empty halo, zero-filled output) next to ranks that do real work; 3 ranks on one card over gloo."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _case():
    from mpassit_amd import synth, target_grid as tg
    g = tg.define_target_grid_params("lambert", 121, 91, dx=30000.0, dy=30000.0, ref_lat=38.5, ref_lon=-97.5, truelat1=38.5,
                                     truelat2=38.5, stand_lon=-97.5)
    # mesh covers only the lower ~55 % of the target rows: the top third of the rows (rank 2 of 3) sees no mesh at all
    m = synth.regional_mesh_for_lambert(g.proj, 121, 50, 12000, margin=0.0)
    return m, g


def _rank(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=__import__("datetime").timedelta(seconds=90))
    try:
        from mpassit_amd import _lib, dist as mdist, regrid as R, synth
        _lib.init(0)
        m, g = _case()

        def ago(obj):
            out = [None] * world
            dist.all_gather_object(out, obj)
            return out
        sr = mdist.ShardedRegrid(m, g, R.REGRIDMETHOD_BILINEAR, rank, world, ago)
        nlev = 3
        full = synth.analytic_field(m.latCell, m.lonCell, nlev, seed=5)
        c0, c1 = sr.sched.own
        local = sr.local_buffer(1, nlev, "cuda")
        local.fill_(float("nan"))
        own = sr.own_view(local)
        own.copy_(torch.as_tensor(np.ascontiguousarray(full[:, c0:c1])))
        sr.sched.exchange(own, local, pack_fn=sr._pack)
        out = sr.rh.regrid(local.view(-1), nlev=nlev)
        torch.cuda.synchronize()
        q.put((rank, sr.j0, sr.j1, out.cpu().numpy(), sr.sched.mode, sr.n_needed, sr.sched.n_local))
        sr.destroy()
        _lib.finalize()
    finally:
        dist.destroy_process_group()


def test_rank_without_sources(gpu_lib):
    import torch.multiprocessing as mp
    from mpassit_amd import regrid as R, synth
    m, g = _case()
    mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_target(g)
    rh = R.regrid_store(mesh, grid, R.REGRIDMETHOD_BILINEAR)
    want = rh.regrid(synth.analytic_field(m.latCell, m.lonCell, 3, seed=5), nlev=3)
    rh.release()
    mesh.destroy()
    grid.destroy()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(3)), key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[2][5] == 0 and res[2][6] == 0            # the top shard references no source cell at all
    assert (res[2][3] == 0.0).all()                      # ... and is zero-filled like the reference's unmapped points
    assert all(r[4] == "range" for r in res)
    got = np.concatenate([r[3] for r in res], axis=2)
    assert np.array_equal(got, want)


def _rank_file_order(rank, world, port, shuffle, q, ownership="aligned", decomp_file=None):
    """float32 sources in MPAS file order on every rank's GPU (what the shipped driver holds, input_data.F90:630-655)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=__import__("datetime").timedelta(seconds=90))
    try:
        from mpassit_amd import _lib, dist as mdist, regrid as R, synth
        _lib.init(0)
        m, g = _case()
        if shuffle:
            m = synth.shuffle_cells(m, seed=3)

        def ago(obj):
            out = [None] * world
            dist.all_gather_object(out, obj)
            return out
        sr = mdist.ShardedRegrid(m, g, R.REGRIDMETHOD_BILINEAR, rank, world, ago, ownership=ownership, decomp_file=decomp_file)
        nlev, nf = 5, 2
        full = np.stack([synth.analytic_field(m.latCell, m.lonCell, nlev, seed=5 + f) for f in range(nf)]).transpose(0, 2, 1).astype(np.float32)
        c0, c1 = sr.sched.own
        local = sr.local_buffer(nf, nlev, "cuda", dtype=torch.float32, layout=R.LAYOUT_LEV_FAST)
        local.fill_(float("nan"))
        sel = sr.sched.owned_ids if sr.sched.mode == "owned" else slice(c0, c1)
        mine = torch.as_tensor(np.ascontiguousarray(full[:, sel]), device="cuda")
        if sr.sched.mode == "range":
            own = sr.own_view(local)
            own.copy_(mine)
        else:
            own = mine
        out = sr.step(own, local, nlev, nf)
        torch.cuda.synchronize()
        plan = sr.sched.plan(nf, local)
        q.put((rank, sr.j0, sr.j1, out.cpu().numpy(), sr.sched.mode, plan.bytes_received, sr.n_needed))
        sr.destroy()
        _lib.finalize()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,shuffle", [(2, False), (3, False), (3, True)])
def test_file_order_float32_sources_sharded(gpu_lib, world, shuffle):
    """The halo exchange on [field][cell][level] float32 slabs (range form: one byte range per neighbour and field; compact form:
    mpg_pack_rows_dev on whole rows): the row blocks of 2 / 3 ranks equal the single-GPU float32 file-order Regrid bit for bit."""
    import torch
    import torch.multiprocessing as mp
    from mpassit_amd import regrid as R, synth
    m, g = _case()
    if shuffle:
        m = synth.shuffle_cells(m, seed=3)
    nlev, nf = 5, 2
    full = np.stack([synth.analytic_field(m.latCell, m.lonCell, nlev, seed=5 + f) for f in range(nf)]).transpose(0, 2, 1).astype(np.float32)
    mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_target(g)
    rh = R.regrid_store(mesh, grid, R.REGRIDMETHOD_BILINEAR)
    want = rh.regrid_typed(torch.as_tensor(np.ascontiguousarray(full), device="cuda").view(-1), nlev=nlev, nfields=nf, layout=R.LAYOUT_LEV_FAST).cpu().numpy()
    rh.release()
    mesh.destroy()
    grid.destroy()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_file_order, args=(r, world, port, shuffle, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(r[4] == ("compact" if shuffle else "range") for r in res)
    got = np.concatenate([r[3] for r in res], axis=2)
    assert got.dtype == np.float32 and np.array_equal(got, want)


@pytest.mark.parametrize("ownership", ["need", "auto"])
def test_owned_form_follows_the_rows_on_a_shuffled_mesh(gpu_lib, ownership):
    """Round 5: a mesh WITHOUT banded numbering sharded with the owned halo form (every cell to the lowest rank whose rows reference it;
    `auto` picks it when the needed ids are not banded): the same bits as the single-GPU float32 file-order Regrid, and only a small
    part of what the ranks need travels (equal id blocks moved two thirds of it: test_file_order_float32_sources_sharded[3-True])."""
    import torch
    import torch.multiprocessing as mp
    from mpassit_amd import regrid as R, synth
    m, g = _case()
    m = synth.shuffle_cells(m, seed=3)
    nlev, nf, world = 5, 2, 3
    full = np.stack([synth.analytic_field(m.latCell, m.lonCell, nlev, seed=5 + f) for f in range(nf)]).transpose(0, 2, 1).astype(np.float32)
    mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_target(g)
    rh = R.regrid_store(mesh, grid, R.REGRIDMETHOD_BILINEAR)
    want = rh.regrid_typed(torch.as_tensor(np.ascontiguousarray(full), device="cuda").view(-1), nlev=nlev, nfields=nf, layout=R.LAYOUT_LEV_FAST).cpu().numpy()
    rh.release()
    mesh.destroy()
    grid.destroy()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_file_order, args=(r, world, port, True, q, ownership)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(r[4] == "owned" for r in res)
    assert np.array_equal(np.concatenate([r[3] for r in res], axis=2), want)
    received = sum(r[5] for r in res) / (nf * nlev * 4)                 # cells that crossed between ranks
    assert 0 < received < 0.2 * sum(r[6] for r in res)                    # a small part of what the ranks need


def test_block_decomp_file_partitions_the_sources(gpu_lib, tmp_path):
    """The namelist's block_decomp_file (an MPAS graph partition file; model_grid.F90:423-438, 2367-2426) as the partition of the source
    cells of a sharded run: the model's decomposition -- here a random one, the hardest case -- through the owned halo form; the
    row blocks of three ranks equal the single-GPU float32 file-order Regrid bit for bit."""
    import torch
    import torch.multiprocessing as mp
    from mpassit_amd import regrid as R, synth
    m, g = _case()
    nlev, nf, world = 5, 2, 3
    owner = np.random.default_rng(9).integers(0, world, m.nCells)
    f = tmp_path / "mesh.graph.info.part.3"
    f.write_text("\n".join("%d" % p for p in owner) + "\n")
    full = np.stack([synth.analytic_field(m.latCell, m.lonCell, nlev, seed=5 + k) for k in range(nf)]).transpose(0, 2, 1).astype(np.float32)
    mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_target(g)
    rh = R.regrid_store(mesh, grid, R.REGRIDMETHOD_BILINEAR)
    want = rh.regrid_typed(torch.as_tensor(np.ascontiguousarray(full), device="cuda").view(-1), nlev=nlev, nfields=nf, layout=R.LAYOUT_LEV_FAST).cpu().numpy()
    rh.release()
    mesh.destroy()
    grid.destroy()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_file_order, args=(r, world, port, False, q, "aligned", str(f))) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(r[4] == "owned" for r in res)
    assert np.array_equal(np.concatenate([r[3] for r in res], axis=2), want)
