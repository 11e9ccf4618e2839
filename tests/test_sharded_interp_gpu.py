"""Row-sharded whole `interp_data` (default lists, wrf_mod_vars: rotation + staggered U/V with the one-row halo
exchange) on 3 ranks sharing one card: the stitched result equals the single-GPU result bit for bit."""
import os
import socket

import numpy as np
import pytest

from test_fields import HIST_2D, HIST_3D, SOIL

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _inputs(m):
    from mpassit_amd import interp as I, synth
    nz, nsoil = 5, 3
    rng = np.random.default_rng(3)
    inp = I.InputData(nz=nz, nzp1=nz + 1, nsoil=nsoil, hgt=synth.analytic_field(m.latCell, m.lonCell, 1, seed=1)[0])
    for k, (n, _) in enumerate(HIST_2D):
        inp.hist[n] = synth.category_field(m.nCells)[0] if n == "xland" else synth.analytic_field(m.latCell, m.lonCell, 1, seed=10 + k)[0]
    for k, (n, _) in enumerate(HIST_3D):
        inp.hist[n] = synth.analytic_field(m.latCell, m.lonCell, nz + 1 if n in ("zgrid", "w") else nz, seed=30 + k)
    for n, _ in SOIL:
        inp.hist[n] = rng.uniform(0, 1, (nsoil, m.nCells))
    cfg = I.InterpConfig(interp_diag=False, wrf_mod_vars=True, hist_2d=HIST_2D, hist_3d=HIST_3D, hist_soil=SOIL)
    return inp, cfg


def _to_device(inp):
    import torch
    inp.hgt = torch.as_tensor(inp.hgt, device="cuda")
    for k in list(inp.hist):
        inp.hist[k] = torch.as_tensor(np.ascontiguousarray(inp.hist[k]), device="cuda")
    return inp


def _rank(rank, world, port, q, device=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import datetime
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=90))
    try:
        from mpassit_amd import _lib, dist as mdist, workloads
        _lib.init(0)
        m, g, _, _ = workloads.workload("tiny")
        inp, cfg = _inputs(m)
        if device:
            inp = _to_device(inp)
        si = mdist.ShardedInterp(m, g, rank, world)
        out = si.interp_data(inp, cfg)
        q.put((rank, si.j0, si.j1, {k: np.asarray(v.cpu() if device else v) for k, v in out.items()}))
        si.destroy()
        _lib.finalize()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("device", [False, True], ids=["host_arrays", "device_resident"])
def test_sharded_interp_equals_single_gpu(gpu_lib, device):
    """device_resident (round 6): the sharded driver's one-pass wind chain -- halo rows of the UNROTATED mass winds, rotation and both
    destaggerings in mpg_wind_destagger_dev on the extended block -- against the single GPU's; host arrays take the three-call route."""
    import torch.multiprocessing as mp
    from mpassit_amd import interp as I, regrid as R, workloads
    m, g, _, _ = workloads.workload("tiny")
    inp, cfg = _inputs(m)
    if device:
        inp = _to_device(inp)
    mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_target(g)
    want = I.interp_data(mesh, grid, g, inp, cfg)
    if device:
        want = {k: v.cpu().numpy() for k, v in want.items()}
    mesh.destroy()
    grid.destroy()
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank, args=(r, world, port, q, device)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=150) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for k, w in want.items():
        parts = [r[3][k] for r in res]
        got = np.concatenate(parts, axis=w.ndim - 2)     # rows are the second-to-last axis
        assert got.shape == w.shape, (k, got.shape, w.shape)
        assert np.array_equal(got, w), k
