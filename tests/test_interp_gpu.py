"""End-to-end hot path on the GPU: the host mirror of interp.F90 (`interp_data`, default parm lists,
wrf_mod_vars=.true.) against the same sequence composed from oracle primitives; plus the 2-rank row-sharded
flow with HIP kernels on the card (gloo transport, both ranks on cuda:0)."""
import os
import socket

import numpy as np
import pytest

from conftest import mesh_xyz
from test_fields import HIST_2D, HIST_3D, SOIL

pytestmark = pytest.mark.gpu
RTOL = 1e-6  # north_star tolerance (float bilinear / conservative); nearest and rotation are bit-exact


def rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def test_interp_data_default_lists(gpu_lib, oracle, regional_case):
    from mpassit_amd import interp as I, regrid as R, synth
    o = oracle
    m, g = regional_case
    nz, nzp1, nsoil = 6, 7, 4
    rng = np.random.default_rng(11)
    inp = I.InputData(nz=nz, nzp1=nzp1, nsoil=nsoil)
    inp.hgt = synth.analytic_field(m.latCell, m.lonCell, 1, seed=1)[0] * 1000
    for k, (n, _) in enumerate(HIST_2D):
        if n == "xland":
            inp.hist[n] = synth.category_field(m.nCells, 1, ncat=3)[0]
        elif n in ("snow", "snowh"):
            inp.hist[n] = synth.snow_field(m.latCell, m.lonCell)[0] * (1 + k)
        else:
            inp.hist[n] = synth.analytic_field(m.latCell, m.lonCell, 1, seed=20 + k)[0] + 280
    for k, (n, _) in enumerate(HIST_3D):
        inp.hist[n] = synth.analytic_field(m.latCell, m.lonCell, nzp1 if n in ("zgrid", "w") else nz, seed=40 + k)
    for k, (n, _) in enumerate(SOIL):
        inp.hist[n] = rng.uniform(0, 1, (nsoil, m.nCells))
    diag = [("rainc", "RAINC"), ("u10", "U10"), ("v10", "V10"), ("refl10cm", "REFL_10CM"), ("t2m", "T2")]
    for k, (n, _) in enumerate(diag):
        inp.diag[n] = synth.analytic_field(m.latCell, m.lonCell, nz if n == "refl10cm" else 1, seed=70 + k)
        if n != "refl10cm":
            inp.diag[n] = inp.diag[n][0]
    cfg = I.InterpConfig(wrf_mod_vars=True, keep_mass_winds=True, diag_list=diag, hist_2d=HIST_2D, hist_3d=HIST_3D, hist_soil=SOIL)
    mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_target(g)
    out = I.interp_data(mesh, grid, g, inp, cfg)
    mesh.destroy()
    grid.destroy()

    # ---- the same sequence from oracle primitives
    cxyz, vxyz = mesh_xyz(o, m)
    tri, _ = o.dual_triangles(m.verticesOnCell, m.nVertices, cxyz)
    pxyz = o.lonlat_deg_to_xyz(g.lon, g.lat)
    bi, bw = o.bilinear_weights(cxyz, tri, pxyz)
    ni = o.nearest(cxyz, pxyz)
    rp, col, val = o.conserve(m.verticesOnCell, vxyz, g.nx, g.ny, o.lonlat_deg_to_xyz(g.lon_c, g.lat_c))
    shp = (g.ny, g.nx)

    def bil(x, L):
        return o.apply_fixed(bi, bw, np.atleast_2d(x), L).reshape((L,) + shp)
    assert rel(out["HGT"], bil(inp.hgt, 1)[0]) < 1e-12
    for n, t in HIST_2D:
        if n in ("snow", "snowh"):
            want = o.apply_csr(rp, col, val, inp.hist[n][None, :], 1).reshape(shp)
            assert rel(out[t], want) < 1e-11 < RTOL
        elif n == "xland":
            assert np.array_equal(out[t], o.apply_nearest(ni, inp.hist[n][None, :], 1).reshape(shp))   # bit exact
        else:
            assert rel(out[t], bil(inp.hist[n], 1)[0]) < 1e-12
    for n, t in HIST_3D:
        if n in ("uReconstructZonal", "uReconstructMeridional"):
            continue
        L = nzp1 if n in ("zgrid", "w") else nz
        assert out[t].shape == (L,) + shp and rel(out[t], bil(inp.hist[n], L)) < 1e-12
    um, vm = o.rotate_winds(g.cosa, g.sina, bil(inp.hist["uReconstructZonal"], nz), bil(inp.hist["uReconstructMeridional"], nz))
    assert rel(out["UMASS"], um) < 1e-12 and rel(out["VMASS"], vm) < 1e-12
    cen = o.lonlat_deg_to_xyz(g.lon, g.lat)
    for name, src, st, lon, lat in (("U", um, 1, g.lon_u, g.lat_u), ("V", vm, 2, g.lon_v, g.lat_v)):
        gi, gw = o.grid_bilinear(g.nx, g.ny, cen, st, o.lonlat_deg_to_xyz(lon, lat))
        want = o.apply_fixed(gi, gw, src.reshape(nz, -1), nz).reshape((nz,) + lon.shape)
        assert out[name].shape == want.shape                       # U on EDGE1 (nx+1), V on EDGE2 (ny+1)
        interior = np.zeros(lon.shape, bool)
        interior[1:-1, 1:-1] = True                                # outer ring: implementation-defined (App. A4)
        assert np.abs(out[name][:, interior] - want[:, interior]).max() < 1e-9
    for n, t in SOIL:     # soil follows the last method set = nearest (App. C3): bit-exact copies
        assert np.array_equal(out[t], o.apply_nearest(ni, inp.hist[n], nsoil).reshape((nsoil,) + shp))
    u10, v10 = o.rotate_winds(g.cosa, g.sina, bil(inp.diag["u10"], 1), bil(inp.diag["v10"], 1))
    assert rel(out["U10"], u10[0]) < 1e-12 and rel(out["V10"], v10[0]) < 1e-12
    assert rel(out["REFL_10CM"], bil(inp.diag["refl10cm"], nz)) < 1e-12 and rel(out["T2"], bil(inp.diag["t2m"], 1)[0]) < 1e-12


def test_interp_device_resident_matches_host_path(gpu_lib, regional_case):
    import torch
    from mpassit_amd import interp as I, regrid as R, synth
    m, g = regional_case
    nz = 5
    h2 = [("skintemp", "TSK"), ("xland", "XLAND")]
    h3 = [("theta", "T"), ("qv", "QVAPOR")]
    host = I.InputData(nz=nz, nzp1=nz + 1, nsoil=1)
    host.hgt = synth.analytic_field(m.latCell, m.lonCell, 1, seed=3)[0]
    host.hist = {"skintemp": synth.analytic_field(m.latCell, m.lonCell, 1, seed=4)[0], "xland": synth.category_field(m.nCells)[0],
                 "theta": synth.analytic_field(m.latCell, m.lonCell, nz, seed=5), "qv": synth.analytic_field(m.latCell, m.lonCell, nz, seed=6)}
    dev = I.InputData(nz=nz, nzp1=nz + 1, nsoil=1, hgt=torch.as_tensor(host.hgt, device="cuda"),
                      hist={k: torch.as_tensor(v, device="cuda") for k, v in host.hist.items()})
    cfg = I.InterpConfig(interp_diag=False, hist_2d=h2, hist_3d=h3)
    mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_target(g)
    a = I.interp_data(mesh, grid, g, host, cfg)
    b = I.interp_data(mesh, grid, g, dev, cfg)
    torch.cuda.synchronize()
    for k in a:
        assert np.array_equal(a[k], b[k].cpu().numpy()), k
    # MPAS file order (level-fastest) inputs: fused-transpose kernel, same results to rounding
    lf = I.InputData(nz=nz, nzp1=nz + 1, nsoil=1, hgt=host.hgt, layout=R.LAYOUT_LEV_FAST,
                     hist={k: (np.ascontiguousarray(v.T) if v.ndim == 2 else v) for k, v in host.hist.items()})
    c = I.interp_data(mesh, grid, g, lf, cfg)
    for k in a:
        assert rel(c[k], a[k]) < 1e-14, k
    mesh.destroy()
    grid.destroy()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_main(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=__import__("datetime").timedelta(seconds=90))
    try:
        from mpassit_amd import _lib, dist as mdist, regrid as R, synth, workloads
        _lib.init(0)
        m, g, nlev, _ = workloads.workload("tiny")

        def ago(obj):
            out = [None] * world
            dist.all_gather_object(out, obj)
            return out
        sr = mdist.ShardedRegrid(m, g, R.REGRIDMETHOD_BILINEAR, rank, world, ago)
        F = 2
        full = np.stack([synth.analytic_field(m.latCell, m.lonCell, nlev, seed=s) for s in (1, 2)]).reshape(F * nlev, -1)
        c0, c1 = sr.sched.own
        local = sr.local_buffer(F, nlev, "cuda")
        local.fill_(float("nan"))
        own = sr.own_view(local)
        own.copy_(torch.as_tensor(full[:, c0:c1]))
        # gloo moves CPU tensors: stage the halo through the host (RCCL moves device tensors in bench.py)
        own_c, local_c = own.cpu(), local.cpu()
        sr.sched.exchange(own_c, local_c)
        local.copy_(local_c)
        out = sr.rh.regrid(local.view(-1), nlev=nlev, nfields=F)
        torch.cuda.synchronize()
        q.put((rank, sr.j0, sr.j1, out.cpu().numpy(), sr.sched.mode))
        sr.destroy()
        _lib.finalize()
    finally:
        dist.destroy_process_group()


def test_two_ranks_row_sharded_on_gpu(gpu_lib, oracle):
    """2 ranks (both on cuda:0), row-sharded target, range halo; stitched result == single-GPU result."""
    import torch.multiprocessing as mp
    from mpassit_amd import regrid as R, synth, workloads
    m, g, nlev, _ = workloads.workload("tiny")
    mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_target(g)
    rh = R.regrid_store(mesh, grid, R.REGRIDMETHOD_BILINEAR)
    full = np.stack([synth.analytic_field(m.latCell, m.lonCell, nlev, seed=s) for s in (1, 2)])
    want = rh.regrid(full, nlev=nlev, nfields=2)
    rh.release()
    mesh.destroy()
    grid.destroy()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(2)), key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][4] == "range"
    got = np.concatenate([r[3] for r in res], axis=2)
    assert got.shape == want.shape
    assert np.array_equal(got, want)
