"""The device entry points are stream-ordered and, once a handle is warm (weights stored, tile lists built), enqueue
kernels only -- no allocation, no synchronisation, no host copy -- so a caller can capture its per-time-step sequence of
Regrids, rotations and post-ops into a hipGraph and replay it on new field values (the launch-bound part of a job: the
many 2-D fields)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_regrid_sequence_is_graph_capturable(gpu_lib, regional_case):
    import torch

    from mpassit_amd import post, regrid as R
    m, g = regional_case
    mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_target(g)
    rh_b = R.regrid_store(mesh, grid, R.REGRIDMETHOD_BILINEAR)
    rh_n = R.regrid_store(mesh, grid, R.REGRIDMETHOD_NEAREST_STOD)
    rh_c = R.regrid_store(mesh, grid, R.REGRIDMETHOD_CONSERVE)
    rh_u = R.regrid_store_grid(grid, R.STAGGERLOC_EDGE1)
    nz = 5
    rng = np.random.default_rng(9)
    dev = torch.device("cuda")
    s3 = torch.as_tensor(rng.standard_normal((nz, m.nCells)), device=dev)
    s3b = torch.as_tensor(rng.standard_normal((nz, m.nCells)), device=dev)
    s2 = torch.as_tensor(np.floor(rng.uniform(0, 20, m.nCells)), device=dev)
    cosa, sina = (torch.as_tensor(np.ascontiguousarray(a), device=dev) for a in (g.cosa, g.sina))
    o3 = torch.empty((1, nz, g.ny, g.nx), dtype=torch.float64, device=dev)
    o3b, oc = torch.empty_like(o3), torch.empty((1, 1, g.ny, g.nx), dtype=torch.float64, device=dev)
    on = torch.empty_like(oc)
    ou = torch.empty((1, nz, g.ny, g.nx + 1), dtype=torch.float64, device=dev)
    of = torch.empty((1, nz, g.ny, g.nx), dtype=torch.float32, device=dev)

    def step():
        rh_b.regrid(s3.view(-1), nlev=nz, out=o3)
        rh_b.regrid(s3b.view(-1), nlev=nz, out=o3b)
        R.rotate_winds_cgrid(cosa, sina, o3[0], o3b[0])
        rh_u.regrid(o3.view(-1), nlev=nz, out=ou)
        rh_n.regrid(s2, nlev=1, out=on)
        rh_c.regrid(s2, nlev=1, out=oc)
        rh_b.regrid_typed(s3.view(-1), nlev=nz, out_dtype=torch.float32, offset=-300.0, out=of)

    step()                                            # warm: tile lists, kernel choice
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            step()
    for trial in range(2):                            # replay on new values, compare with the eager calls
        s3.copy_(torch.as_tensor(rng.standard_normal((nz, m.nCells)), device=dev))
        s3b.mul_(-0.5)
        s2.add_(1.0)
        graph.replay()
        torch.cuda.synchronize()
        got = [t.clone() for t in (o3, o3b, ou, on, oc, of)]
        step()
        torch.cuda.synchronize()
        for a, b in zip(got, (o3, o3b, ou, on, oc, of)):
            assert torch.equal(a, b)
    assert float(of.abs().max()) > 0 and float(ou.abs().max()) > 0
    for h in (rh_b, rh_n, rh_c, rh_u):
        h.release()
    mesh.destroy()
    grid.destroy()


@pytest.mark.parametrize("file_order", [False, True])
def test_graphed_interp_replays_a_time_level(gpu_lib, regional_case, file_order):
    """interp.GraphedInterp: the whole interp_data sequence (default lists, wrf_mod_vars, conservative + nearest + soil +
    diag) as one graph; replays on new field values equal the eager pipeline bit for bit.  file_order: float32 fields in MPAS
    file order, as a driver holds them -- the bundles then go out as mpg_regrid_bundle_typed_dev launches (pointer tables in the
    kernels' argument blocks), which must be capturable like any other launch."""
    import torch

    from test_fields import HIST_2D, HIST_3D, SOIL
    from test_fortran_driver import DIAG
    from mpassit_amd import interp as I, regrid as R
    m, g = regional_case
    nz, nsoil = 5, 3
    gen = torch.Generator(device="cuda")
    gen.manual_seed(5)

    def rnd(*shape):
        t = torch.rand(shape, dtype=torch.float64, device="cuda", generator=gen)
        if not file_order:
            return t
        return (t.t().contiguous() if t.ndim == 2 else t).float()       # [nCells][nlev] float32, as the file stores it
    inp = I.InputData(nz=nz, nzp1=nz + 1, nsoil=nsoil, hgt=torch.rand(m.nCells, dtype=torch.float64, device="cuda", generator=gen),
                      layout=R.LAYOUT_LEV_FAST if file_order else R.LAYOUT_CELL_FAST)
    for n, _ in HIST_2D:
        inp.hist[n] = torch.floor(rnd(m.nCells) * 3) if n == "xland" else rnd(m.nCells)
    for n, _ in HIST_3D:
        inp.hist[n] = rnd(nz + 1 if n in ("zgrid", "w") else nz, m.nCells)
    for n, _ in SOIL:
        inp.hist[n] = rnd(nsoil, m.nCells)
    for n, _ in DIAG:
        inp.diag[n] = rnd(nz, m.nCells) if n == "refl10cm" else rnd(m.nCells)
    cfg = I.InterpConfig(wrf_mod_vars=True, diag_list=DIAG, hist_2d=HIST_2D, hist_3d=HIST_3D, hist_soil=SOIL)
    mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_target(g)
    gi = I.GraphedInterp(mesh, grid, g, inp, cfg)
    for trial in range(2):
        for d in (inp.hist, inp.diag):
            for k, v in d.items():
                if k != "xland":
                    v.mul_(1.0 + 0.1 * (trial + 1)).add_(0.01)
        inp.hgt.add_(1.0)
        out = gi.replay()
        torch.cuda.synchronize()
        got = {k: v.clone() for k, v in out.items()}
        want = I.interp_data(mesh, grid, g, inp, cfg)
        assert set(got) == set(want)
        for k in want:
            assert torch.equal(got[k], want[k]), k
    gi.close()
    mesh.destroy()
    grid.destroy()
