"""mpg_regrid_store_begin / mpg_regrid_store_grid_begin (round 6): the independent RegridStores of a run (interp.F90:123, 207-437 stores
them one after the other) started on the library's worker thread and collected by the plain calls.  The bar: the SAME weights, bit for
bit, whatever runs beside the worker; cache / reference-count behaviour of the plain calls; nothing left behind."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _weights(rh):
    if rh.nnz_per_row == 0:        # conservative: CSR
        return rh.csr()
    return rh.weights()


def _same(a, b):
    return all(np.array_equal(x, y) for x, y in zip(a, b))


def test_begun_stores_give_the_plain_calls_weights(gpu_lib, regional_case):
    from mpassit_amd import regrid as R
    m, g = regional_case
    mesh_a, grid_a = R.Mesh.from_mpas(m), R.Grid.from_target(g)
    mesh_b, grid_b = R.Mesh.from_mpas(m), R.Grid.from_target(g)
    specs = [(R.REGRIDMETHOD_BILINEAR, R.MESHLOC_ELEMENT), (R.REGRIDMETHOD_CONSERVE, R.MESHLOC_ELEMENT), (R.REGRIDMETHOD_NEAREST_STOD, R.MESHLOC_ELEMENT),
             (R.REGRIDMETHOD_BILINEAR, R.MESHLOC_NODE)]
    for method, loc in specs:                       # all queued before any is collected
        R.regrid_store_begin(mesh_a, grid_a, method, meshloc=loc)
    for st in (R.STAGGERLOC_EDGE1, R.STAGGERLOC_EDGE2):
        R.regrid_store_grid_begin(grid_a, st)
    for method, loc in specs:
        ra, rb = R.regrid_store(mesh_a, grid_a, method, meshloc=loc), R.regrid_store(mesh_b, grid_b, method, meshloc=loc)
        assert _same(_weights(ra), _weights(rb)), (method, loc)
        assert ra.store_ms > 0
        ra.release()
        rb.release()
    for st in (R.STAGGERLOC_EDGE1, R.STAGGERLOC_EDGE2):
        ra, rb = R.regrid_store_grid(grid_a, st), R.regrid_store_grid(grid_b, st)
        assert _same(ra.weights(), rb.weights())
        ra.release()
        rb.release()
    for o in (mesh_a, mesh_b, grid_a, grid_b):
        o.destroy()


def test_cache_and_refcount_behave_as_for_plain_stores(gpu_lib, regional_case):
    from mpassit_amd import regrid as R
    m, g = regional_case
    mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_target(g)
    R.regrid_store_begin(mesh, grid, R.REGRIDMETHOD_NEAREST_STOD)
    R.regrid_store_begin(mesh, grid, R.REGRIDMETHOD_NEAREST_STOD)            # already queued / running / cached: nothing new
    a = R.regrid_store(mesh, grid, R.REGRIDMETHOD_NEAREST_STOD)
    b = R.regrid_store(mesh, grid, R.REGRIDMETHOD_NEAREST_STOD)
    assert a._h.value == b._h.value                                          # one weight set, two references
    t = a.store_ms
    a.release()
    b.release()
    R.regrid_store_begin(mesh, grid, R.REGRIDMETHOD_NEAREST_STOD)            # parked in the cache: no second Store
    c = R.regrid_store(mesh, grid, R.REGRIDMETHOD_NEAREST_STOD)
    assert c.store_ms == t
    c.release()
    # begun, never collected, and the mesh goes away while the worker may still be at it: destroy waits for the worker
    R.regrid_store_begin(mesh, grid, R.REGRIDMETHOD_CONSERVE)
    mesh.destroy()
    grid.destroy()


def test_refused_arguments_are_refused_at_once(gpu_lib, regional_case):
    from mpassit_amd import _lib as L, regrid as R
    m, g = regional_case
    mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_target(g)
    with pytest.raises(L.MpgError):
        R.regrid_store_begin(mesh, grid, R.REGRIDMETHOD_CONSERVE, staggerloc=R.STAGGERLOC_EDGE1)
    with pytest.raises(L.MpgError):
        R.regrid_store_begin(mesh, grid, R.REGRIDMETHOD_NEAREST_STOD, meshloc=R.MESHLOC_NODE)
    with pytest.raises(L.MpgError):
        R.regrid_store_grid_begin(grid, R.STAGGERLOC_CENTER)
    mesh.destroy()
    grid.destroy()


def test_regrids_beside_the_worker_and_interp_data_either_way(gpu_lib, global_mesh, conus_grid_30km):
    """The caller's thread regrids on its own stream while the worker builds the other weight sets; then the whole interp_data with
    and without the overlap: every output bit for bit."""
    import torch
    from mpassit_amd import interp as I, regrid as R
    m, g = global_mesh, conus_grid_30km
    nz = 8
    gen = torch.Generator(device="cuda")
    gen.manual_seed(9)
    src = torch.rand((nz, m.nCells), dtype=torch.float64, device="cuda", generator=gen)
    mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_target(g)
    rh = R.regrid_store(mesh, grid, R.REGRIDMETHOD_BILINEAR)
    want = rh.regrid(src.view(-1), nlev=nz).clone()
    torch.cuda.synchronize()
    R.regrid_store_begin(mesh, grid, R.REGRIDMETHOD_CONSERVE)
    R.regrid_store_begin(mesh, grid, R.REGRIDMETHOD_NEAREST_STOD)
    R.regrid_store_grid_begin(grid, R.STAGGERLOC_EDGE2)
    for _ in range(20):
        got = rh.regrid(src.view(-1), nlev=nz)
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    rh.release()
    mesh.destroy()
    grid.destroy()

    inp = I.InputData(nz=nz, nzp1=nz + 1, nsoil=4, hgt=torch.rand(m.nCells, dtype=torch.float64, device="cuda", generator=gen))
    hist2 = [("skintemp", "TSK"), ("snow", "SNOW"), ("xland", "XLAND")]
    hist3 = [("uReconstructZonal", "U"), ("uReconstructMeridional", "V"), ("theta", "T"), ("w", "W")]
    soil = [("smois", "SMOIS")]
    for n, _ in hist2:
        inp.hist[n] = torch.floor(torch.rand(m.nCells, dtype=torch.float64, device="cuda", generator=gen) * 3)
    for n, _ in hist3:
        inp.hist[n] = torch.rand((nz + 1 if n == "w" else nz, m.nCells), dtype=torch.float64, device="cuda", generator=gen)
    inp.hist["smois"] = torch.rand((4, m.nCells), dtype=torch.float64, device="cuda", generator=gen)
    outs = {}
    for overlap in (True, False):
        mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_target(g)          # fresh objects: every Store really runs
        cfg = I.InterpConfig(interp_diag=False, wrf_mod_vars=True, hist_2d=hist2, hist_3d=hist3, hist_soil=soil, overlap_stores=overlap)
        outs[overlap] = {k: v.clone() for k, v in I.interp_data(mesh, grid, g, inp, cfg).items()}
        torch.cuda.synchronize()
        mesh.destroy()
        grid.destroy()
    assert set(outs[True]) == set(outs[False]) and len(outs[True]) >= 8
    for k in outs[True]:
        assert torch.equal(outs[True][k], outs[False][k]), k
