"""The wind chain of interp_hist_data in one pass (mpg_wind_destagger_dev, csrc/k_wind.hip): interp.F90:291-293
rotate_winds_cgrid + :295-311 UMASS -> U(EDGE1) + :313-328 VMASS -> V(EDGE2).

The bar (round-5 review, item 1): BIT-identical to the three-call chain mpg_rotate_winds_dev -> mpg_regrid[_typed]_dev x 2 --
which the oracle tests of test_parity_gpu / test_interp_gpu / test_config5_gpu pin to the restatement -- on every shape the
chain serves: rotated or not, one component or both, float64 / float32 / big-endian results, grids narrower than a tile,
row lengths of every residue mod 16 (the store segments are shifted per row), a periodic global grid with pole caps."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LAMBERT = dict(ref_lat=38.5, ref_lon=-97.5, truelat1=38.5, truelat2=38.5, stand_lon=-97.5)


def _grid(nx, ny, dx=30000.0):
    from mpassit_amd import regrid as R, target_grid as T
    t = T.define_target_grid_params("lambert", nx + 1, ny + 1, dx=dx, dy=dx, **LAMBERT)
    return t, R.Grid.from_target(t)


def _winds(torch, t, nlev, seed):
    gen = torch.Generator(device="cuda")
    gen.manual_seed(seed)
    um = (torch.rand((nlev, t.ny, t.nx), dtype=torch.float64, device="cuda", generator=gen) - 0.5) * 60.0
    vm = (torch.rand((nlev, t.ny, t.nx), dtype=torch.float64, device="cuda", generator=gen) - 0.5) * 60.0
    return um, vm


def _chain(torch, R, t, grid, um, vm, nlev, rot, out_dtype, dst_be):
    """interp.F90:291-328 as three calls (the library's route until round 5)."""
    um, vm = um.clone(), vm.clone()
    if rot:
        cosa = torch.as_tensor(np.ascontiguousarray(t.cosa), device="cuda")
        sina = torch.as_tensor(np.ascontiguousarray(t.sina), device="cuda")
        R.rotate_winds_cgrid(cosa, sina, um, vm)
    res = []
    for st, m in ((R.STAGGERLOC_EDGE1, um), (R.STAGGERLOC_EDGE2, vm)):
        rh = R.regrid_store_grid(grid, st)
        if out_dtype == torch.float64 and not dst_be:
            res.append(rh.regrid(m.reshape(-1), nlev=nlev)[0])
        else:
            res.append(rh.regrid_typed(m.reshape(-1), nlev=nlev, out_dtype=out_dtype, dst_be=dst_be)[0])
        rh.release()
    return res[0], res[1], um, vm


def _same_bits(torch, a, b):
    it = {8: torch.int64, 4: torch.int32}[a.element_size()]
    return a.shape == b.shape and a.dtype == b.dtype and bool((a.contiguous().view(it) == b.contiguous().view(it)).all())


def _fused(torch, R, t, grid, um, vm, nlev, rot, out_dtype=None, dst_be=False, keep=False, which="uv"):
    rh_u = R.regrid_store_grid(grid, R.STAGGERLOC_EDGE1) if "u" in which else None
    rh_v = R.regrid_store_grid(grid, R.STAGGERLOC_EDGE2) if "v" in which else None
    cosa = sina = None
    if rot:
        cosa = torch.as_tensor(np.ascontiguousarray(t.cosa), device="cuda")
        sina = torch.as_tensor(np.ascontiguousarray(t.sina), device="cuda")
    um0, vm0 = um.clone(), vm.clone()
    res = R.wind_destagger(rh_u, rh_v, cosa, sina, um if ("u" in which or rot) else None, vm if ("v" in which or rot) else None, nlev,
                           out_dtype=out_dtype, dst_be=dst_be, keep_mass=keep)
    torch.cuda.synchronize()
    assert _same_bits(torch, um, um0) and _same_bits(torch, vm, vm0), "the mass winds are inputs"
    for rh in (rh_u, rh_v):
        if rh is not None:
            rh.release()
    return res


# row lengths of several residues mod 16 for U (nx + 1) and V (nx); grids narrower / lower than one 64 x 16 tile; one level
@pytest.mark.parametrize("nx,ny,nlev", [(180, 106, 5), (150, 90, 3), (64, 16, 2), (65, 17, 4), (33, 9, 1), (203, 35, 7), (17, 131, 2)])
def test_rotated_chain_bit_identical(gpu_lib, nx, ny, nlev):
    import torch
    from mpassit_amd import regrid as R
    t, grid = _grid(nx, ny)
    um, vm = _winds(torch, t, nlev, 100 + nx)
    u0, v0, ur0, vr0 = _chain(torch, R, t, grid, um, vm, nlev, True, torch.float64, False)
    u, v, ur, vr = _fused(torch, R, t, grid, um, vm, nlev, True, keep=True)
    assert _same_bits(torch, u, u0) and _same_bits(torch, v, v0)
    assert _same_bits(torch, ur, ur0) and _same_bits(torch, vr, vr0), "the rotated mass winds = rotate_winds_cgrid in place"
    u, v, ur, vr = _fused(torch, R, t, grid, um, vm, nlev, True)
    assert ur is None and vr is None and _same_bits(torch, u, u0) and _same_bits(torch, v, v0)
    # the outer half-cell ring lies outside the hull of the centres: 0.0
    assert float(u[:, :, 0].abs().max()) == 0.0 and float(u[:, :, -1].abs().max()) == 0.0
    assert float(v[:, 0, :].abs().max()) == 0.0 and float(v[:, -1, :].abs().max()) == 0.0
    grid.destroy()


@pytest.mark.parametrize("dtype,be", [("float32", False), ("float32", True), ("float64", True)])
def test_typed_results_bit_identical(gpu_lib, dtype, be):
    """What the Fortran driver's device flow asks for: U / V as the file stores them (NF90_FLOAT, big-endian)."""
    import torch
    from mpassit_amd import regrid as R
    t, grid = _grid(150, 90)
    um, vm = _winds(torch, t, 4, 7)
    dt = getattr(torch, dtype)
    for rot in (True, False):
        u0, v0, _, _ = _chain(torch, R, t, grid, um, vm, 4, rot, dt, be)
        u, v, _, _ = _fused(torch, R, t, grid, um, vm, 4, rot, out_dtype=dt, dst_be=be)
        assert _same_bits(torch, u, u0) and _same_bits(torch, v, v0)
    grid.destroy()


@pytest.mark.parametrize("which", ["uv", "u", "v"])
def test_unrotated_and_single_component(gpu_lib, which):
    """proj_code /= PROJ_LC, or only one of do_u_interp / do_v_interp (interp.F90:291,295,313)."""
    import torch
    from mpassit_amd import regrid as R
    t, grid = _grid(131, 77)
    um, vm = _winds(torch, t, 3, 11)
    u0, v0, _, _ = _chain(torch, R, t, grid, um, vm, 3, False, torch.float64, False)
    u, v, _, _ = _fused(torch, R, t, grid, um, vm, 3, False, which=which)
    assert (u is None) == ("u" not in which) and (v is None) == ("v" not in which)
    assert u is None or _same_bits(torch, u, u0)
    assert v is None or _same_bits(torch, v, v0)
    grid.destroy()


def test_periodic_grid_with_pole_caps(gpu_lib):
    """BASELINE configs[4]: global lat-lon target (ESMF_GridCreate1PeriDim + MONOPOLE, model_grid.F90:685-694): the window's
    columns wrap, the V rows at the poles take the row mean (k_pole_fix) -- the same bits as the Regrid of each handle."""
    import torch
    from mpassit_amd import regrid as R, target_grid as T
    for nxn, nyn in ((73, 37), (361, 181)):
        t = T.define_target_grid_params("lat-lon", nx=nxn, ny=nyn, stand_lon=0.0, is_regional=False)
        grid = R.Grid.from_target(t)
        um, vm = _winds(torch, t, 3, 5)
        u0, v0, _, _ = _chain(torch, R, t, grid, um, vm, 3, False, torch.float64, False)
        u, v, _, _ = _fused(torch, R, t, grid, um, vm, 3, False)
        assert _same_bits(torch, u, u0) and _same_bits(torch, v, v0)
        assert float(v[:, 0, :].abs().min()) > 0.0, "pole rows are mapped on a closed sphere"
        u32, v32, _, _ = _fused(torch, R, t, grid, um, vm, 3, False, out_dtype=torch.float32, dst_be=True)
        c32 = _chain(torch, R, t, grid, um, vm, 3, False, torch.float32, True)
        assert _same_bits(torch, u32, c32[0]) and _same_bits(torch, v32, c32[1])
        grid.destroy()


def test_non_finite_sources_stay_local(gpu_lib):
    """An Inf / NaN in a mass wind reaches exactly the points the chain lets it reach (unmapped points stay 0.0)."""
    import torch
    from mpassit_amd import regrid as R
    t, grid = _grid(100, 40)
    um, vm = _winds(torch, t, 2, 3)
    um[0, 0, 0] = float("inf")
    vm[1, 20, 50] = float("nan")
    u0, v0, _, _ = _chain(torch, R, t, grid, um, vm, 2, True, torch.float64, False)
    u, v, _, _ = _fused(torch, R, t, grid, um, vm, 2, True)
    assert _same_bits(torch, u, u0) and _same_bits(torch, v, v0)
    grid.destroy()


def test_refusals(gpu_lib):
    import torch
    from mpassit_amd import _lib as L, regrid as R
    t, grid = _grid(80, 40)
    t2, grid2 = _grid(81, 40)
    um, vm = _winds(torch, t, 2, 1)
    rh_u, rh_v = R.regrid_store_grid(grid, R.STAGGERLOC_EDGE1), R.regrid_store_grid(grid, R.STAGGERLOC_EDGE2)
    rh_v2 = R.regrid_store_grid(grid2, R.STAGGERLOC_EDGE2)
    cosa = torch.as_tensor(np.ascontiguousarray(t.cosa), device="cuda")
    sina = torch.as_tensor(np.ascontiguousarray(t.sina), device="cuda")
    with pytest.raises(L.MpgError) as e:       # handles of two different grids
        R.wind_destagger(rh_u, rh_v2, None, None, um, vm, 2)
    assert e.value.rc == L.MPG_ERR_UNSUPPORTED
    with pytest.raises(L.MpgError) as e:       # EDGE2 handle in the EDGE1 slot
        R.wind_destagger(rh_v, rh_u, None, None, um, vm, 2)
    assert e.value.rc == L.MPG_ERR_UNSUPPORTED
    with pytest.raises(L.MpgError) as e:       # a rotation needs both components
        R.wind_destagger(rh_u, None, cosa, sina, um, vm, 2)
    assert e.value.rc == L.MPG_ERR_INVALID_ARG
    lib = L.load()
    p = lambda x: C.c_void_p(x.data_ptr())   # noqa: E731
    u = torch.empty((2, t.ny, t.nx + 1), dtype=torch.float64, device="cuda")
    v = torch.empty((2, t.ny + 1, t.nx), dtype=torch.float64, device="cuda")
    # the rotated mass winds cannot overwrite the inputs; cos without sin; no handle at all; NULL destination
    assert lib.mpg_wind_destagger_dev(rh_u._h, rh_v._h, p(cosa), p(sina), p(um), p(vm), 2, p(u), p(v), 0, p(um), None, None) == L.MPG_ERR_INVALID_ARG
    assert lib.mpg_wind_destagger_dev(rh_u._h, rh_v._h, p(cosa), None, p(um), p(vm), 2, p(u), p(v), 0, None, None, None) == L.MPG_ERR_INVALID_ARG
    assert lib.mpg_wind_destagger_dev(None, None, None, None, p(um), p(vm), 2, p(u), p(v), 0, None, None, None) == L.MPG_ERR_INVALID_ARG
    assert lib.mpg_wind_destagger_dev(rh_u._h, rh_v._h, None, None, p(um), p(vm), 2, None, p(v), 0, None, None, None) == L.MPG_ERR_INVALID_ARG
    assert lib.mpg_wind_destagger_dev(rh_u._h, rh_v._h, None, None, p(um), p(vm), 0, p(u), p(v), 0, None, None, None) == L.MPG_ERR_INVALID_ARG
    for rh in (rh_u, rh_v, rh_v2):
        rh.release()
    grid.destroy()
    grid2.destroy()


def test_interp_data_fused_equals_three_calls(gpu_lib, global_mesh, conus_grid_30km):
    """interp_data with wrf_mod_vars on device-resident fields: the fused route and the three-call route give the same U, V
    (and, when asked for, the same UMASS / VMASS), bit for bit."""
    import torch
    from mpassit_amd import interp as I, regrid as R
    m, g = global_mesh, conus_grid_30km
    mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_target(g)
    nz = 6
    gen = torch.Generator(device="cuda")
    gen.manual_seed(5)
    inp = I.InputData(nz=nz, nzp1=nz + 1, nsoil=0, hgt=torch.rand(m.nCells, dtype=torch.float64, device="cuda", generator=gen))
    for n in ("uReconstructZonal", "uReconstructMeridional", "theta"):
        inp.hist[n] = torch.rand((nz, m.nCells), dtype=torch.float64, device="cuda", generator=gen) * 20 - 10
    hist3 = [("uReconstructZonal", "U"), ("uReconstructMeridional", "V"), ("theta", "T")]
    outs = {}
    for fused in (True, False):
        cfg = I.InterpConfig(interp_diag=False, wrf_mod_vars=True, hist_3d=hist3, keep_mass_winds=True, fused_winds=fused)
        outs[fused] = I.interp_data(mesh, grid, g, inp, cfg)
    torch.cuda.synchronize()
    assert set(outs[True]) == set(outs[False]) and {"U", "V", "UMASS", "VMASS", "T"} <= set(outs[True])
    for k in outs[True]:
        assert _same_bits(torch, outs[True][k], outs[False][k]), k
    cfg = I.InterpConfig(interp_diag=False, wrf_mod_vars=True, hist_3d=hist3)
    out = I.interp_data(mesh, grid, g, inp, cfg)
    assert "UMASS" not in out and "VMASS" not in out, "u/v_target_grid_nostag are intermediates (write_data.F90 never writes them)"
    assert _same_bits(torch, out["U"], outs[False]["U"]) and _same_bits(torch, out["V"], outs[False]["V"])
    mesh.destroy()
    grid.destroy()


# ---- host arrays: mpg_wind_destagger (the reference's own shape: farrayPtr in, farrayPtr out) ---------------------------------
@pytest.mark.parametrize("nx,ny,nlev,rot,dtype,be", [(180, 106, 5, True, "float64", False), (150, 90, 3, False, "float64", False), (203, 35, 7, True, "float32", True),
                                                     (65, 17, 1, True, "float32", False), (1800, 1060, 9, True, "float64", False)])
def test_host_arrays_same_bits_as_the_device_form(gpu_lib, nx, ny, nlev, rot, dtype, be):
    """The host-array chain (mass winds up once in chunks of levels, only U and V down) gives the bits of mpg_wind_destagger_dev and so of the
    three-call chain; the 1800 x 1060 case takes several chunks (three levels of two fields per 96 MB), the last one short."""
    import torch
    from mpassit_amd import regrid as R
    t, grid = _grid(nx, ny)
    um, vm = _winds(torch, t, nlev, 300 + nx)
    dt = getattr(torch, dtype)
    u0, v0, ur0, vr0 = _fused(torch, R, t, grid, um, vm, nlev, rot, out_dtype=dt, dst_be=be, keep=rot)
    rh_u, rh_v = R.regrid_store_grid(grid, R.STAGGERLOC_EDGE1), R.regrid_store_grid(grid, R.STAGGERLOC_EDGE2)
    umh, vmh = um.cpu().numpy(), vm.cpu().numpy()
    um_in, vm_in = umh.copy(), vmh.copy()
    u, v, ur, vr = R.wind_destagger(rh_u, rh_v, t.cosa if rot else None, t.sina if rot else None, umh, vmh, nlev, out_dtype=getattr(np, dtype), dst_be=be,
                                    keep_mass=rot)
    assert np.array_equal(umh, um_in) and np.array_equal(vmh, vm_in), "the mass winds are inputs"
    it = {"float64": np.int64, "float32": np.int32}[dtype]
    assert u.shape == tuple(u0.shape) and np.array_equal(u.view(it), u0.cpu().numpy().view(it))
    assert v.shape == tuple(v0.shape) and np.array_equal(v.view(it), v0.cpu().numpy().view(it))
    if rot:
        assert np.array_equal(ur.view(np.int64), ur0.cpu().numpy().view(np.int64)) and np.array_equal(vr.view(np.int64), vr0.cpu().numpy().view(np.int64))
    else:
        assert ur is None and vr is None
    # one component only (do_u_interp or do_v_interp alone, interp.F90:295,313)
    if not rot:
        u1, v1, _, _ = R.wind_destagger(rh_u, None, None, None, umh, None, nlev, out_dtype=getattr(np, dtype), dst_be=be)
        assert v1 is None and np.array_equal(u1.view(it), u.view(it))
        u2, v2, _, _ = R.wind_destagger(None, rh_v, None, None, None, vmh, nlev, out_dtype=getattr(np, dtype), dst_be=be)
        assert u2 is None and np.array_equal(v2.view(it), v.view(it))
    rh_u.release()
    rh_v.release()
    grid.destroy()


def test_host_arrays_rotated_in_place_like_the_reference(gpu_lib):
    """rotate_winds_cgrid works in place (interp.F90:702-748): with host arrays the rotated mass winds may come back into the arrays that
    held the earth-relative ones (a level's result comes down after that level went up); arguments are checked as in the device form."""
    import ctypes as C
    import torch
    from mpassit_amd import _lib as L, regrid as R
    t, grid = _grid(150, 90)
    um, vm = _winds(torch, t, 6, 11)
    u0, v0, ur0, vr0 = _fused(torch, R, t, grid, um, vm, 6, True, keep=True)
    rh_u, rh_v = R.regrid_store_grid(grid, R.STAGGERLOC_EDGE1), R.regrid_store_grid(grid, R.STAGGERLOC_EDGE2)
    umh, vmh = um.cpu().numpy().copy(), vm.cpu().numpy().copy()
    u, v = np.empty(tuple(u0.shape)), np.empty(tuple(v0.shape))
    cosa, sina = np.ascontiguousarray(t.cosa, dtype=np.float64), np.ascontiguousarray(t.sina, dtype=np.float64)
    lib = L.load()

    def p(a):
        return a.ctypes.data_as(C.c_void_p) if a is not None else None
    assert lib.mpg_wind_destagger(rh_u._h, rh_v._h, p(cosa), p(sina), p(umh), p(vmh), 6, p(u), p(v), 0, p(umh), p(vmh)) == 0, lib.mpg_last_error()
    assert np.array_equal(u, u0.cpu().numpy()) and np.array_equal(v, v0.cpu().numpy())
    assert np.array_equal(umh, ur0.cpu().numpy()) and np.array_equal(vmh, vr0.cpu().numpy())
    bad = L.MPG_ERR_INVALID_ARG
    assert lib.mpg_wind_destagger(None, None, None, None, p(umh), p(vmh), 6, p(u), p(v), 0, None, None) == bad
    assert lib.mpg_wind_destagger(rh_u._h, rh_v._h, p(cosa), None, p(umh), p(vmh), 6, p(u), p(v), 0, None, None) == bad
    assert lib.mpg_wind_destagger(rh_u._h, None, p(cosa), p(sina), p(umh), p(vmh), 6, p(u), None, 0, None, None) == bad       # the rotation needs both
    assert lib.mpg_wind_destagger(rh_u._h, rh_v._h, None, None, p(umh), p(vmh), 6, None, p(v), 0, None, None) == bad
    assert lib.mpg_wind_destagger(rh_u._h, rh_v._h, None, None, p(umh), p(vmh), 0, p(u), p(v), 0, None, None) == bad
    assert lib.mpg_wind_destagger(rh_u._h, rh_v._h, None, None, p(umh), p(vmh), 6, p(u), p(v), 0, p(umh), None) == bad       # rotated winds without a rotation
    assert lib.mpg_wind_destagger(rh_v._h, rh_u._h, None, None, p(umh), p(vmh), 6, p(u), p(v), 0, None, None) == L.MPG_ERR_UNSUPPORTED
    rh_u.release()
    rh_v.release()
    grid.destroy()
