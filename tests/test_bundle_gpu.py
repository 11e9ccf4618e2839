"""ESMF_FieldBundleRegrid over separate field arrays (interp.F90:240-254 -> mpg_regrid_bundle_typed_dev): one launch for all
fields of a bundle must give, bit for bit, what one call per field gives -- for every kernel family the typed Regrid dispatches
to (staged and gathering 3-point kernels in both layouts, nearest, the 4-point destagger with its pole caps, CSR), float32 and
float64 on either side, either byte order, per-field epilogue offsets, more fields than one table chunk holds."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _check(rh, R, torch, nlev, nf, layout, sdt, ddt, be=False, offsets=True, scale=1.0):
    gen = torch.Generator(device="cuda")
    gen.manual_seed(nlev * 1000 + nf)
    shape = (rh.n_src, nlev) if layout == R.LAYOUT_LEV_FAST else (nlev, rh.n_src)
    srcs = [(torch.rand(shape, dtype=torch.float64, device="cuda", generator=gen) * 40 + 250).to(sdt) for _ in range(nf)]
    if be:   # any bit pattern is a valid big-endian number for this purpose; avoid NaN patterns by swapping real values
        srcs = [t.view(torch.uint8).view(-1, t.element_size()).flip(1).contiguous().view(t.dtype).view(shape) for t in srcs]
    offs = [(-300.0 if f % 3 == 1 else 0.0) for f in range(nf)] if offsets else None
    got = rh.regrid_bundle(srcs, nlev=nlev, layout=layout, out_dtype=ddt, scale=scale, offsets=offs, src_be=be, dst_be=be)
    for f in range(nf):
        want = rh.regrid_typed(srcs[f].view(-1), nlev=nlev, nfields=1, layout=layout, out_dtype=ddt, scale=scale,
                               offset=offs[f] if offs else 0.0, src_be=be, dst_be=be)[0]
        it = torch.int32 if ddt == torch.float32 else torch.int64
        assert torch.equal(got[f].view(it), want.view(it)), (f, nlev, nf, layout, sdt, ddt, be)


def test_bundle_of_separate_fields_equals_single_calls(gpu_lib, regional_case):
    import torch
    from mpassit_amd import regrid as R
    m, g = regional_case
    mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_target(g)
    rb = R.regrid_store(mesh, grid, R.REGRIDMETHOD_BILINEAR)
    rn = R.regrid_store(mesh, grid, R.REGRIDMETHOD_NEAREST_STOD)
    rc = R.regrid_store(mesh, grid, R.REGRIDMETHOD_CONSERVE)
    f32, f64 = torch.float32, torch.float64
    try:
        for layout in (R.LAYOUT_LEV_FAST, R.LAYOUT_CELL_FAST):
            for variant in ((-1,) if layout == R.LAYOUT_CELL_FAST else (0, 1, 2)):
                gpu_lib.tune("lf_variant", variant)
                for sdt, ddt in ((f32, f32), (f32, f64), (f64, f32), (f64, f64)):
                    _check(rb, R, torch, 17, 3, layout, sdt, ddt)
                _check(rb, R, torch, 55, 35, layout, f32, f32, be=True)          # more than one table chunk, big-endian
            gpu_lib.tune("lf_variant", -1)
            for staged in (-2, 0, 2):
                gpu_lib.tune("a3_staged", staged)
                _check(rb, R, torch, 9, 4, R.LAYOUT_CELL_FAST, f64, f64, scale=9.81)
            gpu_lib.tune("a3_staged", -1)
            _check(rn, R, torch, 4, 5, layout, f32, f32)
            _check(rc, R, torch, 1, 2, layout, f64, f32)
            _check(rb, R, torch, 1, 6, R.LAYOUT_CELL_FAST, f32, f32, offsets=False)   # 2-D fields
    finally:
        gpu_lib.tune("lf_variant", -1)
        gpu_lib.tune("a3_staged", -1)
    for rh in (rb, rn, rc):
        rh.release()
    mesh.destroy()
    grid.destroy()


def test_bundle_through_the_destagger_with_pole_caps(gpu_lib):
    """Grid -> Grid CENTER -> EDGE2 on a global lat-lon grid (4-point kernel + the pole-cap rewrite): bundle == single calls."""
    import torch
    from mpassit_amd import regrid as R, workloads
    _, gg, _, _ = workloads.workload("c5_small")
    grid = R.Grid.from_proj(gg, fill_target=False)
    rh = R.regrid_store_grid(grid, R.STAGGERLOC_EDGE2)
    for layout in (R.LAYOUT_CELL_FAST, R.LAYOUT_LEV_FAST):
        _check(rh, R, torch, 5, 3, layout, torch.float64, torch.float32)
    rh.release()
    grid.destroy()


def test_bundle_refuses_bad_arguments(gpu_lib, regional_case):
    import ctypes as C
    import torch
    from mpassit_amd import regrid as R
    m, g = regional_case
    mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_target(g)
    rh = R.regrid_store(mesh, grid, R.REGRIDMETHOD_BILINEAR)
    lib = gpu_lib.load()
    src = torch.zeros(rh.n_src, device="cuda")
    dst = torch.zeros(rh.n_dst, device="cuda")
    sp = (C.c_void_p * 2)(src.data_ptr(), None)
    dp = (C.c_void_p * 2)(dst.data_ptr(), dst.data_ptr())
    assert lib.mpg_regrid_bundle_typed_dev(rh._h, C.c_int(2), sp, C.c_int(1), C.c_int(0), C.c_int(1), dp, C.c_int(1), C.c_double(1.0), None, None) != 0
    assert lib.mpg_regrid_bundle_typed_dev(rh._h, C.c_int(0), sp, C.c_int(1), C.c_int(0), C.c_int(1), dp, C.c_int(1), C.c_double(1.0), None, None) != 0
    rh.release()
    mesh.destroy()
    grid.destroy()
