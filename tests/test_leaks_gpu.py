"""Object lifetimes: creating and destroying meshes, grids and route handles (every Store family, both layouts so that the
tile lists of two shapes exist, typed and graph-free paths) over and over must give all device memory back."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_create_use_destroy_cycles_do_not_leak_device_memory(gpu_lib, regional_case):
    import torch

    from mpassit_amd import regrid as R
    m, g = regional_case
    nz = 4
    rng = np.random.default_rng(2)
    src = rng.standard_normal((nz, m.nCells))
    src_t = np.ascontiguousarray(src.T).astype(np.float32)

    def cycle():
        mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_target(g)
        hs = [R.regrid_store(mesh, grid, meth) for meth in (R.REGRIDMETHOD_BILINEAR, R.REGRIDMETHOD_NEAREST_STOD, R.REGRIDMETHOD_CONSERVE)]
        hs.append(R.regrid_store(mesh, grid, R.REGRIDMETHOD_BILINEAR, meshloc=R.MESHLOC_NODE))
        hs.append(R.regrid_store_grid(grid, R.STAGGERLOC_EDGE1))
        out = hs[0].regrid(src, nlev=nz)                                     # cell-fast: staged tile lists
        hs[0].regrid(src_t, nlev=nz, layout=R.LAYOUT_LEV_FAST)               # level-fast float32: second list shape parked
        hs[0].regrid(src[0], nlev=1)
        hs[1].regrid(src[0], nlev=1)
        hs[2].regrid(src[0], nlev=1)
        hs[4].regrid(out.reshape(-1), nlev=nz)
        ids = hs[0].localize()                                               # re-indexing drops and rebuilds the lists
        hs[0].regrid(np.ascontiguousarray(src[:, ids]), nlev=nz)
        for h in hs:
            h.release()
        mesh.destroy()
        grid.destroy()

    for _ in range(3):
        cycle()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(25):
        cycle()
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < (8 << 20), "device memory shrank by %.1f MB over 25 cycles" % ((free0 - free1) / 2 ** 20)
