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


def test_comm_and_halo_cycles_do_not_leak_device_memory(gpu_lib):
    """The multi-rank objects: a communicator, its virtual ranks, halos of the three forms (range, compact, owned), an exchange
    and a gather through each, row-block grids and windowed meshes -- created and destroyed over and over."""
    import torch

    from mpassit_amd import comm as MC, dist as mdist, regrid as R, synth, target_grid as tg
    g = tg.define_target_grid_params("lambert", 101, 67, dx=30000.0, dy=30000.0, ref_lat=38.5, ref_lon=-97.5, truelat1=38.5, truelat2=38.5,
                                     stand_lon=-97.5)
    m = synth.regional_mesh_for_lambert(g.proj, 95, 60, 24000, margin=0.0)
    V, nlev = 3, 4
    src = torch.rand((nlev, m.nCells), dtype=torch.float64, device="cuda")
    streams = [torch.cuda.Stream() for _ in range(V)]     # made once: every new HIP stream costs device memory of its own (~12 MB, measured with
                                                          # tools/leak_diag.py: torch's pool of 32 streams fills and the growth stops), not this library's

    def cycle(form):
        real = MC.Comm(0, 1)
        vcs = [real.virtual(r, V) for r in range(V)]
        blocks = [mdist.row_block(g.ny, V, r) for r in range(V)]
        grids = [R.Grid.from_target(g, rows=b) for b in blocks]
        meshes = [R.Mesh.from_mpas(m, window_grid=grids[r]) for r in range(V)]
        rhs = [R.regrid_store(meshes[r], grids[r], R.REGRIDMETHOD_BILINEAR) for r in range(V)]
        owned = [None] * V
        if form == "owned":
            seen = np.zeros(m.nCells, bool)
            for r in range(V):
                nd = rhs[r].unique_sources()
                owned[r] = nd[~seen[nd]].astype(np.int32)
                seen[nd] = True
        torch.cuda.synchronize()

        def body(r):
            torch.cuda.set_device(0)
            st = streams[r]
            h = MC.Halo(vcs[r], rhs[r], m.nCells, ownership="aligned" if form != "para_range" else "para_range", owned_ids=owned[r])
            with torch.cuda.stream(st):
                local = torch.zeros((nlev, h.n_local), dtype=torch.float64, device="cuda")
                if h.mode == "owned":
                    own = src[:, torch.as_tensor(owned[r].astype(np.int64), device="cuda")].contiguous()
                else:
                    own = src[:, h.own[0]:h.own[1]].contiguous()
                st.synchronize()
                h.exchange(own, local, stream=st.cuda_stream)
                out = rhs[r].regrid_typed(local.view(-1), nlev=nlev, nfields=1)
                vcs[r].gather_rows(out[0], blocks[r][0], blocks[r][1], g.ny, root=0, stream=st.cuda_stream)
            st.synchronize()
            h.destroy()
            return True

        assert all(MC.run_virtual_ranks(V, body, timeout=120.0))
        for rh, me, gr, vc in zip(rhs, meshes, grids, vcs):
            rh.release()
            me.destroy()
            gr.destroy()
            vc.destroy()
        real.destroy()

    forms = ("aligned", "para_range", "owned")
    for f in forms:
        cycle(f)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()                     # torch parks blocks per stream, and every thread here takes a stream of its own
    free0 = torch.cuda.mem_get_info()[0]
    for i in range(12):
        cycle(forms[i % 3])
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < (8 << 20), "device memory shrank by %.1f MB over 12 cycles" % ((free0 - free1) / 2 ** 20)
