#!/usr/bin/env python3
"""Which object cycle costs device memory: communicators, virtual ranks, row-block grids / windowed meshes / Stores, halos of the three
forms with and without an exchange -- each created and destroyed 12 times, free device memory read before and after (torch's cache
emptied).  Finding of round 5 (the box's output is quoted in tests/test_leaks_gpu.py): nothing of the library's grows; the first loop that
takes a NEW torch stream per thread grows by ~12 MB per stream until torch's pool of 32 streams exists, then stops.
usage (GPU box): python tools/leak_diag.py"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from mpassit_amd import _lib, comm as MC, dist as mdist, regrid as R, synth, target_grid as tg
_lib.init(0)
def free():
    torch.cuda.synchronize(); torch.cuda.empty_cache(); return torch.cuda.mem_get_info()[0]
def rep(name, fn, n=12):
    fn(); fn()
    f0 = free()
    for _ in range(n): fn()
    f1 = free()
    print("%-40s %8.2f MB over %d" % (name, (f0 - f1) / 2**20, n), flush=True)
def c_real():
    c = MC.Comm(0, 1); c.destroy()
rep("Comm(0,1) create/destroy", c_real)
real = MC.Comm(0, 1)
def c_virt():
    v = [real.virtual(r, 3) for r in range(3)]
    for x in v: x.destroy()
rep("3 virtual comms", c_virt)
g = tg.define_target_grid_params("lambert", 101, 67, dx=30000.0, dy=30000.0, ref_lat=38.5, ref_lon=-97.5, truelat1=38.5, truelat2=38.5, stand_lon=-97.5)
m = synth.regional_mesh_for_lambert(g.proj, 95, 60, 24000, margin=0.0)
V = 3
blocks = [mdist.row_block(g.ny, V, r) for r in range(V)]
def c_objs():
    grids = [R.Grid.from_target(g, rows=b) for b in blocks]
    meshes = [R.Mesh.from_mpas(m, window_grid=grids[r]) for r in range(V)]
    rhs = [R.regrid_store(meshes[r], grids[r], R.REGRIDMETHOD_BILINEAR) for r in range(V)]
    for rh, me, gr in zip(rhs, meshes, grids):
        rh.release(); me.destroy(); gr.destroy()
rep("row-block grids/windowed meshes/stores", c_objs)
grids = [R.Grid.from_target(g, rows=b) for b in blocks]
meshes = [R.Mesh.from_mpas(m, window_grid=grids[r]) for r in range(V)]
rhs = [R.regrid_store(meshes[r], grids[r], R.REGRIDMETHOD_BILINEAR) for r in range(V)]
src = torch.rand((4, m.nCells), dtype=torch.float64, device="cuda")
owned_need = [None] * V
seen = np.zeros(m.nCells, bool)
for r in range(V):
    nd = rhs[r].unique_sources(); owned_need[r] = nd[~seen[nd]].astype(np.int32); seen[nd] = True
for x in rhs: x.release()
for form in ("aligned", "para_range", "owned"):
    owned = owned_need if form == "owned" else [None] * V
    for do_ex in (False, True):
        def c_halo():
            vcs = [real.virtual(r, V) for r in range(V)]
            rhs = [R.regrid_store(meshes[r], grids[r], R.REGRIDMETHOD_BILINEAR) for r in range(V)]
            def body(r):
                torch.cuda.set_device(0)
                h = MC.Halo(vcs[r], rhs[r], m.nCells, ownership="aligned" if form != "para_range" else "para_range", owned_ids=owned[r])
                if do_ex:
                    st = torch.cuda.Stream()
                    with torch.cuda.stream(st):
                        local = torch.zeros((4, h.n_local), dtype=torch.float64, device="cuda")
                        own = src[:, torch.as_tensor(owned[r].astype(np.int64), device="cuda")].contiguous() if h.mode == "owned" else src[:, h.own[0]:h.own[1]].contiguous()
                        st.synchronize()
                        h.exchange(own, local, stream=st.cuda_stream)
                        out = rhs[r].regrid_typed(local.view(-1), nlev=4, nfields=1)
                        vcs[r].gather_rows(out[0], blocks[r][0], blocks[r][1], g.ny, root=0, stream=st.cuda_stream)
                    st.synchronize()
                h.destroy()
                return True
            MC.run_virtual_ranks(V, body, timeout=120.0)
            for x in vcs: x.destroy()
            for x in rhs: x.release()
        rep("halo %s exchange=%s" % (form, do_ex), c_halo)
