#!/usr/bin/env python3
"""The C-ABI's RCCL exchange at the PLANNED sizes, rehearsed on one GPU: configuration 4 split over V virtual ranks
(mpg_comm_virtual: one host thread per rank, the real ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd to self), 13 fields x 55
levels per exchange -- mpg_halo_build, mpg_halo_exchange_dev, the Regrid of every rank's slab, mpg_gather_rows of one field.
Checks every rank's row block and the gathered field against the single-GPU Regrid (bit for bit) and prints what the exchange and
the gather took with all V ranks' traffic on the one card (to self: device copies -- no xGMI; NO scaling curve).
usage (GPU box): python tools/vranks_c4.py [--ranks 8] [--fields 13] [--io f64|f32] [--layout cell_fast|lev_fast]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--fields", type=int, default=13)
    ap.add_argument("--io", default="f64", choices=["f64", "f32"])
    ap.add_argument("--layout", default="cell_fast", choices=["cell_fast", "lev_fast"])
    ap.add_argument("--workload", default="c4_3m_regional")
    ap.add_argument("--ownership", default="aligned", choices=["aligned", "para_range", "need"],
                    help="need: the owned halo form with every cell given to the lowest rank whose rows reference it (meshes without banded numbering)")
    args = ap.parse_args()
    import torch
    from mpassit_amd import _lib, comm as MC, dist as mdist, regrid as R, workloads
    os.environ.setdefault("MPG_COMM_TIMEOUT_S", "60")
    _lib.init(0)
    m, g, nlev, desc = workloads.workload(args.workload)
    V, F = args.ranks, args.fields
    lev_fast = args.layout == "lev_fast"
    layout = R.LAYOUT_LEV_FAST if lev_fast else R.LAYOUT_CELL_FAST
    dt = torch.float32 if args.io == "f32" else torch.float64
    gen = torch.Generator(device="cuda")
    gen.manual_seed(7)
    src = torch.rand((F, m.nCells, nlev) if lev_fast else (F * nlev, m.nCells), dtype=dt, device="cuda", generator=gen)
    mesh0, grid0 = R.Mesh.from_mpas(m), R.Grid.from_target(g)
    rh0 = R.regrid_store(mesh0, grid0, R.REGRIDMETHOD_BILINEAR)
    want = rh0.regrid_typed(src.view(-1), nlev=nlev, nfields=F, layout=layout)
    rh0.release()
    mesh0.destroy()
    grid0.destroy()
    real = MC.Comm(0, 1)
    vcs = [real.virtual(r, V) for r in range(V)]
    blocks = [mdist.row_block(g.ny, V, r) for r in range(V)]
    grids = [R.Grid.from_target(g, rows=b) for b in blocks]
    meshes = [R.Mesh.from_mpas(m, window_grid=grids[r]) for r in range(V)]
    rhs = [R.regrid_store(meshes[r], grids[r], R.REGRIDMETHOD_BILINEAR) for r in range(V)]
    owned = [None] * V
    needed_total = 0
    if args.ownership == "need":
        seen = np.zeros(m.nCells, bool)
        for r in range(V):
            nd = rhs[r].unique_sources()
            needed_total += nd.size
            owned[r] = nd[~seen[nd]].astype(np.int32)
            seen[nd] = True
    torch.cuda.synchronize()
    times = {}

    def body(r):
        torch.cuda.set_device(0)
        st = torch.cuda.Stream()
        t0 = time.perf_counter()
        h = MC.Halo(vcs[r], rhs[r], m.nCells, ownership=args.ownership if owned[r] is None else "aligned", owned_ids=owned[r])
        t_build = (time.perf_counter() - t0) * 1e3
        with torch.cuda.stream(st):
            local = torch.empty((F, h.n_local, nlev) if lev_fast else (F * nlev, h.n_local), dtype=dt, device="cuda")
            if h.mode == "range":
                own = local[:, h.own_pos[0]:h.own_pos[1]]
                own.copy_(src[:, h.own[0]:h.own[1]])
            elif h.mode == "owned":
                own = src[:, torch.as_tensor(owned[r].astype(np.int64), device="cuda")].contiguous()
            else:
                own = src[:, h.own[0]:h.own[1]].contiguous()
            st.synchronize()
            ex = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st)
                h.exchange(own, local, stream=st.cuda_stream)
                e1.record(st)
                st.synchronize()
                ex.append(e0.elapsed_time(e1))
            out = rhs[r].regrid_typed(local.view(-1), nlev=nlev, nfields=F, layout=layout)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            whole = vcs[r].gather_rows(out[0], blocks[r][0], blocks[r][1], g.ny, root=0, stream=st.cuda_stream)
            e1.record(st)
        st.synchronize()
        ok = bool(torch.equal(out, want[:, :, blocks[r][0]:blocks[r][1]]))
        if r == 0:
            ok = ok and bool(torch.equal(whole, want[0]))
        times[r] = dict(mode=h.mode, halo_build_ms=round(t_build, 2), exchange_ms=round(min(ex), 4), gather_ms_incl_waiting_for_the_other_threads=round(e0.elapsed_time(e1), 4), n_local=h.n_local,
                        received_per_row=h.received_per_row, ok=ok)
        h.destroy()
        return ok

    oks = MC.run_virtual_ranks(V, body, timeout=300.0)
    groups, sends, recvs, allgathers = real.virtual_stats()
    esz = (4 if args.io == "f32" else 8)
    rows = F * nlev
    res = {"workload": desc, "virtual_ranks": V, "fields": F, "io": args.io, "layout": args.layout, "ownership": args.ownership, "all_row_blocks_and_gathered_field_bit_identical": all(oks),
           "rccl_calls": {"groups": groups, "sends": sends, "recvs": recvs, "allgathers": allgathers},
           "halo_bytes_per_exchange_all_ranks": int(sum(t["received_per_row"] for t in times.values()) * rows * esz),
           "exchange_ms_all_ranks_traffic_on_one_card": max(t["exchange_ms"] for t in times.values()),
           "per_rank": [times[r] for r in range(V)],
           "note": "to self on one GPU: device copies inside RCCL, the threads' rendezvous inside the timings; NO scaling curve"}
    print(json.dumps(res))
    for rh, me, gr, vc in zip(rhs, meshes, grids, vcs):
        rh.release()
        me.destroy()
        gr.destroy()
        vc.destroy()
    real.destroy()
    _lib.finalize()
    return 0 if all(oks) else 1


if __name__ == "__main__":
    sys.exit(main())
