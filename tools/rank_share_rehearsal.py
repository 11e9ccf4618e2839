#!/usr/bin/env python3
"""ONE rank's share of the N-GPU headline step, rehearsed on one GPU: what does a step cost a rank when its Regrid kernel has
shrunk to 1/N of the rows -- kernel, exchange, and the HOST's time to issue it all?

bench.py --gpus N runs, per step and rank: the halo exchange of batch s+1 on a side stream (two strided copies into the packed
send buffer, one all_to_all_single, two strided copies out), event edges between the two streams, and the Regrid of batch s.  At
N = 1 the 5 ms kernel hides every microsecond of that.  At N = 8 a rank's kernel takes ~0.65 ms, and nobody had timed the rest.
This tool builds rank R's objects of an N-rank split of configuration 4 (its row block, its windowed mesh, its handle, its halo
plan out of all N ranks' needed ids -- mpg_halo_plan_host), then runs bench.py's very pipeline with the rank's two neighbour strips
exchanged TO SELF through torch.distributed's all_to_all_single on a one-rank RCCL group (the strips' real sizes; RCCL copies
device memory to device memory instead of crossing xGMI), and reports per step:

  host_issue_us   wall time of the Python + C calls of one step, issued into an idle device (no back-pressure)
  kernel_ms       the rank's Regrid of 13 x 55 levels (HIP events)
  exchange_ms     the exchange as the halo stream saw it, self-loopback (a device copy: a LOWER bound for the real thing)
  xgmi_est_ms     the same bytes over one xGMI link per neighbour at 153 GB/s x 0.8 + 20 us (ASSUMED, not measured)
  step_ms         wall per step of the pipelined loop
  implied_rate    13 fields / max(kernel, xgmi_est, host issue) -- what 8 such ranks would deliver IF they behave like this one

usage (GPU box): python tools/rank_share_rehearsal.py [--ranks 8] [--rank 3] [--fields 13] [--io f64|f32] [--layout cell_fast|lev_fast]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--rank", type=int, default=3)
    ap.add_argument("--fields", type=int, default=13)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--io", default="f64", choices=["f64", "f32"])
    ap.add_argument("--layout", default="cell_fast", choices=["cell_fast", "lev_fast"])
    ap.add_argument("--workload", default="c4_3m_regional")
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    from mpassit_amd import _lib, comm as MC, dist as mdist, regrid as R, workloads
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    _lib.init(0)
    m, g, nlev, desc = workloads.workload(args.workload)
    N, r, F = args.ranks, args.rank, args.fields
    lev_fast = args.layout == "lev_fast"
    layout = R.LAYOUT_LEV_FAST if lev_fast else R.LAYOUT_CELL_FAST
    dt = torch.float32 if args.io == "f32" else torch.float64
    # every rank's needed ids (its Store on its windowed mesh); rank r's objects are kept
    needed, keep = [], None
    t_setup = []
    for q in range(N):
        t0 = time.perf_counter()
        j0, j1 = mdist.row_block(g.ny, N, q)
        grid = R.Grid.from_target(g, rows=(j0, j1))
        mesh = R.Mesh.from_mpas(m, window_grid=grid)
        rh = R.regrid_store(mesh, grid, R.REGRIDMETHOD_BILINEAR)
        needed.append(rh.unique_sources())
        t_setup.append((time.perf_counter() - t0) * 1e3)
        if q == r:
            keep = (grid, mesh, rh, j0, j1)
        else:
            rh.release()
            mesh.destroy()
            grid.destroy()
    grid, mesh, rh, j0, j1 = keep
    plan = MC.plan_host(r, needed, m.nCells)
    assert plan["mode"] == "range", "the headline mesh is banded"
    rh.rebase(plan["base"], plan["n_local"])
    n_local, (own0, own1), (op0, op1) = plan["n_local"], plan["own"], plan["own_pos"]
    recv = [(a, b) for a, b in plan["recv_pos"] if b > a]
    strip = sum(b - a for a, b in recv)
    rows = F if lev_fast else F * nlev
    esz = (4 if args.io == "f32" else 8) * (nlev if lev_fast else 1)
    # the rank's neighbours folded onto itself: what it receives from them, it sends to itself (the same sizes)
    shape = (F, n_local, nlev) if lev_fast else (F * nlev, n_local)
    bufs = [torch.randn(shape, dtype=dt, device=dev) for _ in range(2)]
    out = torch.empty((F, nlev, j1 - j0, g.nx), dtype=dt, device=dev)
    sendbuf = torch.empty(rows * strip * (nlev if lev_fast else 1), dtype=dt, device=dev)
    recvbuf = torch.empty_like(sendbuf)
    splits = [sendbuf.numel()]

    def views(buf):
        v, off = [], 0
        for a, b in recv:
            n = b - a
            k = rows * n * (nlev if lev_fast else 1)
            v.append((buf[off:off + k].view((rows, n, nlev) if lev_fast else (rows, n)), (a, b)))
            off += k
        return v
    sv, rv = views(sendbuf), views(recvbuf)

    def exchange(local):
        """bench.py's HaloSchedule.exchange with this rank's two strips: strided copies in, ONE all_to_all_single, strided copies out."""
        for view, (a, b) in sv:                       # pack: what a neighbour wants of the own block (here: the same-sized strips)
            lo = op0 if a < op0 else op1 - (b - a)
            view.copy_(local[:, lo:lo + (b - a)])
        dist.all_to_all_single(recvbuf, sendbuf, splits, splits)
        for view, (a, b) in rv:
            local[:, a:b].copy_(view)

    halo_stream = torch.cuda.Stream(device=dev)
    pipe = {"n": 0, "halo_done": [torch.cuda.Event(), torch.cuda.Event()], "comp_done": [None, None], "xev": [], "kev": []}

    def exchange_into(b, record):
        with torch.cuda.stream(halo_stream):
            if pipe["comp_done"][b] is not None:
                halo_stream.wait_event(pipe["comp_done"][b])
            if record:
                x0, x1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                x0.record(halo_stream)
            exchange(bufs[b])
            if record:
                x1.record(halo_stream)
                pipe["xev"].append((x0, x1))
            pipe["halo_done"][b].record(halo_stream)

    def one_step(record):
        b = pipe["n"] % 2
        exchange_into(1 - b, record)
        torch.cuda.current_stream().wait_event(pipe["halo_done"][b])
        if record:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        if args.io == "f32":
            rh.regrid_typed(bufs[b].view(-1), nlev=nlev, nfields=F, layout=layout, out=out)
        else:
            rh.regrid(bufs[b].view(-1), nlev=nlev, nfields=F, layout=layout, out=out)
        if record:
            e1.record()
            pipe["kev"].append((e0, e1))
        if pipe["comp_done"][b] is None:
            pipe["comp_done"][b] = torch.cuda.Event()
        pipe["comp_done"][b].record()
        pipe["n"] += 1

    halo_stream.wait_stream(torch.cuda.current_stream())
    exchange_into(0, False)
    for _ in range(5):
        one_step(False)
    torch.cuda.synchronize()
    # (1) host issue time of one step into an IDLE device
    host = []
    for _ in range(30):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        one_step(False)
        host.append((time.perf_counter() - t0) * 1e6)
    torch.cuda.synchronize()
    # (2) the pipelined loop
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step(True)
    torch.cuda.synchronize()
    step_ms = (time.perf_counter() - t0) / args.steps * 1e3
    kern = float(np.median([a.elapsed_time(b) for a, b in pipe["kev"]]))
    exch = float(np.median([a.elapsed_time(b) for a, b in pipe["xev"]]))
    per_neighbour = max((b - a) for a, b in recv) * rows * esz if recv else 0
    xgmi = per_neighbour / (153e9 * 0.8) * 1e3 + 0.02 if recv else 0.0
    host_med = float(np.median(host))
    bound = max(kern, xgmi, host_med * 1e-3)
    res = {"workload": desc, "ranks": N, "rank": r, "rows": j1 - j0, "fields": F, "io": args.io, "layout": args.layout, "n_local": int(n_local),
           "neighbour_strips_cells": [int(b - a) for a, b in recv], "halo_bytes_received_per_step": int(strip * rows * esz),
           "host_issue_us_median": round(host_med, 1), "host_issue_us_min": round(float(min(host)), 1), "kernel_ms": round(kern, 4),
           "exchange_ms_self_loopback": round(exch, 4), "xgmi_est_ms_ASSUMED_153GBs_x0.8_plus_20us": round(xgmi, 4), "step_ms_pipelined": round(step_ms, 4),
           "implied_fields_per_s_if_all_ranks_alike": round(F / (bound * 1e-3), 0), "bound_by": "kernel" if bound == kern else "exchange (estimate)" if bound == xgmi else "host issue",
           "single_gpu_headline_kernel_ms_for_scale": "4.5-5.2 (13 fields, all rows)", "setup_ms_per_rank_geometry_plus_store": [round(x, 1) for x in t_setup],
           "note": "one GPU, the neighbours' strips exchanged to self through all_to_all_single on a one-rank RCCL group: NO scaling curve measured"}
    print(json.dumps(res))
    rh.release()
    mesh.destroy()
    grid.destroy()
    _lib.finalize()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
