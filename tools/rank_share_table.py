#!/usr/bin/env python3
"""EVERY rank's share of the N-GPU headline step for N = 1, 2, 4, 8, each measured on ONE GPU, one after the other: the row split's load
balance, i.e. the compute side of the strong-scaling curve bench.py --gpus N would give (value = fields x steps / MAX over ranks).

tools/rank_share_rehearsal.py times one rank of one split with the whole pipeline around it; this tool asks the other question: do all the
ranks of a split look like that one?  Row blocks of a Lambert grid do not reference equal numbers of source cells (the rows towards the
domain's corners see the mesh's rim), a rank's kernel is bound by its own bytes, and the job runs at the pace of its slowest rank.  For
every N and every rank r it builds what that rank would hold -- its row block of the target grid (model_grid.F90:693), its windowed mesh
(mpg_mesh_create_window), its bilinear Store, the halo plan out of all N ranks' needed ids (mpg_halo_plan_host) with the handle re-based to
the rank's local source range -- and times the rank's Regrid of F fields x nlev levels on a resident local slab (HIP events, median).

Reported per N: every rank's rows / local source cells / halo cells received / kernel ms, the slowest rank, and
  compute_bound_speedup = kernel_ms(N = 1) / max over ranks kernel_ms(N)
which is what the job would deliver IF the exchange hides behind the kernel (rehearsed to self in rank_share_rehearsal.py; over xGMI NOT
measured: one-GPU boxes).  It is an UPPER bound on the curve, measured; not the curve.

usage (GPU box): python tools/rank_share_table.py [--ranks 1,2,4,8] [--fields 13] [--io f64|f32] [--layout cell_fast|lev_fast] [--workload c4_3m_regional]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", default="1,2,4,8")
    ap.add_argument("--fields", type=int, default=13)
    ap.add_argument("--reps", type=int, default=12)
    ap.add_argument("--io", default="f64", choices=["f64", "f32"])
    ap.add_argument("--layout", default="cell_fast", choices=["cell_fast", "lev_fast"])
    ap.add_argument("--workload", default="c4_3m_regional")
    ap.add_argument("--split", default="aligned", choices=["aligned", "para_range"],
                    help="aligned: dist.row_block with the grid's width (block boundaries on whole alignment quanta: what ShardedRegrid / bench.py --gpus N use); para_range: the plain split")
    args = ap.parse_args()
    import torch
    from mpassit_amd import _lib, comm as MC, dist as mdist, regrid as R, workloads
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    _lib.init(0)
    m, g, nlev, desc = workloads.workload(args.workload)
    F = args.fields
    lev_fast = args.layout == "lev_fast"
    layout = R.LAYOUT_LEV_FAST if lev_fast else R.LAYOUT_CELL_FAST
    dt = torch.float32 if args.io == "f32" else torch.float64
    es = 4 if args.io == "f32" else 8
    table = []
    t1 = None
    warm = False
    for N in [int(x) for x in args.ranks.split(",")]:
        objs, needed = [], []
        for q in range(N):
            j0, j1 = mdist.row_block(g.ny, N, q, nx=g.nx if args.split == "aligned" else None)
            grid = R.Grid.from_target(g, rows=(j0, j1))
            mesh = R.Mesh.from_mpas(m, window_grid=grid)
            rh = R.regrid_store(mesh, grid, R.REGRIDMETHOD_BILINEAR)
            needed.append(rh.unique_sources())
            objs.append((grid, mesh, rh, j0, j1))
        per_rank = []
        for q, (grid, mesh, rh, j0, j1) in enumerate(objs):
            plan = MC.plan_host(q, needed, m.nCells)
            mode = plan["mode"]
            if mode == "range":
                rh.rebase(plan["base"], plan["n_local"])
            else:
                rh.localize()
            n_local = int(plan["n_local"])
            halo = int(sum(max(0, b - a) for a, b in plan["recv_pos"])) if mode != "owned" else None
            shape = (F, n_local, nlev) if lev_fast else (F * nlev, n_local)
            src = torch.randn(shape, dtype=dt, device=dev)
            out = torch.empty((F, nlev, j1 - j0, g.nx), dtype=dt, device=dev)

            def launch():
                if args.io == "f32":
                    rh.regrid_typed(src.view(-1), nlev=nlev, nfields=F, layout=layout, out=out)
                else:
                    rh.regrid(src.view(-1), nlev=nlev, nfields=F, layout=layout, out=out)
            for _ in range(3 if warm else 40):   # the first measurement of a process is slow by up to 10 % (clocks, first-touch): N = 1 is everybody's yardstick
                launch()
            warm = True
            torch.cuda.synchronize()
            ms = []
            for _ in range(args.reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                launch()
                e1.record()
                torch.cuda.synchronize()
                ms.append(e0.elapsed_time(e1))
            k = float(np.median(ms))
            n_unique = int(len(needed[q]))
            alg = F * nlev * es * (n_unique + (j1 - j0) * g.nx) + (j1 - j0) * g.nx * 36
            per_rank.append({"rank": q, "rows": j1 - j0, "unique_src_cells": n_unique, "n_local": n_local, "halo_cells_received": halo, "halo_mode": mode,
                             "kernel_ms": round(k, 4), "alg_GB": round(alg / 1e9, 3), "frac_of_8TBs": round(alg / (k * 1e-3) / 8e12, 3)})
            del src, out
        for grid, mesh, rh, _, _ in objs:
            rh.release()
            mesh.destroy()
            grid.destroy()
        torch.cuda.empty_cache()
        worst = max(per_rank, key=lambda x: x["kernel_ms"])
        best = min(per_rank, key=lambda x: x["kernel_ms"])
        if N == 1:
            t1 = worst["kernel_ms"]
        row = {"N": N, "slowest_rank": worst["rank"], "kernel_ms_max": worst["kernel_ms"], "kernel_ms_min": best["kernel_ms"],
               "imbalance_max_over_mean": round(worst["kernel_ms"] / float(np.mean([x["kernel_ms"] for x in per_rank])), 3),
               "fields_per_s_if_exchange_hidden": round(F / (worst["kernel_ms"] * 1e-3), 0),
               "compute_bound_speedup": round(t1 / worst["kernel_ms"], 2) if t1 else None, "ranks": per_rank}
        table.append(row)
        print("# N = %d: kernel ms per rank %s -> max %.3f (rank %d), compute-bound speed-up %s" % (
            N, [x["kernel_ms"] for x in per_rank], worst["kernel_ms"], worst["rank"], row["compute_bound_speedup"]), file=sys.stderr, flush=True)
    print(json.dumps({"workload": desc, "fields": F, "nlev": nlev, "io": args.io, "layout": args.layout, "split": args.split, "table": table,
                      "note": "one GPU, the ranks one after the other: the row split's load balance (an upper bound on the strong-scaling curve), NOT a measurement of N GPUs"}))
    _lib.finalize()


if __name__ == "__main__":
    main()
