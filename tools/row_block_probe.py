#!/usr/bin/env python3
"""Regrid time of ROW BLOCKS of a workload's target grid, by block height: what a rank of a row-sharded job (or a user's grid of
any ny) gets per byte, against the full grid.  tools/rank_share_table.py found 133-row blocks of configuration 4 at 0.47 of the peak
where 132-row blocks reach 0.58 (float32 file order): with nx = 1800 a level plane of 132 rows is a whole number of 128-byte lines,
one of 133 rows is not, and every level of the result then starts 32 bytes further into a line.

For every height R in --rows: the block [j0, j0 + R) in the middle of the grid, its windowed mesh, bilinear Store, handle re-based
to the block's source range; F fields x nlev levels, resident.  --knob key --values a,b,...: the same launches under each setting of
an mpg_tune knob, interleaved, results compared bit for bit with the first setting's.  --dst-shift-bytes: the result array starts
that many bytes into a line (a control: an aligned plane size behind a misaligned base).

usage (GPU box): python tools/row_block_probe.py --rows 128,132,133,265 --io f32 --layout lev_fast --knob lf_rows_store --values 0,1"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", default="128,132,133,264,265,530,1060")
    ap.add_argument("--fields", type=int, default=13)
    ap.add_argument("--reps", type=int, default=12)
    ap.add_argument("--io", default="f32", choices=["f64", "f32"])
    ap.add_argument("--out", default=None, choices=[None, "f64", "f32"], help="result type (default: the source type)")
    ap.add_argument("--layout", default="lev_fast", choices=["cell_fast", "lev_fast"])
    ap.add_argument("--workload", default="c4_3m_regional")
    ap.add_argument("--knob", default=None)
    ap.add_argument("--values", default="")
    ap.add_argument("--dst-shift-bytes", default="0")
    args = ap.parse_args()
    import torch
    from mpassit_amd import _lib, regrid as R, workloads
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    _lib.init(0)
    m, g, nlev, desc = workloads.workload(args.workload)
    F = args.fields
    lev_fast = args.layout == "lev_fast"
    layout = R.LAYOUT_LEV_FAST if lev_fast else R.LAYOUT_CELL_FAST
    dt = torch.float32 if args.io == "f32" else torch.float64
    odt = dt if args.out is None else (torch.float32 if args.out == "f32" else torch.float64)
    es, oes = dt.itemsize, odt.itemsize
    typed = not (dt == torch.float64 and odt == torch.float64)
    values = [int(v) for v in args.values.split(",")] if args.knob else [None]
    shifts = [int(v) for v in args.dst_shift_bytes.split(",")]
    res = []
    for rows in [int(x) for x in args.rows.split(",")]:
        rows = min(rows, g.ny)
        j0 = (g.ny - rows) // 2
        grid = R.Grid.from_target(g, rows=(j0, j0 + rows))
        mesh = R.Mesh.from_mpas(m, window_grid=grid)
        rh = R.regrid_store(mesh, grid, R.REGRIDMETHOD_BILINEAR)
        ids = rh.unique_sources()
        base, n_local = int(ids[0]), int(ids[-1]) + 1 - int(ids[0])
        rh.rebase(base, n_local)
        P = rows * g.nx
        src = torch.randn((F, n_local, nlev) if lev_fast else (F * nlev, n_local), dtype=dt, device=dev)
        alg = F * nlev * (es * len(ids) + oes * P) + P * 36
        for sh in shifts:
            assert sh % oes == 0
            raw = torch.empty(F * nlev * P + 64, dtype=odt, device=dev)
            assert raw.data_ptr() % 256 == 0
            out = raw[sh // oes: sh // oes + F * nlev * P].view(F, nlev, rows, g.nx)

            def launch():
                if typed:
                    rh.regrid_typed(src.view(-1), nlev=nlev, nfields=F, layout=layout, out=out)
                else:
                    rh.regrid(src.view(-1), nlev=nlev, nfields=F, layout=layout, out=out)
            ms = {v: [] for v in values}
            ref = None
            for v in values:
                if args.knob:
                    _lib.tune(args.knob, v)
                out.zero_()
                launch()
                torch.cuda.synchronize()
                if ref is None:
                    ref = out.clone()
                else:
                    assert torch.equal(out.view(torch.int32 if oes == 4 else torch.int64), ref.view(torch.int32 if oes == 4 else torch.int64)), "results differ under %s=%s" % (args.knob, v)
                launch()
            for _ in range(args.reps):
                for v in values:
                    if args.knob:
                        _lib.tune(args.knob, v)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    launch()
                    e1.record()
                    torch.cuda.synchronize()
                    ms[v].append(e0.elapsed_time(e1))
            row = {"rows": rows, "plane_bytes_mod_128": (P * oes) % 128, "dst_shift_bytes": sh, "unique_src_cells": int(len(ids)), "alg_GB": round(alg / 1e9, 3)}
            for v in values:
                k = float(np.median(ms[v]))
                tag = "" if v is None else "%s=%d:" % (args.knob, v)
                row[tag + "ms"] = round(k, 4)
                row[tag + "frac"] = round(alg / (k * 1e-3) / 8e12, 3)
            res.append(row)
            print("# " + json.dumps(row), file=sys.stderr, flush=True)
            del raw, out, ref
        if args.knob:
            _lib.tune(args.knob, values[0])
        del src
        rh.release()
        mesh.destroy()
        grid.destroy()
        torch.cuda.empty_cache()
    print(json.dumps({"workload": desc, "fields": F, "nlev": nlev, "io": args.io, "out": args.out or args.io, "layout": args.layout, "knob": args.knob, "table": res}))
    _lib.finalize()


if __name__ == "__main__":
    main()
