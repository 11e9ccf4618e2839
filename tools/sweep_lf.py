#!/usr/bin/env python3
"""Interleaved A/B sweep of the LEVEL-fast (MPAS file order) Regrid kernels in one process, typed entry point
(mpg_regrid_typed_dev): float32 or float64 elements on either side, every variant checked bit for bit against the
first one before it is timed.
Usage (GPU box): python tools/sweep_lf.py --workload c4_3m_regional --io f32 --variants -1,0,1,2,3 [--be]"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c4_3m_regional")
    ap.add_argument("--fields", type=int, default=4)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--io", default="f32", choices=["f32", "f64"])
    ap.add_argument("--variants", default="-1,0,1,2,3")
    ap.add_argument("--be", action="store_true", help="both sides big-endian (MPG_TYPE_BE): the Fortran driver's file flow")
    ap.add_argument("--layout", default="lev_fast", choices=["lev_fast", "cell_fast"])
    ap.add_argument("--plain", action="store_true", help="float64 through mpg_regrid_dev (the headline entry point)")
    ap.add_argument("--nlev", type=int, default=0, help="override the workload's level count (alignment experiments)")
    ap.add_argument("--knob", default="", help="knob to sweep instead of lf_variant (e.g. a3_staged with --layout cell_fast)")
    ap.add_argument("--set", default="", help="knobs held fixed during the sweep: key=value[,key=value]")
    args = ap.parse_args()
    import torch
    from mpassit_amd import _lib, regrid as R, workloads
    _lib.init(0)
    m, g, nlev, desc = workloads.workload(args.workload)
    nlev = args.nlev or nlev
    mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_target(g)
    rh = R.regrid_store(mesh, grid, R.REGRIDMETHOD_BILINEAR)
    U = rh.unique_sources().size
    F = args.fields
    dt = torch.float32 if args.io == "f32" else torch.float64
    esz = 4.0 if args.io == "f32" else 8.0
    src = (torch.randn((F, nlev, m.nCells), dtype=torch.float64, device="cuda") * 30 + 280).to(dt)
    layout = R.LAYOUT_LEV_FAST if args.layout == "lev_fast" else R.LAYOUT_CELL_FAST
    if layout == R.LAYOUT_LEV_FAST:
        src = src.permute(0, 2, 1).contiguous()
    out = torch.empty((F, nlev, g.ny, g.nx), dtype=dt, device="cuda")
    alg = F * nlev * esz * (U + g.nx * g.ny) + g.nx * g.ny * 36.0
    knob = args.knob or "lf_variant"
    variants = [int(v) for v in args.variants.split(",")]

    def run():
        if args.plain:
            rh.regrid(src.view(-1), nlev=nlev, nfields=F, layout=layout, out=out)
        else:
            rh.regrid_typed(src.view(-1), nlev=nlev, nfields=F, layout=layout, out=out, scale=9.81, offset=-300.0, src_be=args.be, dst_be=args.be)
    for kv in filter(None, args.set.split(",")):
        _lib.tune(kv.split("=")[0], int(kv.split("=")[1]))
    times = {v: [] for v in variants}
    ref = None
    for rnd in range(args.rounds + 1):
        for v in variants:
            _lib.tune(knob, v)
            run()   # untimed (tile lists)
            torch.cuda.synchronize()
            if rnd == 0:
                if ref is None:
                    ref = out.clone()
                elif not torch.equal(out.view(torch.int32), ref.view(torch.int32)):   # bit patterns (big-endian results read as floats hold NaNs)
                    print("# variant %d DIFFERS from variant %d: max abs %g" % (v, variants[0], float((out.double() - ref.double()).abs().max())))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                run()
            e1.record()
            torch.cuda.synchronize()
            if rnd > 0:
                times[v].append(e0.elapsed_time(e1) / 3)
    _lib.tune(knob, {"nn_variant": 1, "store_boxes": 1}.get(knob, -1))
    res = []
    for v, ts in times.items():
        med, mn = float(np.median(ts)), float(np.min(ts))
        res.append(dict(workload=args.workload, io=args.io, layout=args.layout, knob=knob, variant=v, ms_med=med, ms_min=mn,
                        GBs_med=alg / med / 1e6, frac_of_8TBs=alg / med / 1e6 / 8000.0, fields_per_s=F / med * 1e3, kernel_choice=rh.kernel_choice()))
    res.sort(key=lambda r: r["ms_med"])
    print("# nlev %d" % nlev)
    print("# %s  U=%d  P=%d  alg bytes per launch %.3f GB" % (desc, U, g.nx * g.ny, alg / 1e9))
    for r in res:
        print(json.dumps(r))


if __name__ == "__main__":
    main()
