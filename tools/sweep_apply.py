#!/usr/bin/env python3
"""Interleaved A/B sweep of the 3-point apply kernel variants in ONE process (cdna guide s5.4 rule 24).
Usage (GPU box): python tools/sweep_apply.py [--workload c4_3m_regional] [--fields 4] [--rounds 5]"""
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
    ap.add_argument("--variants", default="0,1,2,3,4,5,6,7,8,9,10,11,12")
    ap.add_argument("--chunks", default="0")
    ap.add_argument("--fpw", default="1,0")
    ap.add_argument("--tgroup", default="1")
    ap.add_argument("--lf", default="", help="comma list of lev_fast variants: sweeps the level-fastest kernel instead")
    args = ap.parse_args()
    import torch
    from mpassit_amd import _lib, regrid as R, workloads
    _lib.init(0)
    m, g, nlev, desc = workloads.workload(args.workload)
    mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_target(g)
    rh = R.regrid_store(mesh, grid, R.REGRIDMETHOD_BILINEAR)
    U = rh.unique_sources().size
    F = args.fields
    src = torch.randn((F, nlev, m.nCells), dtype=torch.float64, device="cuda")
    out = torch.empty((F, nlev, g.ny, g.nx), dtype=torch.float64, device="cuda")
    alg = F * nlev * 8.0 * (U + g.nx * g.ny) + g.nx * g.ny * 36.0
    combos = [(int(v), int(c), int(f) + 1000 * int(g)) for v in args.variants.split(",") for c in args.chunks.split(",")
              for f in args.fpw.split(",") for g in args.tgroup.split(",")]
    layout = R.LAYOUT_CELL_FAST
    if args.lf:
        combos = [(int(v), 0, int(f)) for v in args.lf.split(",") for f in args.fpw.split(",")]
        layout = R.LAYOUT_LEV_FAST
        src = src.permute(0, 2, 1).contiguous()
    times = {c: [] for c in combos}
    for rnd in range(args.rounds + 1):
        for v, c, fp in combos:
            if args.lf:
                _lib.tune("lf_variant", v)
                _lib.tune("lf_fields_per_wg", max(1, fp))
            else:
                _lib.tune("a3_staged", v - 100 if v >= 100 else -2)      # 100.. = LDS-staged cell-fast variants
                _lib.tune("a3_variant", v if v < 100 else 1)
                _lib.tune("lev_chunk", c)
                _lib.tune("fields_per_wg", fp % 1000)
                _lib.tune("tile_group", max(1, fp // 1000))
            rh.regrid(src.view(-1), nlev=nlev, nfields=F, out=out, layout=layout)   # untimed: a variant with another tile
            torch.cuda.synchronize()                                                # shape rebuilds the handle's tile lists
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                rh.regrid(src.view(-1), nlev=nlev, nfields=F, out=out, layout=layout)
            e1.record()
            torch.cuda.synchronize()
            if rnd > 0:
                times[(v, c, fp)].append(e0.elapsed_time(e1) / 3)
    res = []
    for (v, c, fp), ts in times.items():
        med, mn = float(np.median(ts)), float(np.min(ts))
        res.append(dict(variant=v, lev_chunk=c, fpw=fp, ms_med=med, ms_min=mn, GBs_med=alg / med / 1e6, fields_per_s=F / med * 1e3))
    res.sort(key=lambda r: r["ms_med"])
    for r in res:
        print(json.dumps(r))


if __name__ == "__main__":
    main()
