#!/usr/bin/env python3
"""One Regrid per field against one Regrid per bundle: float32 file-order fields as a driver holds them (SEPARATE device arrays,
one per variable read from the file), (a) one mpg_regrid_typed_dev per field, (b) mpg_regrid_bundle_typed_dev over the separate
arrays, (c) the same fields copied into consecutive slabs and regridded with nfields = F (what bench.py times).
usage (GPU box): python tools/bundle_timing.py [--workload c4_3m_regional] [--fields 13]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c4_3m_regional")
    ap.add_argument("--fields", type=int, default=13)
    ap.add_argument("--rounds", type=int, default=5)
    args = ap.parse_args()
    import torch
    from mpassit_amd import _lib, regrid as R, workloads
    _lib.init(0)
    m, g, nlev, desc = workloads.workload(args.workload)
    mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_target(g)
    rh = R.regrid_store(mesh, grid, R.REGRIDMETHOD_BILINEAR)
    F = args.fields
    U = rh.unique_sources().size
    alg = F * nlev * 4.0 * (U + g.nx * g.ny) + g.nx * g.ny * 36.0
    srcs = [(torch.randn((m.nCells, nlev), dtype=torch.float64, device="cuda") * 30 + 280).float() for _ in range(F)]
    outs = [torch.empty((nlev, g.ny, g.nx), dtype=torch.float32, device="cuda") for _ in range(F)]
    slab = torch.stack(srcs).contiguous()
    out_slab = torch.empty((F, nlev, g.ny, g.nx), dtype=torch.float32, device="cuda")
    lay = R.LAYOUT_LEV_FAST

    def single():
        for f in range(F):
            rh.regrid_typed(srcs[f].view(-1), nlev=nlev, nfields=1, layout=lay, out=outs[f])

    def bundle():
        rh.regrid_bundle(srcs, nlev=nlev, layout=lay, outs=outs)

    def slabs():
        rh.regrid_typed(slab.view(-1), nlev=nlev, nfields=F, layout=lay, out=out_slab)
    single()
    ref = [o.clone() for o in outs]
    bundle()
    assert all(torch.equal(a.view(torch.int32), b.view(torch.int32)) for a, b in zip(ref, outs))
    slabs()
    assert all(torch.equal(out_slab[f].view(torch.int32), ref[f].view(torch.int32)) for f in range(F))
    res = {}
    for rnd in range(args.rounds):
        for name, fn in (("one call per field", single), ("bundle of separate arrays", bundle), ("bundle of consecutive slabs", slabs)):
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                fn()
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(name, []).append(e0.elapsed_time(e1) / 3)
    print("# %s, %d float32 file-order fields, kernel choice %s" % (desc, F, rh.kernel_choice()))
    for name, ts in res.items():
        med = float(np.median(ts))
        print("%-28s %7.3f ms   %6.0f fields/s   %.3f of 8 TB/s" % (name, med, F / med * 1e3, alg / med / 1e6 / 8000.0))


if __name__ == "__main__":
    main()
