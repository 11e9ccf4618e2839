#!/usr/bin/env python3
"""Where does a workgroup of the rows-resident level-fast kernel spend its life?  Runs the timestamped ablation
(lf_variant 350: s_memtime at the phase boundaries of every wave, written behind the destination array) on a workload and
prints the distribution of the phase durations.  Timing experiment only.
Usage (GPU box): python tools/lfs_stamps.py --workload c4_3m_regional --fields 13"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c4_3m_regional")
    ap.add_argument("--fields", type=int, default=13)
    ap.add_argument("--tick-ns", type=float, default=0.4167, help="s_memtime tick in ns (measured: ~core clock, 2.4 GHz)")
    args = ap.parse_args()
    import torch
    from mpassit_amd import _lib, regrid as R, workloads
    _lib.init(0)
    m, g, nlev, desc = workloads.workload(args.workload)
    mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_target(g)
    rh = R.regrid_store(mesh, grid, R.REGRIDMETHOD_BILINEAR)
    F = args.fields
    src = (torch.randn((F, m.nCells, nlev), dtype=torch.float64, device="cuda") * 30 + 280).float()
    n_out = F * nlev * g.ny * g.nx
    ntile = ((g.nx + 63) // 64) * g.ny
    nwg = ntile * F
    buf = torch.zeros(n_out + nwg * 4 * 8 * 2 + 16, dtype=torch.float32, device="cuda")
    out = buf[:n_out].view(F, nlev, g.ny, g.nx)
    for v in (300, 350, 350):
        _lib.tune("lf_variant", v)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rh.regrid_typed(src.view(-1), nlev=nlev, nfields=F, layout=R.LAYOUT_LEV_FAST, out=out)
        e1.record()
        torch.cuda.synchronize()
        print("variant %d: %.3f ms" % (v, e0.elapsed_time(e1)))
    _lib.tune("lf_variant", -1)
    raw = buf[n_out:n_out + nwg * 4 * 8 * 2].cpu().numpy().view(np.uint64).reshape(nwg, 4, 8)
    st = raw[:, :, :6].astype(np.int64)
    d = np.diff(st, axis=2) * args.tick_ns * 1e-3            # microseconds, [wg][wave][5]
    names = ["prologue: tile pointers -> cell list in LDS, ranks + weights landed", "row loads issued -> landed -> parked (this wave)",
             "barrier wait (slowest wave's rows)", "combine + stores issued", "stores acknowledged (vmcnt 0)"]
    print("# %s, %d workgroups x 4 waves; all times in microseconds" % (desc, nwg))
    for q, nm in enumerate(names):
        x = d[:, :, q].ravel()
        print("%-75s median %6.2f  mean %6.2f  p10 %6.2f  p90 %6.2f" % (nm, np.median(x), x.mean(), np.percentile(x, 10), np.percentile(x, 90)))
    life = (st[:, :, 5].max(axis=1) - st[:, :, 0].min(axis=1)) * args.tick_ns * 1e-3
    print("%-75s median %6.2f  mean %6.2f  p10 %6.2f  p90 %6.2f" % ("workgroup lifetime (first stamp -> last store acknowledged)", np.median(life), life.mean(),
                                                                    np.percentile(life, 10), np.percentile(life, 90)))
    span = (st[:, :, 5].max() - st[:, :, 0].min()) * args.tick_ns * 1e-3
    print("kernel span %.1f us; sum of lifetimes / span = %.1f workgroups in flight on average (%.2f per CU at 256 CUs)" % (
        span, life.sum() / span, life.sum() / span / 256))
    print("unique cells per tile: mean %.1f max %d" % (raw[:, 0, 6].mean(), raw[:, 0, 6].max()))


if __name__ == "__main__":
    main()
