#!/usr/bin/env python3
"""Does RegridStore scale with the row block a rank owns?  One mesh object (whole geometry on the device, as every rank
holds it), target grids made of the first 1/N of the rows (mpassit_amd.dist.row_block, rank 0), the three Stores timed
cold (handle cache bypassed by a fresh Grid per repetition), min of 3.
usage: python tools/store_scaling.py [--workload c4_3m_regional]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c4_3m_regional")
    args = ap.parse_args()
    import torch
    from mpassit_amd import _lib, dist, regrid as R, workloads
    _lib.init(0)
    m, g, nlev, desc = workloads.workload(args.workload)
    t0 = time.perf_counter()
    mesh = R.Mesh.from_mpas(m)
    torch.cuda.synchronize()
    print("# %s\n# mesh geometry to the device (every rank, whole mesh): %.1f ms" % (desc, 1e3 * (time.perf_counter() - t0)))
    print("%-6s %-10s %12s %12s %12s   (ms, rows of rank 0)" % ("ranks", "rows", "bilinear", "nearest", "conserve"))
    for world in (1, 2, 4, 8):
        j0, j1 = dist.row_block(g.ny, world, 0)
        ms = []
        for method in (R.REGRIDMETHOD_BILINEAR, R.REGRIDMETHOD_NEAREST_STOD, R.REGRIDMETHOD_CONSERVE):
            best = 1e9
            for _ in range(3):
                grid = R.Grid.from_target(g, rows=(j0, j1))
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                rh = R.regrid_store(mesh, grid, method)
                torch.cuda.synchronize()
                best = min(best, 1e3 * (time.perf_counter() - t0))
                rh.release()
                grid.destroy()
            ms.append(best)
        print("%-6d %-10s %12.2f %12.2f %12.2f" % (world, "%d..%d" % (j0, j1), *ms))
    mesh.destroy()


if __name__ == "__main__":
    main()
