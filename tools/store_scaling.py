#!/usr/bin/env python3
"""Does a rank's set-up scale with the row block it owns?  For rank 0 of 1 / 2 / 4 / 8 ranks (target rows split as
mpassit_amd.dist.row_block does): the geometry ingest -- mpg_mesh_create of the whole mesh against mpg_mesh_create_window cut
to the block -- and the three RegridStores on it, cold (fresh mesh / grid objects per repetition, nothing from the handle
cache), min of 3 after one warm-up of the process.  --check compares the windowed weights with the whole-mesh ones.
usage: python tools/store_scaling.py [--workload c4_3m_regional] [--check]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c4_3m_regional")
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()
    import numpy as np
    import torch
    from mpassit_amd import _lib, dist, regrid as R, workloads
    _lib.init(0)
    m, g, nlev, desc = workloads.workload(args.workload)
    methods = (("bilinear", R.REGRIDMETHOD_BILINEAR), ("nearest", R.REGRIDMETHOD_NEAREST_STOD), ("conserve", R.REGRIDMETHOD_CONSERVE))

    def one(world, windowed):
        j0, j1 = dist.row_block(g.ny, world, 0)
        best = {"geometry": 1e9, "bilinear": 1e9, "nearest": 1e9, "conserve": 1e9}
        info = None
        for _ in range(4):
            grid = R.Grid.from_target(g, rows=(j0, j1))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            mesh = R.Mesh.from_mpas(m, window_grid=grid if windowed else None)
            torch.cuda.synchronize()
            best["geometry"] = min(best["geometry"], 1e3 * (time.perf_counter() - t0))
            info = mesh.window_info()
            for name, code in methods:
                t0 = time.perf_counter()
                rh = R.regrid_store(mesh, grid, code)
                torch.cuda.synchronize()
                best[name] = min(best[name], 1e3 * (time.perf_counter() - t0))
                rh.release()
            mesh.destroy()
            grid.destroy()
        return (j0, j1), best, info

    print("# %s" % desc)
    print("%-5s %-10s %-8s %10s %10s %10s %10s %10s   %s" % ("ranks", "rows", "mesh", "geometry", "bilinear", "nearest", "conserve", "SUM ms", "resident cells / vertices"))
    sums = {}
    for world in (1, 2, 4, 8):
        for windowed in (False, True):
            rows, b, info = one(world, windowed)
            tot = sum(b.values())
            sums[(world, windowed)] = tot
            print("%-5d %-10s %-8s %10.2f %10.2f %10.2f %10.2f %10.2f   %d / %d" % (world, "%d..%d" % rows, "window" if windowed else "whole", b["geometry"],
                                                                                 b["bilinear"], b["nearest"], b["conserve"], tot, info[1], info[3]))
    print("# rank 0's geometry + Stores, 1 rank (whole mesh) over 8 ranks: whole mesh on every rank %.2fx, windowed %.2fx" % (
        sums[(1, False)] / sums[(8, False)], sums[(1, False)] / sums[(8, True)]))
    if args.check:
        for world in (2, 8):
            for r in sorted({0, world // 2, world - 1}):
                j0, j1 = dist.row_block(g.ny, world, r)
                grid = R.Grid.from_target(g, rows=(j0, j1))
                whole, cut = R.Mesh.from_mpas(m), R.Mesh.from_mpas(m, window_grid=grid)
                for name, code in methods:
                    a, b = R.regrid_store(whole, grid, code), R.regrid_store(cut, grid, code)
                    wa, wb = (a.csr(), b.csr()) if name == "conserve" else (a.weights(), b.weights())
                    ok = all(np.array_equal(x, y) for x, y in zip(wa, wb))
                    print("# check world %d rank %d %-9s identical to the whole-mesh build: %s" % (world, r, name, ok))
                    assert ok
                    a.release()
                    b.release()
                whole.destroy()
                cut.destroy()
                grid.destroy()


if __name__ == "__main__":
    main()
