#!/usr/bin/env python3
"""RegridStore times (mpg_handle_store_ms) of the three methods on a workload; fresh mesh / grid objects per repetition
so that nothing comes from the handle cache.  Usage (GPU box): python tools/store_timing.py --workload c4_3m_regional"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c4_3m_regional")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--methods", default="bilinear,nearest,conserve")
    ap.add_argument("--proj", action="store_true", help="target grid built on the device from its projection (mpg_grid_create_proj): the Stores "
                                                        "find their candidates through the inverse projection")
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=VALUE")
    args = ap.parse_args()
    from mpassit_amd import _lib, regrid as R, workloads
    _lib.init(0)
    for kv in args.tune:
        _lib.tune(kv.split("=")[0], int(kv.split("=")[1]))
    m, g, nlev, desc = workloads.workload(args.workload)
    codes = {"bilinear": R.REGRIDMETHOD_BILINEAR, "nearest": R.REGRIDMETHOD_NEAREST_STOD, "conserve": R.REGRIDMETHOD_CONSERVE}
    print("# %s%s%s" % (desc, "  [grid from its projection]" if args.proj else "", "  " + " ".join(args.tune) if args.tune else ""))
    for name in args.methods.split(","):
        ts, wall = [], []
        for _ in range(args.reps):
            mesh, grid = R.Mesh.from_mpas(m), (R.Grid.from_proj(g, fill_target=False) if args.proj else R.Grid.from_target(g))
            t0 = time.perf_counter()
            rh = R.regrid_store(mesh, grid, codes[name])
            wall.append((time.perf_counter() - t0) * 1e3)
            ts.append(rh.store_ms)
            rh.release()
            mesh.destroy()
            grid.destroy()
        print("%-9s store_ms min %.2f  (all: %s)   host wall of the call min %.2f ms" % (name, min(ts), " ".join("%.2f" % t for t in ts), min(wall)))


if __name__ == "__main__":
    main()
