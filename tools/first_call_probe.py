#!/usr/bin/env python3
"""What a single-shot run pays: the FIRST mpg_mesh_create / mpg_grid_create / RegridStore of each method in a fresh process
against the second (fresh objects, warm process).  MPASSIT is a single-shot tool (mpassit.F90:105-137): the first values are
what a run costs.  usage (GPU box): python tools/first_call_probe.py [--workload c4_3m_regional] [--order nearest,conserve,bilinear]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c4_3m_regional")
    ap.add_argument("--order", default="nearest,conserve,bilinear")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--settle", type=float, default=0.5, help="seconds between mpg_init and the first call (a driver reads its namelist and grid file there)")
    args = ap.parse_args()
    from mpassit_amd import _lib, regrid as R, workloads
    m, g, nlev, desc = workloads.workload(args.workload)
    t0 = time.perf_counter()
    _lib.init(0)
    print("# %s\nmpg_init %.2f ms" % (desc, 1e3 * (time.perf_counter() - t0)))
    time.sleep(args.settle)
    codes = {"bilinear": R.REGRIDMETHOD_BILINEAR, "nearest": R.REGRIDMETHOD_NEAREST_STOD, "conserve": R.REGRIDMETHOD_CONSERVE}
    for rep in range(args.reps):
        t0 = time.perf_counter()
        mesh = R.Mesh.from_mpas(m)
        t1 = time.perf_counter()
        grid = R.Grid.from_target(g)
        t2 = time.perf_counter()
        line = "rep %d  mesh_create %.2f ms  grid_create %.2f ms " % (rep, 1e3 * (t1 - t0), 1e3 * (t2 - t1))
        for name in args.order.split(","):
            t0 = time.perf_counter()
            rh = R.regrid_store(mesh, grid, codes[name])
            wall = 1e3 * (time.perf_counter() - t0)
            line += " | %s store_ms %.2f wall %.2f" % (name, rh.store_ms, wall)
            rh.release()
        print(line, flush=True)
        mesh.destroy()
        grid.destroy()


if __name__ == "__main__":
    main()
