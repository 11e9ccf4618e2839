#!/usr/bin/env python3
"""One-off soak of the index-space candidate searches (Stores on grids that know their projection, meshes cut to a row block)
against the hierarchical searches on the whole mesh: many more random meshes / grids / row blocks than the committed fuzz
cases, wider parameter ranges (Lambert grids far from their standard longitude, wide regional lat-lon grids, coarse and fine
extremes).  Both sides are the library; every weight set must be the same bits.  Prints one line per failure and a summary.
usage (GPU box): python tools/fuzz_soak.py [--cases 300] [--seed 0]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def random_grid(rng):
    from mpassit_amd import target_grid as T
    kind = int(rng.integers(9))
    nx, ny = int(rng.integers(3, 160)), int(rng.integers(3, 100))
    if kind == 7:      # polar stereographic (round 5): either hemisphere, the pole inside the grid, beside it or far from it
        dx = float(rng.uniform(8e3, 150e3))
        south = rng.random() < 0.35
        sgn = -1.0 if south else 1.0
        lat = sgn * float(rng.choice([rng.uniform(86, 90), rng.uniform(55, 86), rng.uniform(20, 55)]))
        return T.define_target_grid_params("polar", nx, ny, dx=dx, dy=dx, ref_lat=lat, ref_lon=float(rng.uniform(-180, 180)),
                                           truelat1=sgn * float(rng.uniform(45, 90)), stand_lon=float(rng.uniform(-180, 180)))
    if kind == 8:      # Mercator (round 5): the known point anywhere in the grid (the map's cut follows the grid's middle), across the date line
        dx = float(rng.uniform(8e3, 150e3))
        kw = {}
        if rng.random() < 0.5:
            kw = dict(ref_x=float(rng.uniform(1, nx)), ref_y=float(rng.uniform(1, ny)))
        return T.define_target_grid_params("mercator", nx, ny, dx=dx, dy=dx, ref_lat=float(rng.uniform(-55, 55)), ref_lon=float(rng.uniform(-180, 180)),
                                           truelat1=float(rng.uniform(-50, 50)), stand_lon=0.0, **kw)
    if kind == 5:      # a fine regional lat-lon grid that reaches the last degrees before a pole (the boxes' pad follows the latitude)
        d = float(rng.uniform(0.05, 0.4))
        south = rng.random() < 0.5
        top = float(rng.uniform(84.0, 89.9))
        ny = int(rng.integers(20, 120))
        lat0 = top - d * (ny - 1)
        if south:
            lat0 = -top
        return T.define_target_grid_params("lat-lon", int(rng.integers(40, 300)), ny, dx=d, dy=d, ref_lat=lat0, ref_lon=float(rng.uniform(-180, 180)),
                                           ref_x=1.0, ref_y=1.0, stand_lon=0.0)
    if kind == 6:      # a Lambert grid close to its pole
        dx = float(rng.uniform(8e3, 60e3))
        stand = float(rng.uniform(-180, 180))
        lat = float(rng.uniform(70, 86)) * (-1 if rng.random() < 0.3 else 1)
        t = float(rng.uniform(50, 80)) * (1 if lat > 0 else -1)
        return T.define_target_grid_params("lambert", nx, ny, dx=dx, dy=dx, ref_lat=lat, ref_lon=stand + float(rng.uniform(-60, 60)), truelat1=t, truelat2=t,
                                           stand_lon=stand)
    if kind in (0, 3):
        dx = float(rng.uniform(8e3, 150e3))
        stand = float(rng.uniform(-180, 180))
        off = float(rng.uniform(-40, 40)) if kind == 0 else float(rng.uniform(-150, 150))   # kind 3: towards the projection's cut
        south = rng.random() < 0.25
        lat = float(rng.uniform(20, 60)) * (-1 if south else 1)
        t1, t2 = sorted([float(rng.uniform(20, 50)), float(rng.uniform(30, 65))])
        if south:
            t1, t2 = -t2, -t1
        return T.define_target_grid_params("lambert", nx, ny, dx=dx, dy=dx, ref_lat=lat, ref_lon=stand + off, truelat1=t1, truelat2=t2,
                                           stand_lon=stand)
    if kind in (1, 4):
        d = float(rng.uniform(0.05, 2.5)) if kind == 1 else float(rng.uniform(1.0, 2.0))
        if kind == 4:
            nx = int(rng.integers(100, 300))       # wide: up to most of the circle
        lat0 = float(rng.uniform(-70, 40))
        ny = max(3, min(ny, int((85 - lat0) / d)))
        return T.define_target_grid_params("lat-lon", nx, ny, dx=d, dy=d, ref_lat=lat0, ref_lon=float(rng.uniform(-180, 180)),
                                           ref_x=1.0, ref_y=1.0, stand_lon=0.0)
    return T.define_target_grid_params("lat-lon", int(rng.integers(8, 400)), int(rng.integers(5, 200)), stand_lon=float(rng.uniform(-180, 180)),
                                       is_regional=False)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    from mpassit_amd import _lib, regrid as R
    import test_fuzz_gpu as F
    _lib.init(0)
    bad = 0
    paths = {0: 0, 1: 0, 2: 0}
    t0 = time.time()
    for case in range(args.cases):
        rng = np.random.default_rng(900000 + 1000 * args.seed + case)
        m = F._mesh(rng, case % 4)
        if rng.random() < 0.12:                   # now and then a mesh much FINER than the grid (many cells per bin / box)
            from mpassit_amd import synth
            m = synth.icosahedral_mesh(int(rng.integers(6, 8)), order=["morton", "native"][int(rng.integers(2))])
        try:
            g = random_grid(rng)
        except Exception as e:                    # a parameter combination the namelist checks refuse
            print("# case %d: grid refused (%s)" % (case, e))
            continue
        whole = rng.random() < 0.3
        j0 = 0 if whole else int(rng.integers(0, max(1, g.ny - 2)))
        j1 = g.ny if whole else int(rng.integers(j0 + 1, g.ny + 1))
        res = []
        for cut in (False, True):
            _lib.tune("store_boxes", 1 if cut else 0)
            try:
                grid = R.Grid.from_target(g, rows=(j0, j1), attach_proj=cut)
                mesh = R.Mesh.from_mpas(m, window_grid=grid if (cut and not whole) else None)
                out = []
                for kw in (dict(regridmethod=R.REGRIDMETHOD_BILINEAR), dict(regridmethod=R.REGRIDMETHOD_NEAREST_STOD),
                           dict(regridmethod=R.REGRIDMETHOD_NEAREST_STOD, staggerloc=1), dict(regridmethod=R.REGRIDMETHOD_CONSERVE),
                           dict(regridmethod=R.REGRIDMETHOD_BILINEAR, meshloc=R.MESHLOC_NODE), dict(regridmethod=R.REGRIDMETHOD_BILINEAR, staggerloc=2)):
                    rh = R.regrid_store(mesh, grid, **kw)
                    out.append(rh.csr() if kw["regridmethod"] == R.REGRIDMETHOD_CONSERVE else rh.weights())
                    if cut and kw["regridmethod"] == R.REGRIDMETHOD_NEAREST_STOD:
                        paths[rh.store_path] += 1
                    rh.release()
                res.append(out)
                mesh.destroy()
                grid.destroy()
            finally:
                _lib.tune("store_boxes", 1)
        names = ("bilinear", "nearest", "nearest@edge1", "conserve", "bilinear@node", "bilinear@edge2")
        for name, a, b in zip(names, *res):
            same = all(np.array_equal(x, y) for x, y in zip(a, b))
            if not same:
                bad += 1
                print("FAIL case %d %s: mesh kind %d (%d cells), grid %s %dx%d rows %d..%d" % (case, name, case % 4, m.nCells, g.proj.code, g.nx, g.ny, j0, j1),
                      flush=True)
        if case % 25 == 24:
            print("# %d cases, %d failures, %.0f s; nearest Stores by path (tree / bins / bins + tree): %s" % (case + 1, bad, time.time() - t0, paths), flush=True)
    print("# done: %d cases, %d failures; nearest Stores by path: %s" % (args.cases, bad, paths))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
