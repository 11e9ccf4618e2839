#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer entry point mpg_regrid (H2D + kernel + D2H), C4 workload, one 3-D field.
Reported in DESIGN.md only -- never bench.py's `value` (which is HBM-resident)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from mpassit_amd import _lib, regrid as R, workloads
    _lib.init(0)
    m, g, nlev, _ = workloads.workload("c4_3m_regional")
    mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_target(g)
    rh = R.regrid_store(mesh, grid, R.REGRIDMETHOD_BILINEAR)
    src = np.random.default_rng(0).standard_normal((nlev, m.nCells))
    out = np.empty((1, nlev, g.ny, g.nx))
    rh.regrid(src, nlev=nlev, out=out)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        rh.regrid(src, nlev=nlev, out=out)
        ts.append(time.perf_counter() - t0)
    t = min(ts)
    gb = (src.nbytes + out.nbytes) / 1e9
    print("host path: %.3f s per 3-D field (%.2f GB over PCIe, %.1f GB/s incl. staging alloc) -> %.2f fields/s" % (t, gb, gb / t, 1 / t))
    # pipelined typed entry point (mpg_regrid_typed): float64 and float32 I/O
    for dt in (np.float64, np.float32):
        s2 = src.astype(dt)
        o2 = np.empty((1, nlev, g.ny, g.nx), dt)
        rh.regrid_typed_host(s2, nlev=nlev, out=o2)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            rh.regrid_typed_host(s2, nlev=nlev, out=o2)
            ts.append(time.perf_counter() - t0)
        t = min(ts)
        gb = (s2.nbytes + o2.nbytes) / 1e9
        print("pipelined %s: %.3f s per 3-D field (%.2f GB over PCIe, %.1f GB/s both directions summed) -> %.2f fields/s"
              % (np.dtype(dt).name, t, gb, gb / t, 1 / t))


if __name__ == "__main__":
    main()
