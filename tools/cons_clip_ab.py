#!/usr/bin/env python3
"""Conservative Store with the Sutherland-Hodgman clip ("cons_clip" 0) against the boundary-integral form (1): times, and the two
matrices compared -- the same (row, col) set up to slivers, weights within 1e-12.  usage (GPU box): python tools/cons_clip_ab.py [workloads...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))


def main():
    from _parity_helpers import assert_csr_equal
    from mpassit_amd import _lib, regrid as R, workloads
    _lib.init(0)
    for wl in (sys.argv[1:] or ["c4_3m_regional", "c5_global_latlon", "c2_655k_global", "x_c4_polar"]):
        m, g, _, desc = workloads.workload(wl)
        res = {}
        for mode in (0, 1, 0, 1):
            _lib.tune("cons_clip", mode)
            ts = []
            for _ in range(3):
                mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_proj(g, fill_target=False)
                rh = R.regrid_store(mesh, grid, R.REGRIDMETHOD_CONSERVE)
                ts.append(rh.store_ms)
                if mode not in res:
                    res[mode] = rh.csr()
                rh.release()
                mesh.destroy()
                grid.destroy()
            print("%-18s cons_clip %d  store_ms %s" % (wl, mode, " ".join("%.2f" % t for t in ts)), flush=True)
        _lib.tune("cons_clip", 0)
        (rp0, c0, v0), (rp1, c1, v1) = res[0], res[1]
        common, only0, only1 = assert_csr_equal(rp0, c0, v0, rp1, c1, v1, m.nCells, tol=1e-11)
        key0 = np.repeat(np.arange(rp0.size - 1, dtype=np.int64), np.diff(rp0)) * m.nCells + c0
        key1 = np.repeat(np.arange(rp1.size - 1, dtype=np.int64), np.diff(rp1)) * m.nCells + c1
        _, i0, i1 = np.intersect1d(key0, key1, assume_unique=True, return_indices=True)
        d = np.abs(v0[i0] - v1[i1])
        P = rp0.size - 1
        rs0 = np.bincount(np.repeat(np.arange(P), np.diff(rp0)), weights=v0, minlength=P)
        rs1 = np.bincount(np.repeat(np.arange(P), np.diff(rp1)), weights=v1, minlength=P)
        print("%-18s entries %d / %d, common %d, only clip %d, only integral %d; max |dw| %.3e; max row-sum difference %.3e; row sums max %.15f / %.15f" % (
            wl, c0.size, c1.size, common, only0, only1, d.max(), np.abs(rs0 - rs1).max(), rs0.max(), rs1.max()), flush=True)


if __name__ == "__main__":
    main()
