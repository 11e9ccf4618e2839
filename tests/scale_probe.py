#!/usr/bin/env python3
"""Beyond the BASELINE sizes: the x1.10485762 (7.5 km, level 10) or x1.41943042 (3.75 km, level 11) global mesh on one MI355X -- the sizes
the 288 GB of HBM are for, and where 32-bit counts, offsets and scans would first break.  Synthetic icosahedral mesh (the construction
behind the MPAS x1 meshes) -> (a) the 1801x1061 3-km CONUS Lambert grid of configuration 4 and (b) a global 0.05-degree lat-lon grid
(7200 x 3600 = 25.9 M points).  For each: mesh creation, the three Stores (times, candidate path, entries), a Regrid of F 55-level float32
fields in file order with its fraction of the HBM roofline, and parity where the oracle can follow at this size:
  * bilinear and nearest on a SAMPLE of target points (a 256 x 256 patch + random points) against the oracle (its own search structures);
  * conservative on a 96 x 96 patch of destination cells against the oracle; every row of the global grid sums to 1;
  * a constant field comes back constant on every mapped point.
usage (GPU box): python tests/scale_probe.py [--level 10] [--fields 4] [--targets conus,global]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))   # (this script lives in tests/: it calls the oracle, which only tests may)


def lap(t0):
    return round(time.perf_counter() - t0, 3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--level", type=int, default=10)
    ap.add_argument("--fields", type=int, default=4)
    ap.add_argument("--targets", default="conus,global")
    ap.add_argument("--no-oracle", action="store_true")
    ap.add_argument("--tune", action="append", default=[], help="key=value for mpg_tune (A/B of a knob)")
    args = ap.parse_args()
    import torch

    from _parity_helpers import assert_csr_equal, assert_fixed_weights_equal
    from mpassit_amd import _lib, regrid as R, synth, target_grid as tg, workloads
    from oracle import oracle as O
    out = {"level": args.level}
    t0 = time.perf_counter()
    m = synth.icosahedral_mesh(args.level)
    out["mesh"] = {"nCells": int(m.nCells), "nVertices": int(m.nVertices), "generated_s": lap(t0)}
    print("mesh: %d cells, %d vertices, generated in %.0f s" % (m.nCells, m.nVertices, out["mesh"]["generated_s"]), flush=True)
    _lib.init(0)
    for kv in args.tune:
        _lib.tune(kv.split("=")[0], int(kv.split("=")[1]))
    t0 = time.perf_counter()
    mesh = R.Mesh.from_mpas(m)
    torch.cuda.synchronize()
    out["mesh"]["mpg_mesh_create_s"] = lap(t0)
    print("mpg_mesh_create: %.3f s" % out["mesh"]["mpg_mesh_create_s"], flush=True)
    if not args.no_oracle:
        O.build()
        t0 = time.perf_counter()
        lon_d, lat_d = O.mesh_coords_deg(m.lonCell, m.latCell)
        vlon_d, vlat_d = O.mesh_coords_deg(m.lonVertex, m.latVertex)
        cxyz, vxyz = O.lonlat_deg_to_xyz(lon_d, lat_d), O.lonlat_deg_to_xyz(vlon_d, vlat_d)
        tri_o, _ = O.dual_triangles(m.verticesOnCell, m.nVertices, cxyz)
        print("oracle: coordinates + dual triangles in %.1f s" % lap(t0), flush=True)
    nlev, F = 55, args.fields
    rng = np.random.default_rng(args.level)
    for tname in args.targets.split(","):
        if tname == "conus":
            g = workloads.conus_lambert_grid()
        else:
            g = tg.define_target_grid_params("lat-lon", 7201, 3601, stand_lon=0.0, is_regional=False)
        P = g.nx * g.ny
        res = {"grid": "%d x %d (%s)" % (g.nx, g.ny, tname), "points": int(P)}
        grid = R.Grid.from_target(g)
        hs = {}
        for name, meth in (("bilinear", R.REGRIDMETHOD_BILINEAR), ("nearest", R.REGRIDMETHOD_NEAREST_STOD), ("conserve", R.REGRIDMETHOD_CONSERVE)):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            h = R.regrid_store(mesh, grid, meth)
            torch.cuda.synchronize()
            hs[name] = h
            res["store_" + name] = {"wall_ms": round((time.perf_counter() - t0) * 1e3, 2), "store_ms": round(float(h.store_ms), 3), "path": int(h.store_path), "nnz": int(h.nnz)}
        print(tname, json.dumps({k: v for k, v in res.items() if k.startswith("store_")}), flush=True)
        # ---- Regrid: F float32 fields of 55 levels in file order ---------------------------------------------------------------
        rb = hs["bilinear"]
        src = torch.empty((F, m.nCells, nlev), dtype=torch.float32, device="cuda")
        base = torch.as_tensor(np.sin(3 * m.latCell) * np.cos(2 * m.lonCell) + 2.0, device="cuda", dtype=torch.float32)
        for f in range(F):
            src[f] = base[:, None] * (1.0 + 0.01 * f) + torch.arange(nlev, device="cuda", dtype=torch.float32)[None, :]
        dst = torch.empty((F, nlev, g.ny, g.nx), dtype=torch.float32, device="cuda")
        for _ in range(2):
            rb.regrid_typed(src.view(-1), nlev=nlev, nfields=F, layout=R.LAYOUT_LEV_FAST, out=dst)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            rb.regrid_typed(src.view(-1), nlev=nlev, nfields=F, layout=R.LAYOUT_LEV_FAST, out=dst)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        uniq = int(rb.unique_sources().size)
        mapped = int((rb.weights()[0][:, 0] >= 0).sum())
        alg = F * nlev * 4.0 * (uniq + P) + P * 36.0
        res["regrid_f32_file_order"] = {"fields": F, "ms": round(ms, 3), "unique_sources": uniq, "mapped_points": mapped, "alg_GB": round(alg / 1e9, 2),
                                        "frac_of_8TBs": round(alg / (ms * 1e-3) / 8e12, 3), "fields_per_s": round(F / (ms * 1e-3), 1),
                                        "kernel": rb.kernel_choice()}
        print(tname, "regrid", json.dumps(res["regrid_f32_file_order"]), flush=True)
        # constants stay constants
        c = torch.full((m.nCells, 3), 2.5, dtype=torch.float64, device="cuda")
        oc = rb.regrid_typed(c.view(-1), nlev=3, nfields=1, layout=R.LAYOUT_LEV_FAST)[0]
        idx_all, w_all = rb.weights()
        mp = torch.as_tensor(idx_all[:, 0] >= 0, device="cuda").view(g.ny, g.nx)
        res["constant_field_max_dev"] = float((oc[:, mp] - 2.5).abs().max())
        del src, dst, c, oc
        torch.cuda.empty_cache()
        rp, col, val = hs["conserve"].csr()
        rs = np.add.reduceat(val, rp[:-1][np.diff(rp) > 0]) if val.size else np.zeros(0)
        res["conserve_rows"] = {"rows_with_entries": int((np.diff(rp) > 0).sum()), "row_sum_max_dev_from_1": float(np.abs(rs - 1).max()) if tname == "global" else None}
        # ---- parity on what the oracle can follow -------------------------------------------------------------------------------
        if not args.no_oracle:
            j0, i0 = g.ny // 3, g.nx // 3
            patch = (np.arange(j0, j0 + 256)[:, None] * g.nx + np.arange(i0, i0 + 256)[None, :]).ravel()
            sample = np.unique(np.concatenate([patch, rng.integers(0, P, 150_000)]))
            pxyz = O.lonlat_deg_to_xyz(g.lon.ravel()[sample], g.lat.ravel()[sample])
            t0 = time.perf_counter()
            io_, wo_ = O.bilinear_weights(cxyz, tri_o, pxyz)
            ties = assert_fixed_weights_equal(io_, wo_, idx_all[sample], w_all[sample], tol=2e-10)
            near_o = O.nearest(cxyz, pxyz)
            near_g = hs["nearest"].weights()[0][sample, 0]
            diff = np.nonzero(near_o != near_g)[0]
            for q in diff:                                          # a differing index must be a distance tie
                d1 = np.linalg.norm(cxyz[near_o[q]] - pxyz[q])
                d2 = np.linalg.norm(cxyz[near_g[q]] - pxyz[q])
                assert abs(d1 - d2) <= 8 * np.finfo(float).eps * max(d1, d2), (q, d1, d2)
            n = 96
            sub = (np.arange(j0, j0 + n + 1)[:, None] * (g.nx + 1) + np.arange(i0, i0 + n + 1)[None, :]).ravel()
            kxyz = O.lonlat_deg_to_xyz(g.lon_c.ravel()[sub], g.lat_c.ravel()[sub])
            rpo, colo, valo = O.conserve(m.verticesOnCell, vxyz, n, n, kxyz)
            rows = (np.arange(j0, j0 + n)[:, None] * g.nx + np.arange(i0, i0 + n)[None, :]).ravel()
            cnt = (rp[rows + 1] - rp[rows]).astype(np.int64)
            take = np.concatenate([np.arange(rp[r], rp[r + 1]) for r in rows]) if cnt.sum() else np.zeros(0, np.int64)
            rpg = np.concatenate([[0], np.cumsum(cnt)])
            h_rad = float(np.linalg.norm(kxyz[0] - kxyz[1]))
            n_common, only_o, only_g = assert_csr_equal(rpo, colo, valo, rpg, col[take], val[take], m.nCells, tol=max(1e-11, 64 * np.finfo(float).eps / h_rad))
            res["oracle_parity"] = {"sample_points": int(sample.size), "bilinear_edge_ties": int(ties), "nearest_distance_ties": int(diff.size),
                                    "conserve_patch_entries": int(n_common), "conserve_only_oracle": int(only_o), "conserve_only_library": int(only_g),
                                    "oracle_s": lap(t0)}
        print(tname, json.dumps({k: res[k] for k in ("constant_field_max_dev", "conserve_rows", "oracle_parity") if k in res}), flush=True)
        for h in hs.values():
            h.release()
        grid.destroy()
        out[tname] = res
    mesh.destroy()
    free, total = torch.cuda.mem_get_info()
    out["hbm_free_after_GB"] = round(free / 1e9, 1)
    print(json.dumps(out))
    _lib.finalize()
    return 0


if __name__ == "__main__":
    sys.exit(main())
