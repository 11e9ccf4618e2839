"""interp.F90:291-328 on the README's 1800 x 1060 Lambert grid, 55 levels: the three-call chain (mpg_rotate_winds_dev + two
Grid -> Grid Regrids) against the one-pass mpg_wind_destagger_dev, device-resident float64 mass winds.  Prints one JSON line
(ms per call, algorithmic bytes, fraction of the 8 TB/s HBM peak) and checks the two give the same bits.
    python tools/wind_chain_probe.py [--nx 1800 --ny 1060 --nlev 55 --reps 20 --f32]"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nx", type=int, default=1800)
    ap.add_argument("--ny", type=int, default=1060)
    ap.add_argument("--nlev", type=int, default=55)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--f32", action="store_true", help="U / V as NF90_FLOAT big-endian (what the Fortran driver's device flow stores)")
    ap.add_argument("--latlon", action="store_true", help="global lat-lon grid (periodic, pole caps, no rotation)")
    ap.add_argument("--host", action="store_true", help="HOST arrays in and out: mpg_rotate_winds + two mpg_regrid against mpg_wind_destagger (wall clock, link included)")
    args = ap.parse_args()
    import torch
    from mpassit_amd import _lib, regrid as R, target_grid as T
    _lib.init(0)
    if args.latlon:
        t = T.define_target_grid_params("lat-lon", nx=args.nx + 1, ny=args.ny + 1, stand_lon=0.0, is_regional=False)
    else:
        t = T.define_target_grid_params("lambert", args.nx + 1, args.ny + 1, dx=3000.0, dy=3000.0, ref_lat=38.5, ref_lon=-97.5, truelat1=38.5,
                                        truelat2=38.5, stand_lon=-97.5)
    grid = R.Grid.from_target(t)
    rot = not args.latlon
    nlev, P = args.nlev, t.nx * t.ny
    gen = torch.Generator(device="cuda")
    gen.manual_seed(1)
    um = torch.rand((nlev, t.ny, t.nx), dtype=torch.float64, device="cuda", generator=gen) * 40 - 20
    vm = torch.rand((nlev, t.ny, t.nx), dtype=torch.float64, device="cuda", generator=gen) * 40 - 20
    cosa = torch.as_tensor(np.ascontiguousarray(t.cosa), device="cuda") if rot else None
    sina = torch.as_tensor(np.ascontiguousarray(t.sina), device="cuda") if rot else None
    rh_u, rh_v = R.regrid_store_grid(grid, R.STAGGERLOC_EDGE1), R.regrid_store_grid(grid, R.STAGGERLOC_EDGE2)
    odt = torch.float32 if args.f32 else torch.float64
    es = 4 if args.f32 else 8

    def chain():
        a, b = um.clone(), vm.clone()
        if rot:
            R.rotate_winds_cgrid(cosa, sina, a, b)
        if args.f32:
            return rh_u.regrid_typed(a.view(-1), nlev=nlev, out_dtype=odt, dst_be=True)[0], rh_v.regrid_typed(b.view(-1), nlev=nlev, out_dtype=odt, dst_be=True)[0]
        return rh_u.regrid(a.view(-1), nlev=nlev)[0], rh_v.regrid(b.view(-1), nlev=nlev)[0]

    def fused():
        u, v, _, _ = R.wind_destagger(rh_u, rh_v, cosa, sina, um, vm, nlev, out_dtype=odt, dst_be=args.f32)
        return u, v

    def clone_only():
        return um.clone(), vm.clone()

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(args.reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts.sort()
        return ts[len(ts) // 2], ts[0]
    if args.host:
        import time
        umh, vmh = um.cpu().numpy(), vm.cpu().numpy()
        ca, sa = (np.ascontiguousarray(t.cosa), np.ascontiguousarray(t.sina)) if rot else (None, None)
        odn = np.float32 if args.f32 else np.float64

        def chain_h():
            a, b = umh.copy(), vmh.copy()
            t0 = time.perf_counter()
            if rot:
                R.rotate_winds_cgrid(ca, sa, a, b)
            if args.f32:
                r = rh_u.regrid_typed_host(a.reshape(-1), nlev=nlev, out_dtype=odn)[0], rh_v.regrid_typed_host(b.reshape(-1), nlev=nlev, out_dtype=odn)[0]
            else:
                r = rh_u.regrid(a.reshape(-1), nlev=nlev)[0], rh_v.regrid(b.reshape(-1), nlev=nlev)[0]
            return r, time.perf_counter() - t0

        def fused_h():
            t0 = time.perf_counter()
            u, v, _, _ = R.wind_destagger(rh_u, rh_v, ca, sa, umh, vmh, nlev, out_dtype=odn)
            return (u, v), time.perf_counter() - t0
        (uc, vc), _ = chain_h()
        (uf, vf), _ = fused_h()
        same = np.array_equal(uc.view(np.uint8), np.asarray(uf).view(np.uint8)) and np.array_equal(vc.view(np.uint8), np.asarray(vf).view(np.uint8))
        tc = sorted(chain_h()[1] for _ in range(max(3, args.reps // 4)))
        tf = sorted(fused_h()[1] for _ in range(max(3, args.reps // 4)))
        up_c, dn_c = nlev * P * 8 * (4 if rot else 2), nlev * (P * 16 if rot else 0) + nlev * (rh_u.n_dst + rh_v.n_dst) * es
        up_f, dn_f = nlev * P * 16, nlev * (rh_u.n_dst + rh_v.n_dst) * es
        print(json.dumps({"grid": "%dx%d" % (t.nx, t.ny), "nlev": nlev, "dst": "f32" if args.f32 else "f64", "host_arrays": True, "bits_equal": bool(same),
                          "chain_ms": round(tc[len(tc) // 2] * 1e3, 1), "fused_ms": round(tf[len(tf) // 2] * 1e3, 1), "speedup": round(tc[len(tc) // 2] / tf[len(tf) // 2], 2),
                          "chain_GB_up_down": [round(up_c / 1e9, 2), round(dn_c / 1e9, 2)], "fused_GB_up_down": [round(up_f / 1e9, 2), round(dn_f / 1e9, 2)]}))
        for rh in (rh_u, rh_v):
            rh.release()
        grid.destroy()
        _lib.finalize()
        return
    uc, vc = chain()
    uf, vf = fused()
    torch.cuda.synchronize()
    it = torch.int32 if args.f32 else torch.int64
    same = bool((uc.view(it) == uf.view(it)).all()) and bool((vc.view(it) == vf.view(it)).all())
    c_med, c_min = timed(chain)
    k_med, k_min = timed(clone_only)
    f_med, f_min = timed(fused)
    alg = nlev * (2 * P * 8 + (rh_u.n_dst + rh_v.n_dst) * es) + (rh_u.n_dst + rh_v.n_dst) * 48 + (P * 16 if rot else 0)
    alg_chain = alg + (nlev * P * 32 if rot else 0) + 0   # the rotation's read + write of both fields on top
    print(json.dumps({"grid": "%dx%d %s" % (t.nx, t.ny, "lat-lon periodic" if args.latlon else "Lambert 3 km"), "nlev": nlev, "dst": "f32be" if args.f32 else "f64",
                      "bits_equal": same, "chain_ms": round(c_med - k_med, 4), "chain_ms_min": round(c_min - k_min, 4), "clone_ms": round(k_med, 4),
                      "fused_ms": round(f_med, 4), "fused_ms_min": round(f_min, 4), "alg_bytes_fused": alg, "alg_bytes_chain": alg_chain,
                      "fused_frac_of_8TBs": round(alg / (f_med * 1e-3) / 8e12, 4), "chain_frac_of_8TBs": round(alg_chain / ((c_med - k_med) * 1e-3) / 8e12, 4),
                      "speedup": round((c_med - k_med) / f_med, 3)}))
    for rh in (rh_u, rh_v):
        rh.release()
    grid.destroy()
    _lib.finalize()


if __name__ == "__main__":
    main()
