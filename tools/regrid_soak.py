#!/usr/bin/env python3
"""One-off soak of the Regrid kernels: random meshes / grids (the generators of tests/test_fuzz_gpu.py and tools/fuzz_soak.py),
random level and field counts, element types, byte orders, layouts and weight sets (bilinear, nearest, conservative, the 4-point
destagger); every kernel family the knobs can select ("lf_variant", "a3_staged", "field_band") and the bundle call over separate
arrays must give the bits of the default path, with the source embedded between bands of NaNs and the destination between bands
of a canary (a load outside the slab that reaches a result shows as NaN, a store outside the destination breaks the canary).
Both sides are the library.  usage (GPU box): python tools/regrid_soak.py [--cases 200] [--seed 0]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
PAD = 4096
CANARY = -7.0e33


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    import torch
    from mpassit_amd import _lib, regrid as R
    import fuzz_soak as S
    import test_fuzz_gpu as F
    _lib.init(0)
    bad = runs = 0
    t0 = time.time()

    def embedded(a, fill):
        big = torch.full((a.numel() + 2 * PAD,), fill, dtype=a.dtype, device="cuda")
        big[PAD:PAD + a.numel()].copy_(a.reshape(-1))
        return big, big[PAD:PAD + a.numel()]

    for case in range(args.cases):
        rng = np.random.default_rng(400000 + 1000 * args.seed + case)
        m = F._mesh(rng, case % 4)
        if rng.random() < 0.15:                   # now and then a mesh with tens of thousands of cells (full tiles, long lists)
            from mpassit_amd import synth
            m = synth.icosahedral_mesh(int(rng.integers(5, 8)), order=["morton", "native"][int(rng.integers(2))])
        try:
            g = S.random_grid(rng)
        except Exception:
            continue
        mesh, grid = R.Mesh.from_mpas(m), R.Grid.from_target(g)
        kind = int(rng.integers(5))
        if kind == 0:
            rh = R.regrid_store(mesh, grid, R.REGRIDMETHOD_NEAREST_STOD)
        elif kind == 1:
            rh = R.regrid_store(mesh, grid, R.REGRIDMETHOD_CONSERVE)
        elif kind == 2:
            rh = R.regrid_store_grid(grid, [R.STAGGERLOC_EDGE1, R.STAGGERLOC_EDGE2][int(rng.integers(2))])
        else:
            rh = R.regrid_store(mesh, grid, R.REGRIDMETHOD_BILINEAR)
        nlev = int(rng.choice([1, 2, 3, 7, 8, 15, 16, 17, 31, 32, 33, 48, 55, 63, 64, 65, 70]))
        nf = int(rng.integers(1, 6))
        sdt = [torch.float32, torch.float64][int(rng.integers(2))]
        ddt = [torch.float32, torch.float64][int(rng.integers(2))]
        be = bool(rng.integers(2))
        layout = [R.LAYOUT_CELL_FAST, R.LAYOUT_LEV_FAST][int(rng.integers(2))]
        scale, offset = (1.0, 0.0) if rng.random() < 0.5 else (9.81, -300.0)
        gen = torch.Generator(device="cuda")
        gen.manual_seed(case)
        shape = (nf, rh.n_src, nlev) if layout == R.LAYOUT_LEV_FAST else (nf, nlev, rh.n_src)
        src = (torch.rand(shape, dtype=torch.float64, device="cuda", generator=gen) * 60.0 + 240.0).to(sdt)
        if be:
            src = src.view(torch.uint8).view(-1, src.element_size()).flip(1).contiguous().view(sdt).view(shape)
        it = torch.int32 if ddt == torch.float32 else torch.int64

        def run(guard):
            if guard:
                sbig, sview = embedded(src, float("nan"))
                out = torch.empty((nf, nlev, rh.ny_dst, rh.nx_dst), dtype=ddt, device="cuda")
                obig, oview = embedded(out, CANARY)
                rh.regrid_typed(sview, nlev=nlev, nfields=nf, layout=layout, out=oview.view(nf, nlev, rh.ny_dst, rh.nx_dst), scale=scale, offset=offset,
                                src_be=be, dst_be=be)
                torch.cuda.synchronize()
                ok = bool((obig[:PAD] == CANARY).all()) and bool((obig[-PAD:] == CANARY).all())
                return oview.clone().view(it), ok
            o = rh.regrid_typed(src.reshape(-1), nlev=nlev, nfields=nf, layout=layout, out_dtype=ddt, scale=scale, offset=offset, src_be=be, dst_be=be)
            return o.reshape(-1).view(it), True

        ref, _ = run(False)
        what = "case %d kind %d nlev %d nf %d %s->%s be %d layout %d mesh %d cells grid %dx%d" % (
            case, kind, nlev, nf, str(sdt)[-7:], str(ddt)[-7:], be, layout, m.nCells, g.nx, g.ny)
        combos = [("a3_staged", v) for v in (-2, 0, 1, 2)] + [("lf_variant", v) for v in (0, 1, 2)] + [("field_band", v) for v in (0, 3, 64)] + \
                 [("lfu_npf", v) for v in (16, 32)]        # round 6: row slots per class of tiles (default) against one size for all
        for knob, v in combos:
            if _lib.load().mpg_tune(knob.encode(), int(v)) != 0:
                continue
            try:
                got, ok = run(True)
            except Exception as e:          # a variant that does not fit this handle (refused, not wrong)
                got, ok = None, True
                if "UNSUPPORTED" not in str(e).upper() and "rc=4" not in str(e):
                    print("ERROR", what, knob, v, str(e)[:120], flush=True)
                    bad += 1
            finally:
                _lib.tune(knob, 0 if knob == "lfu_npf" else -1)
            runs += 1
            if got is not None and (not ok or not torch.equal(got, ref)):
                bad += 1
                print("FAIL", what, knob, v, "canary" if not ok else "bits", flush=True)
        # the bundle call over separate arrays, per-field offsets
        if nf > 1:
            srcs = [src[f].contiguous().reshape(-1) for f in range(nf)]
            offs = [offset + f for f in range(nf)]
            outs = rh.regrid_bundle(srcs, nlev=nlev, layout=layout, out_dtype=ddt, scale=scale, offsets=offs, src_be=be, dst_be=be)
            for f in range(nf):
                one = rh.regrid_typed(srcs[f], nlev=nlev, nfields=1, layout=layout, out_dtype=ddt, scale=scale, offset=offs[f], src_be=be, dst_be=be)
                runs += 1
                if not torch.equal(outs[f].reshape(-1).view(it), one.reshape(-1).view(it)):
                    bad += 1
                    print("FAIL", what, "bundle field", f, flush=True)
        rh.release()
        # round 6: the one-pass wind chain (mpg_wind_destagger_dev) against rotate_winds_cgrid + the two Grid -> Grid Regrids, on this grid
        if rng.random() < 0.6:
            nz = int(rng.choice([1, 2, 5, 16, 33]))
            um = (torch.rand((nz, g.ny, g.nx), dtype=torch.float64, device="cuda", generator=gen) - 0.5) * 80.0
            vm = (torch.rand((nz, g.ny, g.nx), dtype=torch.float64, device="cuda", generator=gen) - 0.5) * 80.0
            rot = getattr(g, "cosa", None) is not None and rng.random() < 0.7
            wdt, wbe = [(torch.float64, False), (torch.float32, True), (torch.float32, False)][int(rng.integers(3))]
            which = ["uv", "uv", "u", "v"][int(rng.integers(4))] if not rot else "uv"
            ru = R.regrid_store_grid(grid, R.STAGGERLOC_EDGE1) if "u" in which else None
            rv = R.regrid_store_grid(grid, R.STAGGERLOC_EDGE2) if "v" in which else None
            ca = torch.as_tensor(np.ascontiguousarray(g.cosa), device="cuda") if rot else None
            sa = torch.as_tensor(np.ascontiguousarray(g.sina), device="cuda") if rot else None
            a, b = um.clone(), vm.clone()
            if rot:
                R.rotate_winds_cgrid(ca, sa, a, b)
            wit = torch.int32 if wdt == torch.float32 else torch.int64

            def one(rh_s, mass):
                if wdt == torch.float64 and not wbe:
                    return rh_s.regrid(mass.reshape(-1), nlev=nz)[0]
                return rh_s.regrid_typed(mass.reshape(-1), nlev=nz, out_dtype=wdt, dst_be=wbe)[0]
            u0 = one(ru, a) if ru is not None else None
            v0 = one(rv, b) if rv is not None else None
            u1, v1, _, _ = R.wind_destagger(ru, rv, ca, sa, um if (ru is not None or rot) else None, vm if (rv is not None or rot) else None, nz,
                                            out_dtype=wdt, dst_be=wbe)
            torch.cuda.synchronize()
            for nm, x0, x1 in (("U", u0, u1), ("V", v0, v1)):
                if x0 is None:
                    continue
                runs += 1
                if not torch.equal(x0.reshape(-1).view(wit), x1.reshape(-1).view(wit)):
                    bad += 1
                    print("FAIL", what, "wind chain", nm, "rot %d nz %d %s be %d which %s" % (rot, nz, str(wdt)[-7:], wbe, which), flush=True)
            for r_ in (ru, rv):
                if r_ is not None:
                    r_.release()
        mesh.destroy()
        grid.destroy()
        if case % 25 == 24:
            print("# %d cases, %d comparisons, %d failures, %.0f s" % (case + 1, runs, bad, time.time() - t0), flush=True)
    print("# done: %d cases, %d comparisons, %d failures" % (args.cases, runs, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
