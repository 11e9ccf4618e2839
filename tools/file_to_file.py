#!/usr/bin/env python3
"""File-to-file wall time at configuration-4 scale (3.0 M cells x 55 levels -> 1800x1060): MPAS-style CDF-5 history file
-> GPU regrid + post-ops -> WRF-style output file, once through host arrays (ncio converts / swaps on a host core) and
once with device-side ingest / egress (raw big-endian bytes, swapped on the GPU, fields device resident in between).
Reported in DESIGN.md s4.3 only -- never bench.py's `value`.  Files live in the directory given (default /dev/shm so
that the numbers are about the conversion passes, not a disk)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    from mpassit_amd import _lib, build, interp as I, io_nc, workloads
    build.build_ncio()
    _lib.init(0)
    d = sys.argv[1] if len(sys.argv) > 1 else "/dev/shm"
    wl = sys.argv[2] if len(sys.argv) > 2 else "c4_3m_regional"
    m, g, nz, _ = workloads.workload(wl, arrays=False)
    rng = np.random.default_rng(3)
    nsoil = 4
    f32 = lambda a: np.asarray(a, np.float32)           # noqa: E731
    base = f32(rng.standard_normal((m.nCells, nz)))
    hist = {"skintemp": f32(rng.uniform(250, 320, m.nCells)), "theta": base + 300, "rho": base * 0.1 + 1.0,
            "uReconstructZonal": base * 10, "uReconstructMeridional": base[::-1].copy() * 10,
            "zgrid": f32(np.sort(rng.uniform(0, 2.0e4, (m.nCells, nz + 1)), axis=1)), "tslb": f32(rng.uniform(260, 300, (m.nCells, nsoil)))}
    gpath, hpath = os.path.join(d, "f2f_grid.nc"), os.path.join(d, "f2f_hist.nc")
    t0 = time.perf_counter()
    io_nc.write_mpas_files(gpath, hpath, m, rng.uniform(0, 3000, m.nCells), [0.05, 0.25, 0.7, 1.5], hist, nz, nsoil)
    in_gb = os.path.getsize(hpath) / 1e9
    print("inputs written: hist %.2f GB in %.1f s" % (in_gb, time.perf_counter() - t0), flush=True)
    del hist, base
    cfg = I.InterpConfig(interp_diag=False, wrf_mod_vars=True, hist_2d=[("skintemp", "TSK")],
                         hist_3d=[("zgrid", "PHB"), ("theta", "T"), ("uReconstructZonal", "U"), ("uReconstructMeridional", "V"), ("rho", "MUB")],
                         hist_soil=[("tslb", "TSLB")])
    nml = dict(dx=3000.0, ref_lat=38.5, ref_lon=-97.5, truelat1=38.5, truelat2=38.5, stand_lon=-97.5)
    target = g
    outs = {}
    for mode in (False, True, False, True):
        opath = os.path.join(d, "f2f_out_%d.nc" % mode)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tm = {}
        res = io_nc.run(gpath, hpath, opath, target, cfg, namelist=nml, fmt=5, device_io=mode, timings=tm)
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
        out_gb = os.path.getsize(opath) / 1e9
        n3d = sum(1 for v in res.values() if getattr(v, "ndim", 0) == 3)
        print("%-22s %.2f s wall  (%.2f GB in, %.2f GB out, %d 3-D output fields -> %.1f fields/s file to file)"
              % ("device ingest/egress:" if mode else "host arrays:", t, in_gb, out_gb, n3d, n3d / t), flush=True)
        print("    " + "  ".join("%s %.2f" % kv for kv in tm.items()), flush=True)
        outs[mode] = opath
        del res
        torch.cuda.empty_cache()
    # a run of time levels (here: the same history file four times): weights once, captured time level, prefetching reader
    tm = {}
    jobs = [(hpath, None, os.path.join(d, "f2f_series_%d.nc" % k)) for k in range(4)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    io_nc.run_series(gpath, jobs, target, cfg, namelist=nml, timings=tm)
    print("run_series, 4 files:   %.2f s wall  (setup %.2f s, per file %s)" % (time.perf_counter() - t0, tm["setup_s"],
          " ".join("%.2f" % v for v in tm["files_s"])), flush=True)
    series_same = all(open(j[2], "rb").read() == open(outs[True], "rb").read() for j in jobs)
    print("series outputs identical to the single run:", series_same)
    for j in jobs:
        os.remove(j[2])
    same = open(outs[False], "rb").read() == open(outs[True], "rb").read() and series_same
    print("output files identical byte for byte:", same)
    for p in (gpath, hpath, outs[False], outs[True]):
        os.remove(p)
    return 0 if same else 1


if __name__ == "__main__":
    sys.exit(main())
