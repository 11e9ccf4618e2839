#!/usr/bin/env python3
"""The whole configuration-4 job, file to file, on one GPU: 3.0 M-cell MPAS grid / history / diag files (CDF-5, float32 as
MPAS writes them; the reference's default histlist_2d / histlist_3d / histlist_soil and a 19-entry diaglist) ->
1801x1061 Lambert WRF-style file.  Three hosts over the same HIP library:
  fortran   mpassit_amd/fortran/mpassit <namelist>  (the north-star driver; host arrays, ncio conversions)
  python    io_nc.run(device_io=False)              (host arrays)
  python    io_nc.run(device_io=True)               (raw big-endian bytes file <-> GPU, fields device resident)
Wall seconds of each, and a comparison of the three output files.  DESIGN.md s6 quotes the numbers; bench.py does not."""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

NAMELIST = """&config
  grid_file_input_grid="{d}/init.nc"
  hist_file_input_grid="{d}/hist.nc"
  diag_file_input_grid="{d}/diag.nc"
  output_file="{d}/out_fortran.nc"
  target_grid_type = 'lambert'
  interp_diag=.true.
  interp_hist=.true.
  wrf_mod_vars=.true.
  esmf_log=.false.
  nx = 1801
  ny = 1061
  dx = 3000.0
  dy = 3000.0
  ref_lat = 38.50
  ref_lon = -97.50
  truelat1 = 38.5
  truelat2 = 38.5
  stand_lon = -97.5
/
"""


NAMELIST_C5 = """&config
  grid_file_input_grid="{d}/init.nc"
  hist_file_input_grid="{d}/hist.nc"
  diag_file_input_grid="{d}/diag.nc"
  output_file="{d}/out_fortran.nc"
  target_grid_type = 'lat-lon'
  interp_diag=.true.
  interp_hist=.true.
  wrf_mod_vars=.true.
  is_regional=.false.
  nx = 3601
  ny = 1801
  stand_lon = 0.0
/
"""


def main():
    import torch
    from test_fields import HIST_2D, HIST_3D, SOIL
    c5 = os.environ.get("C4JOB_CONFIG") == "5"      # configuration 5: 2.6 M-cell global mesh -> 3600x1800 global lat-lon grid
    if c5:
        HIST_3D = [p for p in HIST_3D if p[0] in ("zgrid", "theta", "uReconstructZonal", "uReconstructMeridional", "pressure", "qv")]

    from mpassit_amd import _lib, build, interp as I, io_nc, ncio, synth, workloads
    build.build()
    exe = build.build_fortran()
    _lib.init(0)
    d = sys.argv[1] if len(sys.argv) > 1 else "/dev/shm/c4job"
    os.makedirs(d, exist_ok=True)
    if c5:
        m, g, nz, _ = workloads.workload("c5_global_latlon")
    else:
        m, g, nz, _ = workloads.workload("c4_3m_regional", arrays=False)
    nsoil = 4
    diag_list = [("u10", "U10"), ("v10", "V10"), ("refl10cm", "REFL_10CM")] + [("d2d_%02d" % k, "D2D_%02d" % k) for k in range(2 if c5 else 16)]
    rng = np.random.default_rng(4)
    f32 = lambda a: np.asarray(a, np.float32)           # noqa: E731
    t0 = time.perf_counter()
    base = f32(rng.standard_normal((m.nCells, nz + 1)))
    hist, diag = {}, {}
    for k, (n, _) in enumerate(HIST_2D):
        hist[n] = f32(np.floor(rng.uniform(1, 3, m.nCells))) if n == "xland" else \
            (f32(synth.snow_field(m.latCell, m.lonCell)[0] * (1 + k)) if n in ("snow", "snowh") else f32(rng.uniform(250, 320, m.nCells)))
    for k, (n, _) in enumerate(HIST_3D):
        L = nz + 1 if n in ("zgrid", "w") else nz
        a = np.roll(base[:, :L], k, axis=0) * np.float32(1 + k)
        if n == "zgrid":
            a = np.cumsum(np.abs(a) * 300, axis=1, dtype=np.float32)
        if n == "pressure":
            a = np.float32(1.0e5) - np.cumsum(np.abs(a) * 100, axis=1, dtype=np.float32)
        hist[n] = np.ascontiguousarray(a, np.float32)
    for n, _ in SOIL:
        hist[n] = f32(rng.uniform(0, 1, (m.nCells, nsoil)))
    for k, (n, _) in enumerate(diag_list):
        diag[n] = np.ascontiguousarray(np.roll(base[:, :nz], 7, axis=0)) if n == "refl10cm" else f32(rng.normal(0, 8, m.nCells))
    ter = rng.uniform(0, 3000, m.nCells)
    io_nc.write_mpas_files(os.path.join(d, "init.nc"), os.path.join(d, "hist.nc"), m, ter, [0.05, 0.25, 0.7, 1.5], hist, nz, nsoil,
                           diag_path=os.path.join(d, "diag.nc"), diag=diag, xtime="2024-08-07_12:00:00", fmt=5)
    if os.environ.get("C4JOB_NC4") and ncio.has_netcdf4():   # the same inputs once more as NetCDF-4 files (HDF5 containers; MPAS's io_type "netcdf4")
        t4 = time.perf_counter()
        os.makedirs(os.path.join(d, "nc4"), exist_ok=True)
        io_nc.write_mpas_files(os.path.join(d, "nc4", "init.nc"), os.path.join(d, "nc4", "hist.nc"), m, ter, [0.05, 0.25, 0.7, 1.5], hist, nz, nsoil,
                               diag_path=os.path.join(d, "nc4", "diag.nc"), diag=diag, xtime="2024-08-07_12:00:00", fmt=4)
        print("the same inputs as NetCDF-4 files: written in %.0f s" % (time.perf_counter() - t4), flush=True)
    del hist, diag, base
    gb_in = sum(os.path.getsize(os.path.join(d, f)) for f in ("init.nc", "hist.nc", "diag.nc")) / 1e9
    print("inputs: %.2f GB (grid + history + diag) generated and written in %.0f s" % (gb_in, time.perf_counter() - t0), flush=True)
    for fname, lst in (("diaglist", diag_list), ("histlist_2d", HIST_2D), ("histlist_3d", HIST_3D), ("histlist_soil", SOIL)):
        with open(os.path.join(d, fname), "w") as f:
            f.write("".join("%s\t\t%s\n" % p for p in lst))
    open(os.path.join(d, "namelist.input"), "w").write((NAMELIST_C5 if c5 else NAMELIST).format(d=d))

    for nt in [v for v in os.environ.get("C4JOB_NCIO_THREADS", "").split(",") if v]:   # optional sweep of ncio's thread count
        t0 = time.perf_counter()
        r = subprocess.run([exe, "namelist.input"], cwd=d, capture_output=True, text=True, timeout=900, env=dict(os.environ, NCIO_THREADS=nt))
        if r.returncode != 0:
            print("fortran driver (NCIO_THREADS=%s) failed with rc %d" % (nt, r.returncode))
            print(r.stdout[-3000:], r.stderr[-3000:])
            return 1
        print("fortran driver, NCIO_THREADS=%s: %.2f s   %s" % (nt, time.perf_counter() - t0, "  ".join(
            ln.strip() for ln in r.stdout.splitlines() if ln.lstrip().startswith("["))), flush=True)
    t0 = time.perf_counter()       # the driver with host arrays (what it does for raw-container inputs / outputs)
    r = subprocess.run([exe, "namelist.input"], cwd=d, capture_output=True, text=True, timeout=900, env=dict(os.environ, MPASSIT_HOST_ARRAYS="1"))
    if r.returncode != 0:          # the driver's own message, not a FileNotFoundError from the rename below
        print("fortran driver (host arrays) failed with rc %d" % r.returncode)
        print(r.stdout[-3000:], r.stderr[-3000:])
        return 1
    print("fortran, host arrays:  %.2f s wall   %s" % (time.perf_counter() - t0, "  ".join(
        ln.strip() for ln in r.stdout.splitlines() if ln.lstrip().startswith("["))), flush=True)
    os.replace(os.path.join(d, "out_fortran.nc"), os.path.join(d, "out_fortran_host.nc"))
    t0 = time.perf_counter()       # default for NetCDF in and out: fields device resident between the files
    r = subprocess.run([exe, "namelist.input"], cwd=d, capture_output=True, text=True, timeout=900)
    t_f = time.perf_counter() - t0
    if r.returncode != 0:
        print(r.stdout[-3000:], r.stderr[-3000:])
        return 1
    gb_out = os.path.getsize(os.path.join(d, "out_fortran.nc")) / 1e9
    print("fortran, device flow:  %.2f s wall (process start to exit; %.2f GB out)" % (t_f, gb_out), flush=True)
    print("    " + "  ".join(ln.strip() for ln in r.stdout.splitlines() if ln.lstrip().startswith("[")), flush=True)

    if os.environ.get("C4JOB_NC4"):                   # C4JOB_NC4=1: the same job with a NetCDF-4 output (what the reference creates): host arrays, libhdf5
        from mpassit_amd import ncio
        if not ncio.has_netcdf4():
            print("NetCDF-4 leg skipped: libmpassit_ncio was built without HDF5")
        else:
            nml4 = open(os.path.join(d, "namelist.input")).read().replace("out_fortran.nc", "out_fortran_nc4.nc")
            open(os.path.join(d, "namelist.nc4"), "w").write(nml4)
            t0 = time.perf_counter()
            r = subprocess.run([exe, "namelist.nc4"], cwd=d, capture_output=True, text=True, timeout=900, env=dict(os.environ, MPASSIT_OUTPUT_FORMAT="netcdf4"))
            t4 = time.perf_counter() - t0
            if r.returncode != 0:
                print(r.stdout[-3000:], r.stderr[-3000:])
                return 1
            p4 = os.path.join(d, "out_fortran_nc4.nc")
            print("fortran, NetCDF-4 out: %.2f s wall (%.2f GB)   %s" % (t4, os.path.getsize(p4) / 1e9, "  ".join(
                ln.strip() for ln in r.stdout.splitlines() if ln.lstrip().startswith("["))), flush=True)
            worst, nvar = 0, 0
            with ncio.Reader(p4) as a, ncio.Reader(os.path.join(d, "out_fortran.nc")) as b:
                same_hdr = list(a.vars) == list(b.vars) and a.dims == b.dims and list(a.atts()) == list(b.atts())
                for name in a.vars:
                    nvar += 1
                    worst += int(a.get(name).tobytes() != b.get(name).tobytes())
            print("    NetCDF-4 output vs CDF-5 output: same dimensions / variables / attributes: %s; %d variables compared, %d differ in any bit" % (
                same_hdr, nvar, worst), flush=True)
            os.remove(p4)
            if worst or not same_hdr:
                return 1
            # NetCDF-4 INPUTS (through libhdf5, host arrays), classic output: the device flow's file, byte for byte
            nml_in = open(os.path.join(d, "namelist.input")).read().replace("%s/init.nc" % d, "%s/nc4/init.nc" % d).replace(
                "%s/hist.nc" % d, "%s/nc4/hist.nc" % d).replace("%s/diag.nc" % d, "%s/nc4/diag.nc" % d).replace("out_fortran.nc", "out_fortran_nc4in.nc")
            open(os.path.join(d, "namelist.nc4in"), "w").write(nml_in)
            t0 = time.perf_counter()
            r = subprocess.run([exe, "namelist.nc4in"], cwd=d, capture_output=True, text=True, timeout=900)
            t4 = time.perf_counter() - t0
            if r.returncode != 0:
                print(r.stdout[-3000:], r.stderr[-3000:])
                return 1
            same_in = open(os.path.join(d, "out_fortran_nc4in.nc"), "rb").read() == open(os.path.join(d, "out_fortran.nc"), "rb").read()
            print("fortran, NetCDF-4 in:  %.2f s wall   %s" % (t4, "  ".join(ln.strip() for ln in r.stdout.splitlines() if ln.lstrip().startswith("["))), flush=True)
            print("    output identical to the classic inputs' byte for byte: %s" % same_in, flush=True)
            os.remove(os.path.join(d, "out_fortran_nc4in.nc"))
            import shutil
            shutil.rmtree(os.path.join(d, "nc4"))
            if not same_in:
                return 1

    nr = int(os.environ.get("C4JOB_RANKS", "0"))      # C4JOB_RANKS=N: the same job as N driver images (row blocks of the target grid)
    if nr > 1:
        import mpassit_ranks
        nml_txt = open(os.path.join(d, "namelist.input")).read().replace("out_fortran.nc", "out_fortran_r%d.nc" % nr)
        open(os.path.join(d, "namelist.ranks"), "w").write(nml_txt)
        t0 = time.perf_counter()
        res = mpassit_ranks.launch("namelist.ranks", nr, gpus=int(os.environ.get("C4JOB_GPUS", "1")), exe=exe, cwd=d, timeout=900)
        t_r = time.perf_counter() - t0
        bad = [(i, r[0]) for i, r in enumerate(res) if r[0] != 0]
        if bad:
            print("multi-image run failed:", bad, res[bad[0][0]][1][-2000:], res[bad[0][0]][2][-2000:])
            return 1
        same_r = open(os.path.join(d, "out_fortran.nc"), "rb").read() == open(os.path.join(d, "out_fortran_r%d.nc" % nr), "rb").read()
        print("fortran, %d images on %s GPU(s): %.2f s wall; output identical to the single image's byte for byte: %s" % (
            nr, os.environ.get("C4JOB_GPUS", "1"), t_r, same_r), flush=True)
        for i, r in enumerate(res):
            print("    image %d: %s" % (i, "  ".join(ln.strip() for ln in r[1].splitlines() if ln.lstrip().startswith("["))), flush=True)
        os.remove(os.path.join(d, "out_fortran_r%d.nc" % nr))
        if not same_r:
            return 1

    cfg = I.InterpConfig(wrf_mod_vars=True, proj_is_lambert=not c5, diag_list=diag_list, hist_2d=HIST_2D, hist_3d=HIST_3D, hist_soil=SOIL)
    nml = dict(dx=3000.0, ref_lat=38.5, ref_lon=-97.5, truelat1=38.5, truelat2=38.5, stand_lon=-97.5)
    if c5:
        nml = dict(dx=6370000.0 * 2 * np.pi / 3600, stand_lon=0.0)
    for mode, out in ((False, "out_host.nc"), (True, "out_dev.nc"), (True, "out_dev.nc")):
        tm = {}
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = io_nc.run(os.path.join(d, "init.nc"), os.path.join(d, "hist.nc"), os.path.join(d, out), g, cfg,
                        diag_path=os.path.join(d, "diag.nc"), namelist=nml, fmt=5, device_io=mode, timings=tm)
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
        n3 = sum(1 for v in res.values() if getattr(v, "ndim", 0) == 3)
        print("%-22s %.2f s wall (%d 3-D + %d 2-D output variables)   %s" % ("python, device i/o:" if mode else "python, host arrays:", t, n3,
              len(res) - n3, "  ".join("%s %.2f" % kv for kv in tm.items())), flush=True)
        del res
        torch.cuda.empty_cache()
    same = open(os.path.join(d, "out_host.nc"), "rb").read() == open(os.path.join(d, "out_dev.nc"), "rb").read()
    import hashlib
    hsh = hashlib.sha256()
    with open(os.path.join(d, "out_fortran.nc"), "rb") as fh:
        for blk in iter(lambda: fh.read(1 << 24), b""):
            hsh.update(blk)
    print("sha256 of the fortran (device flow) output: %s" % hsh.hexdigest())
    print("fortran outputs (host arrays / device flow) identical byte for byte:",
          open(os.path.join(d, "out_fortran_host.nc"), "rb").read() == open(os.path.join(d, "out_fortran.nc"), "rb").read())
    print("python outputs identical byte for byte:", same)
    worst = 0.0
    with ncio.Reader(os.path.join(d, "out_fortran.nc")) as a, ncio.Reader(os.path.join(d, "out_dev.nc")) as b:
        names = [n for n in b.vars if n in a.vars and b.vars[n]["type"] == ncio.FLOAT]
        missing = [n for n in b.vars if n not in a.vars]
        for n in names:
            x, y = a.get(n, rec=0), b.get(n, rec=0)
            worst = max(worst, float(np.abs(x - y).max() / max(1.0, float(np.abs(y).max()))))
    if missing:
        print("not in the fortran output:", missing)
    print("fortran vs python output: %d variables compared, %d missing, worst |diff| / max(1, |field|max) = %.2e" % (len(names), len(missing), worst))
    if not os.environ.get("C4JOB_KEEP"):             # C4JOB_KEEP=1: leave inputs, lists and namelist for a profiler run of the driver
        for f in os.listdir(d):
            os.remove(os.path.join(d, f))
        os.rmdir(d)
    return 0 if same and not missing and worst < 1e-6 else 1


if __name__ == "__main__":
    sys.exit(main())
