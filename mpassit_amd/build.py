"""Build recipe for libmpassit_amd.so (hand-written HIP for gfx950, explicit hipcc, in-tree output).

`python -m mpassit_amd.build` or `mpassit_amd.build.build()`.  hipcc cross-compiles without a GPU.
The shared object lands next to this file so that it travels with the repo snapshot to the GPU box.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJDIR = os.path.join(HERE, "_build")
SO = os.path.join(HERE, "libmpassit_amd.so")
SOURCES = ["mpg_api.hip", "mpg_comm.hip", "mpg_hostpipe.hip", "mpg_fileio.hip", "k_setup.hip", "k_mesh_window.hip", "k_target_grid.hip", "k_store_bilinear.hip", "k_store_nearest.hip", "k_store_conserve.hip",
           "k_store_gridbil.hip", "k_apply.hip", "k_apply_lfu.hip", "k_apply_typed.hip", "k_wind.hip", "k_pole.hip", "k_post.hip", "k_halo.hip", "k_prims.hip", "k_sort.hip"]
HEADERS = ["mpg_internal.h", "geom.h", os.path.join("..", "..", "include", "mpassit_amd.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# (Round 5 carried an MPASSIT_STRIP_DEVICE switch here whose one documented setting, `-Xoffload-linker --strip-all`, drops the device objects'
# .symtab and makes the first hipLaunchKernel of this runtime SEGFAULT after a successful mpg_init -- profiles/r05_init_breakdown.md,
# gpurun_out/r05/strip_probe.txt.  A build option whose only known value crashes a process that holds the GPU is not an option: removed
# in round 6; the 1.2 MB it saved are not worth a loader-dependent failure mode.)
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function", "-fno-gpu-rdc"]


def _newer(a, deps):
    if not os.path.exists(a):
        return False
    t = os.path.getmtime(a)
    return all(os.path.getmtime(d) <= t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(OBJDIR, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    jobs = []
    objs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJDIR, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or not _newer(obj, [src] + hdrs):
            jobs.append([HIPCC] + FLAGS + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stderr[-8000:]))
        return r.stderr

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            for warn in ex.map(run, jobs):
                if verbose and warn.strip():
                    print(warn)
    if jobs or force or not _newer(SO, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO] + objs + ["-ldl"])
    return SO


def build_alt(name, defines, sources=("k_apply_typed.hip", "k_apply_lfu.hip", "k_apply.hip")):
    """A/B build of an experiment: the listed sources recompiled with extra -D flags, the rest of the objects taken from the
    product build, linked into mpassit_amd/_alt/lib<name>.so (git-ignored; travels to the GPU box).  Selected at run time
    with MPASSIT_AMD_LIB=<path> (_lib.py).  Never the shipped library."""
    build()
    alt = os.path.join(HERE, "_alt")
    od = os.path.join(alt, "obj_" + name)
    os.makedirs(od, exist_ok=True)
    objs, jobs = [], []
    for s in SOURCES:
        if s in sources:
            o = os.path.join(od, s.replace(".hip", ".o"))
            jobs.append([HIPCC] + FLAGS + ["-D" + d for d in defines] + ["-c", os.path.join(CSRC, s), "-o", o])
        else:
            o = os.path.join(OBJDIR, s.replace(".hip", ".o"))
        objs.append(o)

    def run(cmd):
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stderr[-8000:]))
    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    so = os.path.join(alt, "lib%s.so" % name)
    run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so] + objs + ["-ldl"])
    return so


NCIO_SRC = os.path.join(HERE, "hostio", "ncclassic.c")
NCIO_SO = os.path.join(HERE, "hostio", "libmpassit_ncio.so")


def find_hdf5():
    """(include_dir, lib_dir) of an HDF5 C library with its high-level library (hdf5.h, libhdf5.so, libhdf5_hl.so), or None.
    MPASSIT_HDF5_ROOT names a prefix (empty string / "none" = build without); otherwise the usual prefixes -- the image has 1.10.6 under
    /opt/conda.  NetCDF-4 is an HDF5 container: with the library the ncio interface reads and writes it too."""
    root = os.environ.get("MPASSIT_HDF5_ROOT")
    if root is not None and root.strip().lower() in ("", "none", "0"):
        return None
    cands = [root] if root else ["/usr", "/usr/local", "/opt/conda"]
    for c in cands:
        for inc, lib in ((os.path.join(c, "include"), os.path.join(c, "lib")),
                         (os.path.join(c, "include", "hdf5", "serial"), os.path.join(c, "lib", "x86_64-linux-gnu", "hdf5", "serial"))):
            if all(os.path.exists(x) for x in (os.path.join(inc, "hdf5.h"), os.path.join(inc, "hdf5_hl.h"), os.path.join(lib, "libhdf5.so"),
                                               os.path.join(lib, "libhdf5_hl.so"))):
                return inc, lib
    return None


def build_ncio(force=False):
    """Host-side NetCDF I/O (plain C, gcc): mpassit_amd/hostio/libmpassit_ncio.so -- the classic formats always, NetCDF-4 through libhdf5
    where find_hdf5() finds one."""
    hdr = os.path.join(HERE, "..", "include", "mpassit_ncio.h")
    h5 = find_hdf5()
    stamp = NCIO_SO + ".cfg"
    cfg = "hdf5=%s" % (h5,)
    same = os.path.exists(stamp) and open(stamp).read() == cfg
    if force or not same or not _newer(NCIO_SO, [NCIO_SRC, hdr, os.path.join(os.path.dirname(NCIO_SRC), "nc4hdf5.h")]):
        cmd = ["gcc", "-O2", "-Wall", "-shared", "-fPIC", "-pthread", "-o", NCIO_SO, NCIO_SRC]
        if h5:
            # the two libraries by full path, run path = their directory only for THEIR lookup (DT_RUNPATH is not inherited by dependencies)
            cmd += ["-DMPASSIT_HAVE_HDF5", "-I" + h5[0], os.path.join(h5[1], "libhdf5_hl.so"), os.path.join(h5[1], "libhdf5.so"),
                    "-Wl,--enable-new-dtags,-rpath," + h5[1]]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("gcc failed:\n%s" % r.stderr[-4000:])
        with open(stamp, "w") as f:
            f.write(cfg)
    return NCIO_SO


FORTRAN_SRC = ["mpg_mod.F90", "ncio_mod.F90", "host_mod.F90", "interp_mod.F90", "ncfiles_mod.F90", "mpassit_driver.F90"]
FLANG = os.environ.get("FLANG", "/opt/rocm/bin/amdflang")
DRIVER = os.path.join(HERE, "fortran", "mpassit")


def build_fortran(force=False):
    """Fortran driver (reference surface: namelist + parm lists) linked against the C-ABI library."""
    fdir = os.path.join(HERE, "fortran")
    srcs = [os.path.join(fdir, f) for f in FORTRAN_SRC]
    build_ncio()
    if not force and _newer(DRIVER, srcs + [SO, NCIO_SO]):
        return DRIVER
    mod = os.path.join(OBJDIR, "fmod")
    os.makedirs(mod, exist_ok=True)
    objs = []
    for s in srcs:
        o = os.path.join(mod, os.path.basename(s).replace(".F90", ".o"))
        r = subprocess.run([FLANG, "-O2", "-fPIC", "-module-dir", mod, "-I", mod, "-c", s, "-o", o], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("amdflang failed on %s:\n%s" % (s, r.stderr[-6000:]))
        objs.append(o)
    r = subprocess.run([FLANG, "-o", DRIVER] + objs + ["-L" + HERE, "-lmpassit_amd", "-L" + os.path.join(HERE, "hostio"), "-lmpassit_ncio", "-Wl,-rpath," + HERE,
                        "-Wl,-rpath," + os.path.join(HERE, "hostio"), "-Wl,-rpath,/opt/rocm/lib"],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("linking the Fortran driver failed:\n%s" % r.stderr[-6000:])
    return DRIVER


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_fortran(force="--force" in sys.argv))
