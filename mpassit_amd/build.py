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
SOURCES = ["mpg_api.hip", "k_setup.hip", "k_store_bilinear.hip", "k_store_nearest.hip", "k_store_conserve.hip",
           "k_store_gridbil.hip", "k_apply.hip", "k_halo.hip"]
HEADERS = ["mpg_internal.h", "geom.h", os.path.join("..", "..", "include", "mpassit_amd.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
         "-fgpu-rdc" if False else "-fno-gpu-rdc"]


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
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO] + objs)
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
