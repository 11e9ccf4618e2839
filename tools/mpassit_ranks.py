#!/usr/bin/env python3
"""Start the Fortran driver as N images, one per GPU -- the stand-in for `mpirun -np N mpassit namelist` of the reference
(mpassit.F90:84-96; this image has no MPI for amdflang).  Image r gets MPASSIT_RANK=r, MPASSIT_NRANKS=N, a common
MPASSIT_RUN_ID and MPASSIT_DEVICE = r modulo --gpus; it regrids its block of target rows (every image reads the input files
whole, like the reference's ranks) and writes them into the one output file.  Exit code: the first non-zero one.

  python tools/mpassit_ranks.py --ranks 8 --gpus 8 namelist.input"""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def launch(namelist, ranks, gpus=1, exe=None, cwd=None, env=None, timeout=1800):
    """-> list of CompletedProcess-like (returncode, stdout, stderr) per image."""
    exe = exe or os.path.join(ROOT, "mpassit_amd", "fortran", "mpassit")
    base = dict(os.environ if env is None else env)
    base.update(MPASSIT_NRANKS=str(ranks), MPASSIT_RUN_ID=str(os.getpid()))
    procs = []
    for r in range(ranks):
        e = dict(base, MPASSIT_RANK=str(r), MPASSIT_DEVICE=str(r % max(1, gpus)))
        procs.append(subprocess.Popen([exe, namelist], cwd=cwd, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    out = []
    for p in procs:
        try:
            so, se = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        out.append((p.returncode, so, se))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("namelist")
    ap.add_argument("--ranks", type=int, required=True)
    ap.add_argument("--gpus", type=int, default=1, help="GPUs of the node the images are spread over")
    ap.add_argument("--exe", default=None)
    args = ap.parse_args()
    res = launch(args.namelist, args.ranks, args.gpus, args.exe)
    rc = 0
    for r, (code, so, se) in enumerate(res):
        sys.stdout.write("".join("[%d] %s\n" % (r, ln) for ln in so.splitlines()))
        if code:
            sys.stderr.write("[%d] exit code %d\n%s\n" % (r, code, se[-2000:]))
            rc = rc or code
    return rc


if __name__ == "__main__":
    sys.exit(main())
