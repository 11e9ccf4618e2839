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
import glob
import tempfile
import time
import uuid

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def launch(namelist, ranks, gpus=1, exe=None, cwd=None, env=None, timeout=1800):
    """-> list of CompletedProcess-like (returncode, stdout, stderr) per image."""
    exe = exe or os.path.join(ROOT, "mpassit_amd", "fortran", "mpassit")
    base = dict(os.environ if env is None else env)
    # unique per LAUNCH (a PID alone repeats; a marker a killed run left behind must never satisfy a later one)
    run_id = "%d-%s" % (os.getpid(), uuid.uuid4().hex[:16])
    base.update(MPASSIT_NRANKS=str(ranks), MPASSIT_RUN_ID=run_id)
    procs, logs = [], []
    for r in range(ranks):
        e = dict(base, MPASSIT_RANK=str(r), MPASSIT_DEVICE=str(r % max(1, gpus)))
        fo, fe = tempfile.TemporaryFile("w+"), tempfile.TemporaryFile("w+")      # files, not pipes: nothing to fill up while we poll
        logs.append((fo, fe))
        procs.append(subprocess.Popen([exe, namelist], cwd=cwd, env=e, stdout=fo, stderr=fe, text=True))
    # wait for all images; the first one that fails takes the others down (they would wait for its marker files otherwise)
    t0 = time.monotonic()
    failed = False
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs):
            failed = True
        if failed or time.monotonic() - t0 > timeout:
            for q in procs:
                if q.poll() is None:
                    q.kill()
            for q in procs:
                q.wait()
            if not failed:
                raise subprocess.TimeoutExpired([exe, namelist], timeout)
            break
        time.sleep(0.05)
    # whatever happened, no marker of this launch survives it (images killed above leave theirs behind)
    # (harmless if the output lives elsewhere: the id is never used again)
    for stale in glob.glob(os.path.join(cwd or ".", "*.%s.*" % run_id)):
        try:
            os.remove(stale)
        except OSError:
            pass
    out = []
    for p, (fo, fe) in zip(procs, logs):
        p.wait()
        fo.seek(0)
        fe.seek(0)
        out.append((p.returncode, fo.read(), fe.read()))
        fo.close()
        fe.close()
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
