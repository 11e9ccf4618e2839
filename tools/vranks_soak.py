#!/usr/bin/env python3
"""One-off soak of the C-ABI's multi-GPU verbs through virtual ranks (mpg_comm_virtual: the real RCCL group calls, to self, on one GPU):
random meshes (global Voronoi / variable resolution / icosahedral in Morton or native order / regional hex, now and then with shuffled
numbering), random target grids (the generators of tests/test_fuzz_gpu.py), 2..8 virtual ranks, every halo form -- the library's
partitions (aligned, para_range: range or compact form as the numbering decides) and the caller's (owned: by need, or a random
assignment) --, float32 / float64, cell-fast / file-order slabs.  Per case: every rank's slab holds the field's bytes at every id its
rows reference, its Regrid equals the single-GPU Regrid BIT FOR BIT, the gathered field too, the schedules equal the pure plan's.
usage (GPU box): python tools/vranks_soak.py [--cases 150] [--seed 0]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=150)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    os.environ.setdefault("MPG_COMM_TIMEOUT_S", "60")
    import torch
    from mpassit_amd import _lib, synth
    import test_fuzz_gpu as F
    import test_vranks_gpu as T
    _lib.init(0)
    bad, forms = 0, {}
    t0 = time.time()
    for case in range(args.cases):
        rng = np.random.default_rng(700000 + 1000 * args.seed + case)
        m = F._mesh(rng, case % 4)
        if rng.random() < 0.3:
            m = synth.shuffle_cells(m, seed=int(rng.integers(1 << 30)))
        g = None
        for _ in range(20):
            try:
                g = F._grid(rng, kind=int(rng.integers(5)))
            except Exception:        # a parameter combination the namelist checks refuse
                continue
            if g.ny >= 8:
                break
        V = int(rng.integers(2, min(8, g.ny) + 1))
        dt = [torch.float32, torch.float64][int(rng.integers(2))]
        lev_fast = bool(rng.integers(2))
        form = ["aligned", "para_range", "need", "random"][int(rng.integers(4))]
        what = "case %d: mesh kind %d (%d cells), grid code %d %dx%d, V %d, %s, %s, %s" % (case, case % 4, m.nCells, g.proj.code, g.nx, g.ny, V, form,
                                                                                           str(dt).split(".")[1], "file order" if lev_fast else "cell-fast")
        try:
            if form in ("need", "random"):
                T._rehearse_owned(m, g, V, form, dt, lev_fast, strict=False)
            else:
                T._rehearse(m, g, V, None, dt, lev_fast, own_streams=bool(rng.integers(2)), ownership=form)
            forms[form] = forms.get(form, 0) + 1
        except Exception as e:     # noqa: BLE001 -- a soak reports and goes on
            bad += 1
            print("FAIL %s: %s: %s" % (what, type(e).__name__, str(e)[:300]), flush=True)
            if isinstance(e, (TimeoutError, _lib.MpgError)):
                print("# a rendezvous failed: the virtual group is in an undefined state, stopping", flush=True)
                break
        if case % 10 == 9:
            print("# %d cases, %d failures, %.0f s; by form %s" % (case + 1, bad, time.time() - t0, forms), flush=True)
    print("# done: %d cases, %d failures; by form %s" % (args.cases, bad, forms))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
