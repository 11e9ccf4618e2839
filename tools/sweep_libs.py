#!/usr/bin/env python3
"""A/B runs of experiment BUILDS (mpassit_amd.build.build_alt -> mpassit_amd/_alt/lib<name>.so): tools/sweep_lf.py as a child
process per (library, knob setting), one after the other.  Each line of the plan: <lib name or 'main'> <sweep_lf.py args...>
usage: python tools/sweep_libs.py plan.txt"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    for ln in open(sys.argv[1]):
        ln = ln.strip()
        if not ln or ln.startswith("#"):
            continue
        lib, *args = ln.split()
        env = dict(os.environ)
        if lib != "main":
            env["MPASSIT_AMD_LIB"] = os.path.join(ROOT, "mpassit_amd", "_alt", "lib%s.so" % lib)
        print("## %s :: %s" % (lib, " ".join(args)), flush=True)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sweep_lf.py")] + args, env=env, capture_output=True, text=True, timeout=600)
        for out in r.stdout.splitlines():
            if out.startswith("{"):
                import json
                d = json.loads(out)
                print("   variant %3d  ms_med %.3f  frac %.3f  fields/s %.0f  %s" % (d["variant"], d["ms_med"], d["frac_of_8TBs"], d["fields_per_s"], d.get("kernel_choice")), flush=True)
            elif out.startswith("#"):
                print("   " + out, flush=True)
        if r.returncode != 0:
            print("   FAILED rc %d: %s" % (r.returncode, r.stderr[-1500:]), flush=True)
            return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
