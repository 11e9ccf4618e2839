#!/bin/bash
# per-kernel durations of bench.py --leg job under the product library and under an alt library (default: libst_nt.so), side by side
# usage (GPU box): tools/ab_job_kernels.sh [alt-lib-name]
REPO=${GRAFT_REPO_ROOT:-$PWD}
ALT=${1:-st_nt}
OUT=$REPO/gpurun_out/prof_ab
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for lib in product $ALT; do
  if [ $lib = product ]; then unset MPASSIT_AMD_LIB; else export MPASSIT_AMD_LIB=$REPO/mpassit_amd/_alt/lib$lib.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$lib -o s -- python3 $REPO/bench.py --leg job > $OUT/$lib.log 2>&1
done
python3 - $OUT $ALT <<'P'
import csv, glob, sys
out, alt = sys.argv[1], sys.argv[2]
def load(lib):
    f = glob.glob("%s/%s/**/s_kernel_stats.csv" % (out, lib), recursive=True)[0]
    return {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(f))}
a, b = load("product"), load(alt)
rows = sorted(set(a) | set(b), key=lambda n: -(a.get(n, (0, 0))[1] + b.get(n, (0, 0))[1]))
print("%-90s %6s %12s %12s %7s" % ("kernel", "calls", "product us", alt + " us", "ratio"))
for n in rows[:24]:
    ca, ta = a.get(n, (0, 0)); cb, tb = b.get(n, (0, 0))
    print("%-90s %6d %12.1f %12.1f %7.3f" % (n[:90], ca, ta / 1e3, tb / 1e3, ta / tb if tb else 0))
P
