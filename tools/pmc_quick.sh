#!/bin/bash
# Quick HBM-traffic check of one bench.py configuration: FETCH_SIZE and WRITE_SIZE in separate passes, per-launch
# bytes of the largest-grid kernel printed (FETCH_SIZE doubled: gfx950 tallies 128-B requests at 64 B, calibrated in
# profiles/*_pmc_summary.json).  usage: tools/pmc_quick.sh <tag> <bench.py args...>
set -e
REPO=${GRAFT_REPO_ROOT:-$PWD}
TAG=$1; shift
OUT=$REPO/gpurun_out/pmcq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline $*"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o fetch -- python3 $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o write -- python3 $ARGS > $OUT/write.log 2>&1
python3 - $OUT $TAG <<'PY'
import csv, sys, collections, glob
out, tag = sys.argv[1], sys.argv[2]
res = {}
for name, fac in (("fetch", 2.0), ("write", 1.0)):
    f = glob.glob("%s/%s/**/*counter_collection.csv" % (out, name), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    per = collections.defaultdict(list)
    for r in rows:
        g = int(r.get("Grid_Size") or r.get("Grid_Size_X"))
        per[(r["Kernel_Name"].split("(")[0], g)].append(float(r["Counter_Value"]))
    # the apply kernel = the k_apply* entry with the largest grid
    cand = [k for k in per if "k_apply" in k[0]]
    k = max(cand, key=lambda k: k[1])
    v = per[k]
    res[name] = (k[0], len(v), sum(v) / len(v) * 1024 * fac)
print("%s: %s launches=%d read %.3f GB  write %.3f GB  total %.3f GB" % (tag, res["fetch"][0][-60:], res["fetch"][1], res["fetch"][2] / 1e9, res["write"][2] / 1e9, (res["fetch"][2] + res["write"][2]) / 1e9))
PY
