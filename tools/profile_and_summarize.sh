#!/bin/bash
# tools/profile_bench.sh for one bench.py configuration, condensed on the GPU box itself (the raw rocprofv3 CSVs are too large
# to travel back): gpurun_out/sum_<tag>/profiles/<tag>_kernel_stats.csv, <tag>_pmc_summary.json and traffic.json.
# usage: tools/profile_and_summarize.sh <tag> [bench.py args...]
set -e
REPO=${GRAFT_REPO_ROOT:-$PWD}
TAG=$1; shift
BENCH_EXTRA="$*" bash $REPO/tools/profile_bench.sh > $REPO/gpurun_out/prof_$TAG.log 2>&1
mkdir -p $REPO/gpurun_out/sum_$TAG && cd $REPO/gpurun_out/sum_$TAG
python3 $REPO/tools/summarize_profile.py $REPO/gpurun_out/prof $TAG traffic.json > summary.log 2>&1
rm -rf $REPO/gpurun_out/prof
