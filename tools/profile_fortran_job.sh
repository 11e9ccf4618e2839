#!/bin/bash
# rocprofv3 kernel trace of the whole configuration-4 job as the Fortran driver runs it (every Store, Regrid, rotation,
# destaggering and post-op kernel of one file-to-file run) -> gpurun_out/prof_job/.  Run through gpurun from the
# repo root; the inputs are produced by tools/config4_file_job.py (C4JOB_KEEP=1 leaves them in /dev/shm/c4job).
set -e
trap 'rm -rf /dev/shm/c4job' EXIT   # gigabytes of inputs live in memory-backed /dev/shm: never leave them behind, whatever fails
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/prof_job
mkdir -p $OUT
C4JOB_KEEP=1 python3 $REPO/tools/config4_file_job.py > $OUT/job.log 2>&1
cd /dev/shm/c4job && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- $REPO/mpassit_amd/fortran/mpassit namelist.input > $OUT/driver.log 2>&1
cp $OUT/stats/*/stats_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null || cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
rm -rf $OUT/stats     # the raw trace is large; gpurun_out/ comes back only below 64 MiB
head -40 $OUT/kernel_stats.csv | cut -c1-160
