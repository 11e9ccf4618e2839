#!/bin/bash
# Collects the rocprofv3 evidence for bench.py on the GPU box (run through gpurun from the repo root):
#   1. --kernel-trace --stats  (per-kernel durations)            -> gpurun_out/prof/stats
#   2. --pmc FETCH_SIZE        (own pass, TCC has 4 slots)       -> gpurun_out/prof/fetch
#   3. --pmc WRITE_SIZE        (own pass)                        -> gpurun_out/prof/write
#   4. --pmc raw TCC request counters                            -> gpurun_out/prof/tcc
# bench.py --calib adds a known-byte-count streaming kernel (k_pack, 1 GiB read + 1 GiB write, 8 B/lane)
# that calibrates the counters for this access width (MI355X_MICROARCH.md "HBM": FETCH_SIZE may read 1/2).
set -e
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --calib ${BENCH_EXTRA}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 $ARGS > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o fetch -- python3 $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o write -- python3 $ARGS > $OUT/write.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/tcc -o tcc -- python3 $ARGS > $OUT/tcc.log 2>&1
find $OUT -name "*.csv" | head -20
