#!/bin/bash
# Round-5 A/B on ONE box: the library of round 4's HEAD (staged under mpassit_amd/_alt/r04tree by the builder: git worktree of bc38b81,
# built there) against the current one -- Store kernels (rocprofv3 kernel stats of tools/store_timing.py), first calls with and
# without mpg_init's helper thread (code-object loading), then the rank-share rehearsal.  Output: gpurun_out/r05/ab_*.
set -e
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/r05
OLD=$REPO/mpassit_amd/_alt/r04tree
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for side in old new; do
  T=$REPO; [ $side = old ] && T=$OLD
  for wl in c4_3m_regional c5_global_latlon; do
    python3 $T/tools/store_timing.py --workload $wl --proj > $OUT/ab_store_${side}_$wl.txt 2>&1
  done
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_$side -o t -- python3 $T/tools/store_timing.py --workload c4_3m_regional --proj --methods conserve > $OUT/ab_trace_$side.log 2>&1
  cp $(find /tmp/ab_$side -name "*kernel_stats.csv" | head -1) $OUT/ab_conserve_kernel_stats_$side.csv
  rm -rf /tmp/ab_$side
  MPG_NO_WARMUP=1 python3 $T/tools/first_call_probe.py --reps 2 > $OUT/ab_first_call_nowarm_$side.txt 2>&1
  python3 $T/tools/first_call_probe.py --reps 2 > $OUT/ab_first_call_warm_$side.txt 2>&1
done
MPG_INIT_TRACE=1 python3 $REPO/tools/first_call_probe.py --reps 1 > $OUT/init_trace_new.txt 2>&1
cd $REPO
python3 tools/rank_share_rehearsal.py > $OUT/rank_share_f64_cell_fast.json 2> $OUT/rank_share_f64.err
python3 tools/rank_share_rehearsal.py --io f32 --layout lev_fast > $OUT/rank_share_f32_lev_fast.json 2> $OUT/rank_share_f32.err
tail -n 3 $OUT/ab_store_*_c4_3m_regional.txt
cat $OUT/rank_share_f64_cell_fast.json
