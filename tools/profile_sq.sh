#!/bin/bash
# SQ / TA / TCP counters of the bench kernels (diagnosis only): gpurun_out/prof_sq/{tag}
set -e
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/prof_sq
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for tag in cf lf; do
  EXTRA=""; [ $tag = lf ] && EXTRA="--layout lev_fast"
  ARGS="$REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline $EXTRA"
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --output-format csv -d $OUT/${tag}_sq -o sq -- python3 $ARGS > $OUT/${tag}_sq.log 2>&1
  rocprofv3 --pmc TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/${tag}_ta -o ta -- python3 $ARGS > $OUT/${tag}_ta.log 2>&1
done
