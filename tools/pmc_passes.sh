#!/bin/bash
# What binds a kernel?  SQ / TA / TCP / TCC counters of every kernel a command launches, in separate rocprofv3 --pmc passes
# (8 SQ slots, 4 TCC slots per pass: MI355X_MICROARCH.md "rocprofv3 PMC slots"; --pmc is never combined with a trace), plus
# one --kernel-trace pass for durations, registers, LDS and scratch.  Summarised per kernel by tools/pmc_summary.py into
# gpurun_out/pmc_<tag>/summary.{md,json}.   usage: tools/pmc_passes.sh <tag> <python script> [args...]
set -e
REPO=${GRAFT_REPO_ROOT:-$PWD}
TAG=$1; shift
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
[ -f "$REPO/$1" ] && { S="$REPO/$1"; shift; set -- "$S" "$@"; }   # the passes run from /tmp: a script path relative to the repo is made absolute
cd /tmp && export TMPDIR=/tmp
CMD="$*"
timeout -k 10 ${PMC_TIMEOUT:-300} rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $CMD > $OUT/trace.log 2>&1
# every pass under its own timeout; a pass that is killed ends the script (set -e): no further GPU step after a hang
p() { n=$1; shift; timeout -k 10 ${PMC_TIMEOUT:-300} rocprofv3 --pmc "$@" --output-format csv -d $OUT/$n -o c -- python3 $CMD > $OUT/$n.log 2>&1; }
p sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
p sq2 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU
p sq3 SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_ACTIVE_INST_SCA SQ_INSTS_FLAT SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_THREAD_CYCLES_VALU
# TCC only: a TA_* / TCP_* pass hung rocprofv3 on this pool in round 2 (tools/README.md) and is not collected
p mem TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum
python3 $REPO/tools/pmc_summary.py $OUT $TAG
# keep the digest and the kernel stats; the raw traces (tens of MB) stay behind: gpurun_out/ comes back only below 64 MiB
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv 2>/dev/null || true
rm -rf $OUT/trace $OUT/sq1 $OUT/sq2 $OUT/sq3 $OUT/mem
