#!/bin/bash
# A/B of the streaming-store policy (geom.h stream_nt) between BUILDS on one box: the product (per level, by the alignment of its plane),
# every plane plain (-DMPG_STREAM_STORE_MODE=1) and every plane non-temporal (=2: what rounds 2-6a shipped).  The alt libraries come from
#   python -c "from mpassit_amd import build; s=('k_apply_typed.hip','k_apply_lfu.hip','k_apply.hip','k_wind.hip'); build.build_alt('st_plain',['MPG_STREAM_STORE_MODE=1'],s); build.build_alt('st_nt',['MPG_STREAM_STORE_MODE=2'],s)"
# usage (GPU box): tools/ab_stream_store.sh > gpurun_out/r06/ab_stream_store.txt
set -e
cd "$(dirname "$0")/.."
for lib in product st_plain st_nt; do
  if [ $lib = product ]; then unset MPASSIT_AMD_LIB; else export MPASSIT_AMD_LIB=$PWD/mpassit_amd/_alt/lib$lib.so; fi
  echo "== $lib: wind chain float64 / float32 big-endian (1800 x 1060 x 55)"
  python tools/wind_chain_probe.py 2>/dev/null | cut -c1-400
  python tools/wind_chain_probe.py --f32 2>/dev/null | cut -c1-400
  echo "== $lib: float32 file order (k_apply3_lf_rows), row blocks"
  python tools/row_block_probe.py --rows 132,133,265,530,1060 --io f32 --layout lev_fast 2>&1 >/dev/null | grep '^# '
  echo "== $lib: float64 cell-fast (k_apply3_cfu), row blocks"
  python tools/row_block_probe.py --rows 132,133,265,530,1060 --io f64 --layout cell_fast 2>&1 >/dev/null | grep '^# '
  echo "== $lib: float32 cell-fast, row blocks"
  python tools/row_block_probe.py --rows 132,133,1060 --io f32 --layout cell_fast 2>&1 >/dev/null | grep '^# '
done
