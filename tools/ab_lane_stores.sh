#!/bin/bash
# per-lane streaming stores (geom.h stream_store_lane / buf_store_lane: the product) against every lane non-temporal (-DMPG_STREAM_STORE_MODE=2,
# mpassit_amd/_alt/libst_nt.so from build.build_alt: what rounds 2-6a shipped), alternating processes, two rounds -- for the kernels that
# have no in-process knob (wind chain, configuration 5's staged kernel).  usage (GPU box): tools/ab_lane_stores.sh
cd "$(dirname "$0")/.."
for rep in 1 2; do for lib in product st_nt; do
  if [ $lib = product ]; then unset MPASSIT_AMD_LIB; else export MPASSIT_AMD_LIB=$PWD/mpassit_amd/_alt/lib$lib.so; fi
  echo "== $lib rep $rep: wind chain 1800 x 1060 x 55 float64 / float32 big-endian; 1799 x 1059"
  python tools/wind_chain_probe.py 2>/dev/null | grep -o '"dst": "[a-z0-9]*"\|"fused_ms": [0-9.]*\|"bits_equal": [a-z]*' | paste - - -
  python tools/wind_chain_probe.py --f32 2>/dev/null | grep -o '"dst": "[a-z0-9]*"\|"fused_ms": [0-9.]*\|"bits_equal": [a-z]*' | paste - - -
  python tools/wind_chain_probe.py --nx 1799 --ny 1059 2>/dev/null | grep -o '"dst": "[a-z0-9]*"\|"fused_ms": [0-9.]*\|"bits_equal": [a-z]*' | paste - - -
  python tools/wind_chain_probe.py --nx 1799 --ny 1059 --f32 2>/dev/null | grep -o '"dst": "[a-z0-9]*"\|"fused_ms": [0-9.]*\|"bits_equal": [a-z]*' | paste - - -
  echo "== $lib rep $rep: configuration 5 float32 file order (k_apply3_lfu), 1800 rows and row blocks"
  python tools/row_block_probe.py --workload c5_global_latlon --rows 225,226,1800 --io f32 --layout lev_fast --fields 4 2>&1 >/dev/null | grep '^# ' | cut -c1-60,118-200
done; done
