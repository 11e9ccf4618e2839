set -e
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for lib in product st_nt; do
  if [ $lib = product ]; then unset MPASSIT_AMD_LIB; else export MPASSIT_AMD_LIB=$PWD/mpassit_amd/_alt/lib$lib.so; fi
  echo "== $lib rep $rep: float64 cell-fast row blocks"
  python tools/row_block_probe.py --rows 132,133,265,1060 --io f64 --layout cell_fast 2>&1 >/dev/null | grep '^# ' | cut -c1-60,118-200
  echo "== $lib rep $rep: 132 rows, result shifted into a line"
  python tools/row_block_probe.py --rows 132 --io f64 --layout cell_fast --dst-shift-bytes 0,8,24,64,104 2>&1 >/dev/null | grep '^# ' | cut -c1-75,118-200
done; done
