set -e
fmt() { grep -v "^/opt" | python -c "
import sys,json,os
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(os.environ.get('TAG'), d['workload'], d['io'], d['layout'], d['variant'], round(d['ms_med'],3), round(d['frac_of_8TBs'],3), d['kernel_choice'])
    elif 'DIFFERS' in l: print(l.strip())
"; }
for rep in 1 2; do
for TAG in a32 a16 a1; do
export TAG
if [ $TAG = a32 ]; then unset MPASSIT_AMD_LIB; else export MPASSIT_AMD_LIB=$PWD/mpassit_amd/_alt/libmpassit_amd_$TAG.so; fi
wl=c4_3m_regional
python tools/sweep_lf.py --workload $wl --io f64 --plain --layout cell_fast --fields 13 --knob tile_band --variants=0 2>&1 | fmt
python tools/sweep_lf.py --workload $wl --io f32 --layout cell_fast --fields 13 --knob tile_band --variants=0 2>&1 | fmt
done
done
