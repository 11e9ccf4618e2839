set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/r03_t10.log 2>&1 || { tail -30 gpurun_out/r03_t10.log; exit 1; }
tail -3 gpurun_out/r03_t10.log
bash tools/profile_and_summarize.sh r03b || { tail -20 gpurun_out/prof_r03b.log; exit 1; }
bash tools/profile_and_summarize.sh r03b_f32_lev_fast --io f32 --layout lev_fast || { tail -20 gpurun_out/prof_r03b_f32_lev_fast.log; exit 1; }
bash tools/profile_and_summarize.sh r03b_c5_f32_lev_fast --workload c5_global_latlon --io f32 --layout lev_fast || { tail -20 gpurun_out/prof_r03b_c5_f32_lev_fast.log; exit 1; }
ls gpurun_out/sum_*/profiles
timeout -k 10 600 python bench.py > gpurun_out/r03_bench2.json 2> gpurun_out/r03_bench2.err || { tail -20 gpurun_out/r03_bench2.err; exit 1; }
python -c "
import json
d=json.loads(open('gpurun_out/r03_bench2.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], d['production_path']['roofline_frac'], d['job']['cold_ms'], d['job']['warm_ms'])"
