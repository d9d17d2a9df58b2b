# Round-3 artefacts on one GPU box: driver-style bench line, the same command under rocprofv3 (kernel stats), drop-in kernel stats.
set -e
mkdir -p gpurun_out
R=$PWD
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03_bench_prof -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-psnr --no-cpu-baseline --no-dual --no-dropin --no-inference --no-fp32 > $R/gpurun_out/r03_bench_under_rocprof.json 2> $R/gpurun_out/r03_bench_prof.err
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03_dropin_prof -- python3 $R/tools/dbg/dropin_prof.py 20 > $R/gpurun_out/r03_dropin_prof.log 2>&1
cd $R
python3 tools/show_stats.py gpurun_out/r03_bench_prof 14 > gpurun_out/r03_bench_stats.txt
python3 tools/show_stats.py gpurun_out/r03_dropin_prof 14 > gpurun_out/r03_dropin_stats.txt
cat gpurun_out/r03_bench_stats.txt
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','dtype')}, d['roofline']['frac'], d['roofline']['ms_per_launch'])
print('dropin', d['dropin_train_step']['ms_per_step'], d['dropin_train_step']['ratio_to_engine'])
print('step roofline', {k:v['frac'] for k,v in d['roofline_step'].items() if isinstance(v,dict)})
print('psnr', d['psnr_parity']['abs_delta_db_by_arithmetic'], 'dual', d['dual_branch_ms_per_step'])
PY
