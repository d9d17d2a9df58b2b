set -e
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/test_hip_step.py tests/test_hip_configs.py tests/test_hip_fullsize.py -x -q -m gpu -k "fused_step or config_step or compaction or ref96" > gpurun_out/r3_t12.log 2>&1 || { tail -40 gpurun_out/r3_t12.log | cut -c1-300; exit 1; }
tail -3 gpurun_out/r3_t12.log
F="--gpus 1 --steps 100 --warmup 10 --no-psnr --no-cpu-baseline --no-dual --no-dropin --no-inference --no-fp32"
for v in 0 1 0 1 0 1; do
  PP_RAY_SETUP=$v python bench.py $F 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ray_setup', $v, 'ms/step', round(d['ms_per_step'],4), 'rays/s', round(d['value']), 'grid us', round(d['roofline_grid']['ms_per_launch']*1e3,1), 'rest', round(d['ms_per_step']*1e3-d['roofline_grid']['ms_per_launch']*1e3,1))
"
done
