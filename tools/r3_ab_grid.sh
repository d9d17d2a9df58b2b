set -e
mkdir -p gpurun_out
F="--gpus 1 --steps 60 --warmup 5 --no-psnr --no-cpu-baseline --no-dual --no-dropin --no-inference --no-fp32"
for skew in 0 12288 4096 16384 53248 0 12288; do
  PP_GRID_SKEW=$skew python bench.py $F 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('skew', $skew, 'ms/step', round(d['ms_per_step'],4), 'rays/s', round(d['value']), 'grid us', round(d['roofline_grid']['ms_per_launch']*1e3,1), 'frac', round(d['roofline_grid']['frac'],3))
"
done
