set -e
python bench.py --gpus 1 --steps 50 --warmup 5 --no-psnr --no-cpu-baseline --no-dropin --no-inference --no-fp32 > gpurun_out/r3_b24.json 2> gpurun_out/r3_b24.err || { tail -20 gpurun_out/r3_b24.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3_b24.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])
print(json.dumps(d['dual_branch'], indent=0)[:1800])
print(json.dumps(d['roofline_scene'], indent=0)[:900])
PY
