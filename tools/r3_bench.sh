set -e
mkdir -p gpurun_out
start=$(date +%s)
timeout -k 10 1000 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3_bench.json 2> gpurun_out/r3_bench.err
end=$(date +%s)
echo "bench wall: $((end-start)) s"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3_bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','dtype')})
print('roofline', {k:d['roofline'][k] for k in ('kernel','frac','ms_per_launch')})
print('roofline_step', {k:(v if not isinstance(v,dict) else {kk:v[kk] for kk in ('bytes','frac')}) for k,v in d['roofline_step'].items()})
print('dropin', d['dropin_train_step'])
print('fp32', d['fp32_instructions'])
print('dual', d.get('dual_branch_ms_per_step'), d.get('roofline_scene',{}).get('frac'))
print('psnr', {k:v for k,v in d['psnr_parity'].items() if k not in ('curve','free_running','workload')})
print('cpu', d['cpu_baseline'])
for k,v in d['roofline_mlp']['kernels'].items(): print(k, round(v['ms']*1e3,1),'us', round(v['frac'],3), round(v['hbm_gbs']))
PY
