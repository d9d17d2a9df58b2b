#!/bin/bash
# usage (GPU box): tools/pmc_mlp_traffic.sh tag  - HBM traffic of the six object-branch MLP kernels inside the train step: two separate rocprofv3
# --pmc passes (FETCH_SIZE, WRITE_SIZE; --kernel-trace only) over a short bench.py run; FETCH_SIZE doubled (gfx950, MI355X_MICROARCH.md)
R=$PWD
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmcm_$1_$c -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-psnr --no-dual --no-dropin --no-inference --no-fp32 > $R/gpurun_out/pmcm_$1_$c.log 2>&1 || exit 1
done
cd $R
python3 - <<PY
import csv, glob, json, collections
out = collections.defaultdict(dict)
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    f = glob.glob('gpurun_out/pmcm_$1_%s/*/*counter_collection.csv' % c)[0]
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != c:
            continue
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')[:40]
        per[k][r['Dispatch_Id']] += float(r['Counter_Value'])
    for k, d in per.items():
        out[k][c + '_KB'] = sum(d.values()) / len(d)
        out[k]['launches'] = len(d)
M = 54613
alg = {'k_warp_fused_fwd_s': 12 + 4 * 2048 + 64, 'k_warp_fused_bwd_s': 2048 + 3 * 512 + 64 + 12 + 3 * 2048 + 12, 'k_wgrad_chain_s<128, 4>': 3 * 4096,
       'k_rgb_fused_fwd_s': 256 + 3 * 512 + 12, 'k_rgb_fused_bwd_s': 3 * 512 + 12 + 12 + 3 * 512 + 256, 'k_wgrad_chain_s<64, 1>': 768 + 2 * 1024}
res = {}
for k, v in out.items():
    if not any(k.startswith(a) for a in alg) and not k.startswith('k_grid') and not k.startswith('k_color') and not k.startswith('k_k0'):
        continue
    hbm = (2 * v.get('FETCH_SIZE_KB', 0) + v.get('WRITE_SIZE_KB', 0)) * 1024
    a = [alg[x] for x in alg if k.startswith(x)]
    res[k] = dict(v, hbm_bytes_per_launch=hbm, read_MB=2 * v.get('FETCH_SIZE_KB', 0) / 1024, write_MB=v.get('WRITE_SIZE_KB', 0) / 1024,
                  algorithmic_MB=(a[0] * M / 1e6 if a else None))
    print(k, {kk: (round(vv, 1) if isinstance(vv, float) else vv) for kk, vv in res[k].items() if kk in ('read_MB', 'write_MB', 'algorithmic_MB', 'launches')})
json.dump(res, open('gpurun_out/pmcm_$1.json', 'w'), indent=1)
PY
