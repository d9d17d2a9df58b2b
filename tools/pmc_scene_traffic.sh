#!/bin/bash
# usage (GPU box): tools/pmc_scene_traffic.sh tag  - HBM traffic of the scene-branch kernels of one optimisation step (1023 x 128 samples): two
# separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; --kernel-trace only); FETCH_SIZE doubled (gfx950, MI355X_MICROARCH.md)
R=$PWD
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmcs_$1_$c -- python3 $R/tools/bench_scene.py 1023 128 3 > $R/gpurun_out/pmcs_$1_$c.log 2>&1 || exit 1
done
cd $R
python3 - <<PY
import csv, glob, json, collections
out = collections.defaultdict(dict)
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    f = glob.glob('gpurun_out/pmcs_$1_%s/*/*counter_collection.csv' % c)[0]
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != c:
            continue
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')[:44]
        per[k][r['Dispatch_Id']] += float(r['Counter_Value'])
    for k, d in per.items():
        out[k][c + '_KB'] = sum(d.values()) / len(d)
        out[k]['launches'] = len(d)
M = 1023 * 128
# algorithmic bytes per sample: forward chain = 256 B encoded points in, 8 x 1 KB activations + 8 x 32 B masks + 8 B density out; backward chain =
# 512 B d(hidden) + 8 x 32 B masks + 4 B d raw in, 1152 B d(layer 7) (288-float rows, 256 written) + 7 x 1 KB out; a weight gradient = 2 KB in
alg = {'k_nerf_trunk<false': 256 + 8 * 1024 + 8 * 32 + 8, 'k_nerf_trunk<true': 512 + 8 * 32 + 4 + 8 * 1024, 'k_gemm_tn_split': 2048}
res = {}
for k, v in sorted(out.items(), key=lambda kv: -(2 * kv[1].get('FETCH_SIZE_KB', 0) + kv[1].get('WRITE_SIZE_KB', 0)) * kv[1].get('launches', 1))[:12]:
    a = [alg[x] for x in alg if k.startswith(x)]
    res[k] = dict(v, read_MB=2 * v.get('FETCH_SIZE_KB', 0) / 1024, write_MB=v.get('WRITE_SIZE_KB', 0) / 1024,
                  algorithmic_MB=(a[0] * M / 1e6 if a else None))
    print(k, {kk: (round(vv, 1) if isinstance(vv, float) else vv) for kk, vv in res[k].items() if kk in ('read_MB', 'write_MB', 'algorithmic_MB', 'launches')})
json.dump(res, open('gpurun_out/pmcs_$1.json', 'w'), indent=1)
PY
