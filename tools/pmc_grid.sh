#!/bin/bash
# usage (GPU box): tools/pmc_grid.sh tag   - HBM traffic of k_grid_tv_adam: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE)
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /root/repo/gpurun_out/pmc_$1_$c -- python3 /root/repo/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-psnr --no-dual --no-dropin --no-inference --no-fp32 > /root/repo/gpurun_out/pmc_$1_$c.log 2>&1 || exit 1
done
python3 - <<PY
import csv, glob, json
out = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    f = glob.glob('/root/repo/gpurun_out/pmc_$1_%s/*/*counter_collection.csv' % c)[0]
    vals = [float(r['Counter_Value']) for r in csv.DictReader(open(f)) if r['Kernel_Name'].startswith('k_grid_tv_adam') and r['Counter_Name'] == c]
    # one row per XCD/instance and dispatch: sum per dispatch
    rows = [r for r in csv.DictReader(open(f)) if r['Kernel_Name'].startswith('k_grid_tv_adam') and r['Counter_Name'] == c]
    per = {}
    for r in rows:
        per[r['Dispatch_Id']] = per.get(r['Dispatch_Id'], 0.0) + float(r['Counter_Value'])
    out[c] = sum(per.values()) / len(per)
    print(c, 'KB per launch', out[c], 'launches', len(per))
print(json.dumps({'FETCH_SIZE_KB': out['FETCH_SIZE'], 'WRITE_SIZE_KB': out['WRITE_SIZE'],
                  'hbm_bytes_per_launch': (2 * out['FETCH_SIZE'] + out['WRITE_SIZE']) * 1024}))
PY
