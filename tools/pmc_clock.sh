#!/bin/bash
# usage (GPU box): tools/pmc_clock.sh  - effective shader clock under the scene-branch GEMMs: GRBM_GUI_ACTIVE cycles / kernel duration
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /root/repo/gpurun_out/pmc_clock -- python3 /root/repo/tools/bench_scene.py 1023 128 3 > /root/repo/gpurun_out/pmc_clock.log 2>&1 || { tail -5 /root/repo/gpurun_out/pmc_clock.log; exit 1; }
python3 - <<PY
import csv, glob, collections, json
d = glob.glob('/root/repo/gpurun_out/pmc_clock/*/')[0]
cc = list(csv.DictReader(open(glob.glob(d + '*counter_collection.csv')[0])))
kt = {r['Dispatch_Id']: r for r in csv.DictReader(open(glob.glob(d + '*kernel_trace.csv')[0]))}
per = collections.defaultdict(list)
for r in cc:
    if r['Counter_Name'] != 'GRBM_GUI_ACTIVE': continue
    t = kt.get(r['Dispatch_Id'])
    if not t: continue
    dur = float(t['End_Timestamp']) - float(t['Start_Timestamp'])
    if dur < 20000: continue
    per[r['Kernel_Name'].split('(')[0][:50]].append((float(r['Counter_Value']), dur))
out = {}
for k, v in per.items():
    cyc = sum(a for a, _ in v); ns = sum(b for _, b in v)
    out[k] = {'dispatches': len(v), 'cycles_per_ns_raw': cyc / ns}
    print(k, out[k])
json.dump(out, open('/root/repo/gpurun_out/pmc_clock.json', 'w'), indent=1)
PY
