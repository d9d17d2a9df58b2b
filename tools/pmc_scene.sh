#!/bin/bash
# usage (GPU box): tools/pmc_scene.sh   - SQ counters of the scene-branch kernels (one rocprofv3 --pmc pass, SQ block only)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS \
  --kernel-trace --output-format csv -d /root/repo/gpurun_out/pmc_scene -- python3 /root/repo/tools/bench_scene.py 1023 128 3 > /root/repo/gpurun_out/pmc_scene.log 2>&1 || { tail -5 /root/repo/gpurun_out/pmc_scene.log; exit 1; }
python3 - <<PY
import csv, glob, json, collections
f = glob.glob('/root/repo/gpurun_out/pmc_scene/*/*counter_collection.csv')[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
nd = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0][:60]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    nd[k].add(r['Dispatch_Id'])
out = {}
for k, c in sorted(agg.items(), key=lambda kv: -kv[1].get('SQ_BUSY_CU_CYCLES', 0))[:8]:
    busy = c.get('SQ_BUSY_CU_CYCLES', 0) or 1
    wave = c.get('SQ_WAVE_CYCLES', 0) or 1
    out[k] = {'dispatches': len(nd[k]), 'mfma_busy_over_cu_busy': c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / busy,
              'lds_bank_conflict_over_cu_busy': c.get('SQ_LDS_BANK_CONFLICT', 0) / busy, 'wait_any_frac': c.get('SQ_WAIT_ANY', 0) / wave,
              'wait_inst_frac': c.get('SQ_WAIT_INST_ANY', 0) / wave, 'wait_inst_lds_frac': c.get('SQ_WAIT_INST_LDS', 0) / wave,
              'active_inst_frac': c.get('SQ_ACTIVE_INST_ANY', 0) / wave}
    print(k, json.dumps(out[k]))
json.dump(out, open('/root/repo/gpurun_out/pmc_scene.json', 'w'), indent=1)
PY
