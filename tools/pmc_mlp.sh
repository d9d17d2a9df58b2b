#!/bin/bash
# usage (GPU box): tools/pmc_mlp.sh [PP_MLP_SPLIT value]  - SQ counters of the object-branch MLP kernels (one rocprofv3 --pmc pass,
# SQ block only) over tools/bench_mlp.py; summary -> gpurun_out/pmc_mlp_<value>.json
V=${1:-15}
cd /tmp && export TMPDIR=/tmp
export PP_MLP_SPLIT=$V
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU \
  --kernel-trace --output-format csv -d /root/repo/gpurun_out/pmc_mlp_$V -- python3 /root/repo/tools/bench_mlp.py --iters 3 > /root/repo/gpurun_out/pmc_mlp_$V.log 2>&1 || { tail -5 /root/repo/gpurun_out/pmc_mlp_$V.log; exit 1; }
python3 - <<PY
import csv, glob, json, collections
f = glob.glob('/root/repo/gpurun_out/pmc_mlp_$V/*/*counter_collection.csv')[0]
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
    n = len(nd[k])
    out[k] = {'dispatches': n, 'mfma_busy_over_cu_busy': c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / busy,
              'lds_bank_conflict_over_cu_busy': c.get('SQ_LDS_BANK_CONFLICT', 0) / busy, 'wait_any_frac': c.get('SQ_WAIT_ANY', 0) / wave,
              'wait_inst_frac': c.get('SQ_WAIT_INST_ANY', 0) / wave, 'active_inst_frac': c.get('SQ_ACTIVE_INST_ANY', 0) / wave,
              'valu_insts_per_dispatch': c.get('SQ_INSTS_VALU', 0) / n, 'wave_cycles_per_dispatch': wave / n}
    print(k, json.dumps(out[k]))
json.dump(out, open('/root/repo/gpurun_out/pmc_mlp_$V.json', 'w'), indent=1)
PY
