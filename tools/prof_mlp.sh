#!/bin/bash
# usage (GPU box): tools/prof_mlp.sh tag  - rocprofv3 kernel stats of the MLP micro-benchmark for the current build / env
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/pm_$1 -- python3 /root/repo/tools/bench_mlp.py --iters 10 > /root/repo/gpurun_out/pm_$1.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob('/root/repo/gpurun_out/pm_$1/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print(f"{r['Name'][:64]:64s} calls={r['Calls']:>4s} avg={float(r['AverageNs'])/1e3:8.1f} us")
PY
