#!/bin/bash
# usage: tools/prof_gemm.sh tag   (GPU box) - rocprof kernel stats of the GEMM kernels for the current build / env
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/pg_$1 -- python3 /root/repo/bench.py --steps 6 --warmup 2 --no-cpu-baseline > /root/repo/gpurun_out/pg_$1.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob('/root/repo/gpurun_out/pg_$1/*/*kernel_stats.csv')[0]
rows=list(csv.DictReader(open(f)))
print('== $1', ' '.join(f"{r['Name'][5:22]}={float(r['AverageNs'])/1e3:.1f}" for r in rows if 'gemm' in r['Name']))
PY
