#!/bin/bash
# usage (GPU box): tools/r02_final.sh  - end-of-round evidence: bench JSON, the same command under rocprofv3 (kernel stats), SQ counters of
# the MLP kernels in both arithmetic modes; summaries are copied into profiles/ by hand afterwards
R=/root/repo
cd $R && python3 bench.py > gpurun_out/r02_final_bench.json 2> gpurun_out/r02_final_bench.err || { tail -5 gpurun_out/r02_final_bench.err; exit 1; }
echo bench done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02_final_prof -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-psnr > $R/gpurun_out/r02_final_bench_under_rocprof.json 2> $R/gpurun_out/r02_final_prof.err || { tail -5 $R/gpurun_out/r02_final_prof.err; exit 1; }
echo bench-prof done
$R/tools/pmc_mlp.sh 31 && $R/tools/pmc_mlp.sh 0
