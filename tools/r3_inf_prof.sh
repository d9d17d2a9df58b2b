set -e
R=$PWD
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/inf_prof -- python3 $R/tools/dbg/inference_prof.py > $R/gpurun_out/inf_prof.log 2>&1
tail -2 $R/gpurun_out/inf_prof.log
python3 $R/tools/show_stats.py $R/gpurun_out/inf_prof 30
for c in 8192 16384 40000; do python3 $R/tools/dbg/inference_prof.py $c 2>&1 | tail -1; done
