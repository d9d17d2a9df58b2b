set -e
mkdir -p gpurun_out
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/dropin_prof -- python3 $R/tools/dbg/dropin_prof.py 20 > $R/gpurun_out/dropin_prof.log 2>&1
cd $R
python3 tools/show_stats.py gpurun_out/dropin_prof 45
tail -2 gpurun_out/dropin_prof.log | cut -c1-600
