#!/bin/bash
# usage (GPU box): tools/prof_scene.sh tag [rays samples]  - rocprofv3 kernel stats of one scene-branch optimisation step
cd /tmp && export TMPDIR=/tmp
R=${2:-1023}; S=${3:-128}
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/ps_$1 -- python3 /root/repo/tools/bench_scene.py $R $S 20 > /root/repo/gpurun_out/ps_$1.log 2>&1 || { tail -5 /root/repo/gpurun_out/ps_$1.log; exit 1; }
tail -1 /root/repo/gpurun_out/ps_$1.log
