#!/bin/bash
# usage (GPU box): tools/r02_profiles.sh   - round-2 evidence: kernel stats of the bench (under rocprofv3) and of one scene step,
# SQ counters of the scene-branch kernels; summaries are copied into profiles/ by hand afterwards
cd /tmp && export TMPDIR=/tmp
R=/root/repo
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02_bench_prof -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-psnr > $R/gpurun_out/r02_bench_under_rocprof.json 2> $R/gpurun_out/r02_bench_prof.err || { tail -5 $R/gpurun_out/r02_bench_prof.err; exit 1; }
echo bench-prof done
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02_scene_prof -- python3 $R/tools/bench_scene.py 1023 128 20 > $R/gpurun_out/r02_scene_step_1023x128.json 2>&1 || exit 1
echo scene-prof done
$R/tools/pmc_scene.sh
cp $R/gpurun_out/pmc_scene.json $R/gpurun_out/r02_scene_pmc.json
