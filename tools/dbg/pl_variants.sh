#!/bin/bash
# builds and times the k_gemm256p experiment variants on the GPU box
cd /root/repo
for v in 0 1 2 3; do
  PP_EXTRA_HIPCC_FLAGS="-DPL_DBG=$v" python -m poseprobe_amd.build_ext --force > /dev/null 2>&1
  tools/prof_scene.sh dbg$v > /dev/null 2>&1
  echo "PL_DBG=$v"; python tools/show_stats.py gpurun_out/ps_dbg$v 3 | grep gemm256
done
PP_EXTRA_HIPCC_FLAGS="" python -m poseprobe_amd.build_ext --force > /dev/null 2>&1
