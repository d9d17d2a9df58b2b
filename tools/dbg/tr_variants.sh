#!/bin/bash
# builds and times the k_nerf_trunk_fwd experiment variants on the GPU box
cd /root/repo
for v in ${@:-0 1 2 3 4 5}; do
  PP_EXTRA_HIPCC_FLAGS="-DTR_DBG=$v" python -m poseprobe_amd.build_ext --force > /dev/null 2>&1
  tools/prof_scene.sh tr$v > /dev/null 2>&1
  echo "TR_DBG=$v"; python tools/show_stats.py gpurun_out/ps_tr$v 12 | grep -E "trunk|pack_trunk|gemm256p"
done
PP_EXTRA_HIPCC_FLAGS="" python -m poseprobe_amd.build_ext --force > /dev/null 2>&1
