#!/bin/bash
# builds and times k_nerf_trunk_fwd experiment variants on the GPU box: arguments are hipcc flag sets, e.g. "-DTR_DBG=4" "-DTR_NT=0"
cd /root/repo
for v in "$@"; do
  PP_EXTRA_HIPCC_FLAGS="$v" python -m poseprobe_amd.build_ext --force > /dev/null 2>&1
  tag=$(echo "$v" | tr -c 'A-Za-z0-9\n' '_')
  tools/prof_scene.sh tr$tag > /dev/null 2>&1
  echo "$v"; python tools/show_stats.py gpurun_out/ps_tr$tag 12 | grep -E "trunk"
done
PP_EXTRA_HIPCC_FLAGS="" python -m poseprobe_amd.build_ext --force > /dev/null 2>&1
