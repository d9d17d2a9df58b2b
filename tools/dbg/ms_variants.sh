#!/bin/bash
# builds pp_mlp_split.hip with -DMS_DBG=n for each n given and times the warp forward / backward kernels
cd /root/repo
for v in "$@"; do
  touch poseprobe_amd/csrc/pp_mlp_split.hip
  PP_EXTRA_HIPCC_FLAGS="-DMS_DBG=$v" python -m poseprobe_amd.build_ext > /dev/null 2>&1 || { echo build failed; exit 1; }
  echo "MS_DBG=$v"; timeout -k 10 120 python tools/dbg/time_warp_fwd.py 15 || exit 1
done
touch poseprobe_amd/csrc/pp_mlp_split.hip
