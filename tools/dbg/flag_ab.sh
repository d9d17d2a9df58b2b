#!/bin/bash
# A/B of extra compiler flags for pp_mlp_split.hip: builds with each flag set given as one quoted argument and times the warp kernels
cd /root/repo
for f in "$@"; do
  touch poseprobe_amd/csrc/pp_mlp_split.hip
  PP_EXTRA_HIPCC_FLAGS="$f" python -m poseprobe_amd.build_ext > /dev/null 2>&1 || { echo build failed; exit 1; }
  echo "flags: $f"; timeout -k 10 120 python tools/dbg/time_warp_fwd.py 3 || exit 1
done
touch poseprobe_amd/csrc/pp_mlp_split.hip
