#!/bin/bash
# phase timers of the split forward kernel for experiment builds (-DMS_DBG=n)
cd /root/repo
for v in "$@"; do
  touch poseprobe_amd/csrc/pp_mlp_split.hip
  PP_EXTRA_HIPCC_FLAGS="-DMS_TIMERS -DMS_DBG=$v" python -m poseprobe_amd.build_ext > /dev/null 2>&1 || { echo build failed; exit 1; }
  echo "MS_DBG=$v"; timeout -k 10 120 python tools/dbg/ms_timers.py || exit 1
done
touch poseprobe_amd/csrc/pp_mlp_split.hip
