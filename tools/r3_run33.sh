set -e
PP_EXTRA_HIPCC_FLAGS="-DTN_TIMERS" python -m poseprobe_amd.build_ext --force > /dev/null 2>&1
python tools/dbg/tn_timers.py
PP_EXTRA_HIPCC_FLAGS="" python -m poseprobe_amd.build_ext --force > /dev/null 2>&1
