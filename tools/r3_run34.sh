cd /root/repo
for v in "-DTN_TIMERS -DTN_DBG=0" "-DTN_TIMERS -DTN_DBG=1" "-DTN_TIMERS -DTN_DBG=2" "-DTN_TIMERS -DTN_DBG=3"; do
  PP_EXTRA_HIPCC_FLAGS="$v" python -m poseprobe_amd.build_ext --force > /dev/null 2>&1
  echo "flags: $v"; python tools/dbg/tn_timers.py 2>/dev/null
done
PP_EXTRA_HIPCC_FLAGS="" python -m poseprobe_amd.build_ext --force > /dev/null 2>&1
