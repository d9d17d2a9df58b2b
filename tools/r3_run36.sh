cd /root/repo
for v in "-DTN_DBG=0" "-DTN_DBG=4"; do
  PP_EXTRA_HIPCC_FLAGS="$v" python -m poseprobe_amd.build_ext --force > /dev/null 2>&1
  bash tools/prof_scene.sh tn > /dev/null 2>&1
  echo "flags: $v"; python tools/show_stats.py gpurun_out/ps_tn 3 | grep gemm_tn
done
PP_EXTRA_HIPCC_FLAGS="" python -m poseprobe_amd.build_ext --force > /dev/null 2>&1
