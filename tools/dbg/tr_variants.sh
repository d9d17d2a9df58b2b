#!/bin/bash
# builds and times k_nerf_trunk experiment variants on the GPU box: arguments are hipcc flag sets, e.g. "-DTR_DBG=4" "-DTR_NT=0"
cd /root/repo
for v in "$@"; do
  PP_EXTRA_HIPCC_FLAGS="$v" python -m poseprobe_amd.build_ext --force > /dev/null 2>&1
  tag=$(echo "$v" | tr -c 'A-Za-z0-9\n' '_')
  for nw in 8 4; do
    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/ps_tr${tag}_$nw -- python3 /root/repo/tools/bench_scene.py 1023 128 20 3 $nw > /root/repo/gpurun_out/ps_tr${tag}_$nw.log 2>&1
    cd /root/repo
    echo "$v nw=$nw"; python tools/show_stats.py gpurun_out/ps_tr${tag}_$nw 12 | grep -E "trunk"
  done
done
PP_EXTRA_HIPCC_FLAGS="" python -m poseprobe_amd.build_ext --force > /dev/null 2>&1
