set -e
timeout -k 10 600 python -m pytest tests/test_hip_scene.py -x -q -m gpu -k "oracle_on_seeded or buffers or reference_outputs" 2>&1 | tail -3
for o in "nerf_tn_stages=1" "nerf_tn_stages=2" "nerf_tn_stages=1" "nerf_tn_stages=2"; do python tools/bench_scene.py 1023 128 30 3 4 $o 2>&1 | grep nerf_chain | cut -c1-150; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/ps_tn2 -- python3 /root/repo/tools/bench_scene.py 1023 128 20 3 4 nerf_tn_stages=2 > /root/repo/gpurun_out/ps_tn2.log 2>&1
cd /root/repo; python tools/show_stats.py gpurun_out/ps_tn2 4
