set -e
for o in "nerf_tn256=0" "nerf_tn256=1" "nerf_tn_split_wgs=256" "nerf_tn_split_wgs=64"; do python tools/bench_scene.py 1023 128 30 3 4 $o 2>&1 | grep nerf_chain | cut -c1-150; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/ps_tn256 -- python3 /root/repo/tools/bench_scene.py 1023 128 20 3 4 nerf_tn256=1 > /root/repo/gpurun_out/ps_tn256.log 2>&1
cd /root/repo; python tools/show_stats.py gpurun_out/ps_tn256 6
