set -e
timeout -k 10 600 python -m pytest tests/test_hip_scene.py -x -q -m gpu -k "oracle_on_seeded or buffers or reference_outputs or trunk" 2>&1 | tail -3
for i in 1 2; do for h in 0 1; do python tools/bench_scene.py 1023 128 30 3 4 nerf_tn_tr=$h 2>&1 | grep nerf_chain | cut -c1-150; done; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/ps_tr1 -- python3 /root/repo/tools/bench_scene.py 1023 128 20 3 4 nerf_tn_tr=1 > /root/repo/gpurun_out/ps_tr1.log 2>&1
cd /root/repo; python tools/show_stats.py gpurun_out/ps_tr1 4
