set -e
timeout -k 10 600 python -m pytest tests/test_hip_scene.py -x -q -m gpu -k "oracle_on_seeded or buffers or reference_outputs or trunk or edge_shapes" > gpurun_out/r3_t35.log 2>&1 || { tail -30 gpurun_out/r3_t35.log | cut -c1-300; exit 1; }
tail -2 gpurun_out/r3_t35.log
for i in 1 2; do for h in 0 1 2; do python tools/bench_scene.py 1023 128 30 3 4 nerf_tn_tr=$h 2>&1 | grep nerf_chain | cut -c1-150; done; done
for h in 1 2; do python tools/bench_scene.py 3072 128 20 3 4 nerf_tn_tr=$h 2>&1 | grep nerf_chain | cut -c1-150; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/ps_tr2 -- python3 /root/repo/tools/bench_scene.py 1023 128 20 3 4 nerf_tn_tr=2 > /root/repo/gpurun_out/ps_tr2.log 2>&1
cd /root/repo; python tools/show_stats.py gpurun_out/ps_tr2 4
