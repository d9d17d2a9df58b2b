set -e
timeout -k 10 600 python -m pytest tests/test_hip_scene.py -x -q -m gpu -k "oracle_on_seeded or reference_outputs or small_magnitude or trunk or buffers" 2>&1 | tail -2
for i in 1 2; do python tools/bench_scene.py 1023 128 30 3 4 2>&1 | grep nerf_chain | cut -c1-130; done
bash tools/prof_scene.sh w > /dev/null 2>&1
python tools/show_stats.py gpurun_out/ps_w 14
