set -e
timeout -k 10 600 python -m pytest tests/test_hip_scene.py -x -q -m gpu -k "trunk or oracle_on_seeded or buffers or reference_outputs" 2>&1 | tail -4
for i in 1 2; do python tools/bench_scene.py 1023 128 30 3 4 2>&1 | grep nerf_chain | cut -c1-130; done
bash tools/prof_scene.sh c3 > /dev/null 2>&1
python tools/show_stats.py gpurun_out/ps_c3 5
