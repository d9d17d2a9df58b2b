set -e
timeout -k 10 600 python -m pytest tests/test_hip_scene.py -x -q -m gpu -k "trunk or oracle_on_seeded or buffers" 2>&1 | tail -4
(python tools/bench_scene.py 1023 128 20 3 && python tools/bench_scene.py 3072 128 20 3) 2>&1 | grep nerf_chain
bash tools/prof_scene.sh c3 > /dev/null 2>&1
python tools/show_stats.py gpurun_out/ps_c3 4
