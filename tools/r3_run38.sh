set -e
timeout -k 10 600 python -m pytest tests/test_hip_scene.py -x -q -m gpu -k "oracle_on_seeded or reference_outputs or small_magnitude or trunk" 2>&1 | tail -2
bash tools/prof_scene.sh w > /dev/null 2>&1
python tools/show_stats.py gpurun_out/ps_w 30 | grep -E "wmax|bench"
tail -1 gpurun_out/ps_w.log | cut -c1-140
