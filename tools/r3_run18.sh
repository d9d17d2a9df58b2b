set -e
bash tools/prof_scene.sh c3 > /dev/null 2>&1
python tools/show_stats.py gpurun_out/ps_c3 14
