set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_hip_scene.py -x -q -m gpu -k "trunk" > gpurun_out/r3_t17a.log 2>&1 || { tail -40 gpurun_out/r3_t17a.log | cut -c1-400; exit 1; }
tail -3 gpurun_out/r3_t17a.log
timeout -k 10 900 python -m pytest tests/test_hip_scene.py tests/test_hip_joint.py -x -q -m gpu > gpurun_out/r3_t17b.log 2>&1 || { tail -40 gpurun_out/r3_t17b.log | cut -c1-400; exit 1; }
tail -3 gpurun_out/r3_t17b.log
(python tools/bench_scene.py 1023 128 20 0 && python tools/bench_scene.py 1023 128 20 1 && python tools/bench_scene.py 1023 128 20 3 && python tools/bench_scene.py 3072 128 20 0 && python tools/bench_scene.py 3072 128 20 3) > gpurun_out/r3_scene17.log 2>&1 || { tail -20 gpurun_out/r3_scene17.log; exit 1; }
grep nerf_chain gpurun_out/r3_scene17.log
bash tools/prof_scene.sh c3 > /dev/null 2>&1
python tools/show_stats.py gpurun_out/ps_c3 16
