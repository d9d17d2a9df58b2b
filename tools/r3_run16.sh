set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_hip_scene.py -x -q -m gpu -k "trunk" > gpurun_out/r3_t16a.log 2>&1 || { tail -40 gpurun_out/r3_t16a.log | cut -c1-400; exit 1; }
tail -3 gpurun_out/r3_t16a.log
timeout -k 10 900 python -m pytest tests/test_hip_scene.py tests/test_hip_joint.py -x -q -m gpu > gpurun_out/r3_t16b.log 2>&1 || { tail -40 gpurun_out/r3_t16b.log | cut -c1-400; exit 1; }
tail -3 gpurun_out/r3_t16b.log
(python tools/bench_scene.py 1023 128 20 0 && python tools/bench_scene.py 1023 128 20 1 && python tools/bench_scene.py 3072 128 20 0 && python tools/bench_scene.py 3072 128 20 1) > gpurun_out/r3_scene16.log 2>&1 || { tail -20 gpurun_out/r3_scene16.log; exit 1; }
tail -12 gpurun_out/r3_scene16.log
