set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_hip_scene.py -x -q -m gpu -k "stay_inside" > gpurun_out/r3_t7.log 2>&1 || { tail -30 gpurun_out/r3_t7.log; exit 1; }
tail -3 gpurun_out/r3_t7.log
