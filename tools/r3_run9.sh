set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_hip_step.py tests/test_hip_configs.py -x -q -m gpu -k "trajectory or every_optimiser_step" > gpurun_out/r3_t9.log 2>&1 || { tail -40 gpurun_out/r3_t9.log | cut -c1-300; exit 1; }
tail -3 gpurun_out/r3_t9.log
