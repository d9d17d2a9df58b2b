set -e
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/test_hip_configs.py -x -q -m gpu -k "ref_96 or ref96" > gpurun_out/r3_t2.log 2>&1
