set -e
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/test_hip_psnr.py -x -q -m gpu -s -k reference_configuration > gpurun_out/r3_t4.log 2>&1
