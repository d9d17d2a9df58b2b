set -e
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu --deselect tests/test_hip_psnr.py --deselect tests/test_hip_convergence.py > gpurun_out/r3_t3.log 2>&1
