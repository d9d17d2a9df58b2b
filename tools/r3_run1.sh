set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_recon_utils.py tests/test_hip_joint.py tests/test_hip_dropin.py -x -q -m gpu > gpurun_out/r3_t1.log 2>&1
