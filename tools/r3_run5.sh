set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_hip_fullsize.py tests/test_hip_kernels.py tests/test_hip_dropin.py tests/test_hip_noncubic.py tests/test_recon_utils.py -x -q -m gpu > gpurun_out/r3_t5.log 2>&1 || { tail -30 gpurun_out/r3_t5.log; exit 1; }
tail -3 gpurun_out/r3_t5.log
bash tools/r3_dropin_prof.sh | head -16
