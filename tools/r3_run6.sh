set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_hip_fullsize.py tests/test_hip_kernels.py tests/test_hip_dropin.py tests/test_hip_noncubic.py tests/test_recon_utils.py tests/test_hip_joint.py -x -q -m gpu > gpurun_out/r3_t6.log 2>&1 || { tail -30 gpurun_out/r3_t6.log; exit 1; }
tail -3 gpurun_out/r3_t6.log
python tools/dbg/dropin_prof.py 20 2>/dev/null | tail -1 | cut -c1-700
