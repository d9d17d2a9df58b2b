set -e
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/test_hip_dropin.py tests/test_hip_kernels.py tests/test_hip_fullsize.py tests/test_hip_noncubic.py tests/test_recon_utils.py tests/test_hip_joint.py tests/test_hip_step.py -x -q -m gpu > gpurun_out/r3_t13.log 2>&1 || { tail -40 gpurun_out/r3_t13.log | cut -c1-300; exit 1; }
tail -3 gpurun_out/r3_t13.log
bash tools/r3_dropin_prof.sh | head -12
grep -a ms_per_step gpurun_out/dropin_prof.log | cut -c1-700
python tools/dbg/dropin_prof.py 30 2>/dev/null | tail -1 | cut -c1-700
