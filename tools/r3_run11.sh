set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_hip_dropin.py tests/test_hip_step.py tests/test_recon_utils.py tests/test_hip_dvgo_ops.py -x -q -m gpu > gpurun_out/r3_t11.log 2>&1 || { tail -40 gpurun_out/r3_t11.log | cut -c1-300; exit 1; }
tail -3 gpurun_out/r3_t11.log
python tools/dbg/dropin_prof.py 20 2>/dev/null | tail -1 | cut -c1-800
