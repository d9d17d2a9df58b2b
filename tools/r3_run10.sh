set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_hip_step.py tests/test_hip_dropin.py -x -q -m gpu -k "optimizer_state or checkpoint or reference_style" > gpurun_out/r3_t10.log 2>&1 || { tail -40 gpurun_out/r3_t10.log | cut -c1-300; exit 1; }
tail -3 gpurun_out/r3_t10.log
