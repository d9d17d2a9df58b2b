set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_recon_utils.py -x -q -m gpu > gpurun_out/r3_t14.log 2>&1 || { tail -40 gpurun_out/r3_t14.log | cut -c1-300; exit 1; }
tail -3 gpurun_out/r3_t14.log
