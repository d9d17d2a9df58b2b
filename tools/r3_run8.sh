set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_hip_fullsize.py -x -q -m gpu -k "dist_choreographies" > gpurun_out/r3_t8.log 2>&1 || { tail -40 gpurun_out/r3_t8.log | cut -c1-300; exit 1; }
tail -3 gpurun_out/r3_t8.log
