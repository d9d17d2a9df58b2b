set -e
mkdir -p gpurun_out
timeout -k 10 1150 python -m pytest tests -x -q -m gpu > gpurun_out/r3_full_gpu.log 2>&1 || { tail -40 gpurun_out/r3_full_gpu.log | cut -c1-300; exit 1; }
tail -4 gpurun_out/r3_full_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
