set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_hip_dropin.py tests/test_hip_fullsize.py -x -q -m gpu -k "inference or viewpoints or sync_free" > gpurun_out/r3_t15.log 2>&1 || { tail -40 gpurun_out/r3_t15.log | cut -c1-300; exit 1; }
tail -3 gpurun_out/r3_t15.log
python - <<'PY'
import torch, bench
print(bench.inference_leg(torch.device('cuda', 0), 160, 400, 400))
PY
