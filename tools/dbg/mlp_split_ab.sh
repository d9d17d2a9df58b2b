#!/bin/bash
# A/B of the split-precision fused MLP kernels against the fp32 fused kernels (numerics + time)
cd /root/repo
PP_MLP_SPLIT=0 timeout -k 10 300 python tools/bench_mlp.py --save gpurun_out/mlp_ref.pt || exit 1
PP_MLP_SPLIT=${1:-15} timeout -k 10 300 python tools/bench_mlp.py --check gpurun_out/mlp_ref.pt
