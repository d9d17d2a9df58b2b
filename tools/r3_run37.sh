set -e
mkdir -p gpurun_out
PP_BENCH_SHARED_GPU=1 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 --warmup 3 --no-psnr --no-cpu-baseline --no-dropin --no-inference --no-fp32 > gpurun_out/r03_2rank.json 2> gpurun_out/r03_2rank.err || { tail -20 gpurun_out/r03_2rank.err; exit 1; }
tail -1 gpurun_out/r03_2rank.json | cut -c1-600
