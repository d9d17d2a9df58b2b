cd /root/repo
for v in "-DMS_NT=0" "-DMS_NT=1" "-DMS_NT=0" "-DMS_NT=1"; do
  PP_EXTRA_HIPCC_FLAGS="$v" python -m poseprobe_amd.build_ext --force > /dev/null 2>&1
  echo "flags: $v"; python bench.py --gpus 1 --steps 100 --warmup 10 --no-psnr --no-cpu-baseline --no-dual --no-dropin --no-inference --no-fp32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], {k:round(v['ms'],4) for k,v in d['roofline_mlp']['kernels'].items()})"
done
PP_EXTRA_HIPCC_FLAGS="" python -m poseprobe_amd.build_ext --force > /dev/null 2>&1
