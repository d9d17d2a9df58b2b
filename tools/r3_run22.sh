set -e
for i in 1 2; do
(python tools/bench_scene.py 1023 128 30 3 8 && python tools/bench_scene.py 1023 128 30 3 4 && python tools/bench_scene.py 1023 128 30 0 4) 2>&1 | grep nerf_chain | cut -c1-120
done
(python tools/bench_scene.py 3072 128 20 3 8 && python tools/bench_scene.py 3072 128 20 3 4) 2>&1 | grep nerf_chain | cut -c1-120
PP_EXTRA_HIPCC_FLAGS="-DTR_TIMERS" python -m poseprobe_amd.build_ext --force > /dev/null 2>&1
python tools/dbg/tr_timers.py
PP_EXTRA_HIPCC_FLAGS="" python -m poseprobe_amd.build_ext --force > /dev/null 2>&1
