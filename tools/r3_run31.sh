set -e
for i in 1 2 3; do for h in 0 1; do python tools/bench_scene.py 1023 128 30 3 4 nerf_chain_head=$h 2>&1 | grep nerf_chain | cut -c1-150; done; done
for h in 0 1; do python tools/bench_scene.py 3072 128 20 3 4 nerf_chain_head=$h 2>&1 | grep nerf_chain | cut -c1-150; done
