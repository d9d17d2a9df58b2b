"""The drop-in train step alone (bench.dropin_leg) - for rocprofv3 --kernel-trace --stats: which kernels, how much GPU time vs wall."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from poseprobe_amd import synthetic as syn
from poseprobe_amd.engine import SceneConfig
G, H, W, V, N = 160, 400, 400, 3, 1024
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device('cuda', 0)
rs = syn.range_shape()
cfg = SceneConfig(syn.XYZ_MIN, syn.XYZ_MAX, G ** 3, out_range=float(rs.max()))
views = syn.make_views(V, H, W)
idx_all, jit_all = [], []
for s in range(steps + 15):
    idx, jit = syn.step_randomness(V * H * W, N, seed=2000 + s)
    idx_all.append(idx); jit_all.append(jit)
idx_all = torch.tensor(np.stack(idx_all), dtype=torch.int32, device=dev)
jit_all = torch.tensor(np.stack(jit_all), dtype=torch.float32, device=dev)
r = bench.dropin_leg(dev, cfg, views, idx_all, jit_all, 10, G, H, W, V, N, 1.107, steps=steps, warmup=5)
print(json.dumps(r))
