"""Whole-view inference under rocprofv3: prints wall time per view; the kernel stats come from the profiler (tools/r3_inf_prof.sh)."""
import sys, time
sys.path.insert(0, '/root/repo')
import torch, bench
dev = torch.device('cuda', 0)
chunk = int(sys.argv[1]) if len(sys.argv) > 1 else None
from poseprobe_amd import nvs_fun
if chunk:
    nvs_fun.CHUNK = chunk
    nvs_fun.render_view.__defaults__ = tuple(chunk if d == 4096 else d for d in nvs_fun.render_view.__defaults__)
print(bench.inference_leg(dev, 160, 400, 400, reps=5))
