"""Time of the atomic and the sorted k0 scatter at the bench workload (160^3, capacity 1024 x 186, 55 k samples)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from poseprobe_amd import ops, synthetic as syn
from poseprobe_amd.engine import SceneConfig
dev = 'cuda'
rs = syn.range_shape()
cfg = SceneConfig(syn.XYZ_MIN, syn.XYZ_MAX, 160 ** 3, out_range=float(rs.max()))
X, Y, Z = cfg.world_size
cap, M = 1024 * 186, 55000
g = torch.Generator().manual_seed(0)
lo, hi = torch.tensor(cfg.xyz_min), torch.tensor(cfg.xyz_max)
pts = (lo + (hi - lo) * torch.rand(cap, 3, generator=g)).float().to(dev)
gf = torch.randn(cap, 64, generator=g).to(dev)
cnt = torch.tensor([M], dtype=torch.int32, device=dev)
grad = torch.zeros(X, Y, Z, 12, device=dev); t = torch.zeros(X * Y * Z, dtype=torch.uint8, device=dev)
work = torch.empty(ops.k0_scatter_sorted_workspace(cap), dtype=torch.uint8, device=dev)
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print(f'atomic {timeit(lambda: ops.k0_scatter_samples(cfg.pp, pts, cnt, cap, gf, grad, t)):.1f} us, sorted '
      f'{timeit(lambda: ops.k0_scatter_samples_sorted(cfg.pp, pts, cnt, cap, gf, grad, work, t)):.1f} us, workspace {work.numel() / 1e6:.1f} MB')
