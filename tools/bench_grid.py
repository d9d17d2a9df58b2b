"""Isolated A/B of the fused TV+Adam grid pass: dense vs sparse-gradient (touched-voxel byte map).  python tools/bench_grid.py [G] [frac]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poseprobe_amd import ops
G = int(sys.argv[1]) if len(sys.argv) > 1 else 160
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.084
dev = 'cuda'
C = 12
p = torch.randn(G, G, G, C, device=dev) * 0.1
po = torch.empty_like(p)
m, v, g = torch.zeros_like(p), torch.zeros_like(p), torch.zeros_like(p)
nvox = G ** 3
# clustered marks (rays are lines): mark runs of 4 consecutive voxels
hit = (torch.rand(nvox // 4, device=dev) < frac).repeat_interleave(4)
touched = hit.to(torch.uint8)
other = torch.zeros(nvox, dtype=torch.uint8, device=dev)
tv = torch.zeros(1, device=dev) if os.environ.get('NO_TV') != '1' else None      # NO_TV=1: no TV value (no same-address atomic per work-group)
print(f'G={G} marked {float(hit.float().mean()):.3f}')
def run(sparse, n=20):
    a = (p, po, g, m, v, (G, G, G), C, 0, G, 1e-4, 1.0, 0.1, 0.9, 0.99, 1e-8, 3, tv)
    f = (lambda: ops.grid_tv_adam_step_sparse(*a, touched, other)) if sparse else (lambda: ops.grid_tv_adam_step(*a))
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def run_cold(sparse, n=10):
    """Each launch preceded by a 2 GB streaming write + a burst of MFMA-free ALU work: caches / TLBs see other data first."""
    big = torch.empty(512 * 1024 * 1024, device=dev)
    a = (p, po, g, m, v, (G, G, G), C, 0, G, 1e-4, 1.0, 0.1, 0.9, 0.99, 1e-8, 3, tv)
    f = (lambda: ops.grid_tv_adam_step_sparse(*a, touched, other)) if sparse else (lambda: ops.grid_tv_adam_step(*a))
    ts = []
    for _ in range(n):
        big.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sum(ts[2:]) / len(ts[2:])
print(f'cold (2 GB written before every launch): dense {run_cold(False):7.1f} us   sparse {run_cold(True):7.1f} us')
for rep in range(2):
    td, ts = run(False), run(True)
    bd, bs = 384 * nvox, (288 + 96 * float(hit.float().mean())) * nvox
    print(f'dense {td:7.1f} us = {bd / td / 1e6:6.2f} TB/s   sparse {ts:7.1f} us = {bs / ts / 1e6:6.2f} TB/s (algorithmic)')
