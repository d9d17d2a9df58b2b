"""Does the fused grid pass care how its five arrays are placed relative to each other (HBM channel / bank aliasing)?  Same kernel, arrays
carved out of one arena at different relative offsets."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from poseprobe_amd import ops
G, C, dev = 160, 12, 'cuda'
n = G * G * G * C
nvox = G ** 3
hit = (torch.rand(nvox // 4, device=dev) < 0.084).repeat_interleave(4)
touched, other = hit.to(torch.uint8), torch.zeros(nvox, dtype=torch.uint8, device=dev)
tv = torch.zeros(1, device=dev)
big = torch.empty(512 * 1024 * 1024, device=dev)

def carve(skew_floats):
    """five arrays of n floats; array k starts at k * (n + pad) + k * skew"""
    arena = torch.zeros(5 * n + 5 * (1 << 22), device=dev)
    base = arena.data_ptr()
    outs, o = [], (-(base // 4)) % 1024                      # 4 KB aligned start
    for k in range(5):
        start = o + k * skew_floats
        outs.append(arena[start:start + n].view(G, G, G, C))
        o += n
        o += (-o) % 1024
    outs[0].normal_(0, 0.1)
    return outs, arena

def bench(arrs, cold=True, reps=12):
    p, po, g, m, v = arrs
    a = (p, po, g, m, v, (G, G, G), C, 0, G, 1e-4, 1.0, 0.1, 0.9, 0.99, 1e-8, 3, tv)
    ts = []
    for _ in range(reps):
        if cold:
            big.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.grid_tv_adam_step_sparse(*a, touched, other); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts = sorted(ts[2:])
    return ts[len(ts) // 2]

import sys
skews = [int(a) // 4 for a in sys.argv[1:]] or [0, 64, 1024, 1024 + 64, 16384 + 64, 65536 + 256, 262144 + 1024 + 64, 1 << 20]
for skew in skews:
    arrs, arena = carve(skew)
    ptrs = [a.data_ptr() for a in arrs]
    print(f'skew {skew * 4:9d} B: cold {bench(arrs):7.1f} us  warm {bench(arrs, cold=False):7.1f} us   (ptr deltas mod 1 MiB: {[ (q - ptrs[0]) % (1 << 20) for q in ptrs]})')
    del arrs, arena
    torch.cuda.empty_cache()
