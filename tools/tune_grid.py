"""Micro-benchmark of the fused TV+Adam grid pass (dominant kernel): GB/s of algorithmic traffic vs PP_GRID_CHUNKS."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poseprobe_amd import ops, _lib  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 160
C = 12
bufs = [torch.randn(G, G, G, C, device='cuda') * 0.1 for _ in range(5)]
bufs[4].abs_()
tv = torch.zeros(1, device='cuda')
for chunks in [int(c) for c in (sys.argv[2].split(',') if len(sys.argv) > 2 else ['0'])]:
    _lib.set_option('grid_chunks', chunks)          # 0 = the library's heuristic
    for _ in range(3):
        ops.grid_tv_adam_step(bufs[0], bufs[1], bufs[2], bufs[3], bufs[4], (G, G, G), C, 0, G, 1e-9, 1.0, 0.1, 0.9, 0.99, 1e-8, 3, tv)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 20
    for _ in range(n):
        ops.grid_tv_adam_step(bufs[0], bufs[1], bufs[2], bufs[3], bufs[4], (G, G, G), C, 0, G, 1e-9, 1.0, 0.1, 0.9, 0.99, 1e-8, 3, tv)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f'G={G} chunks={chunks or "default"}: {ms * 1e3:.1f} us  {384 * G ** 3 / ms / 1e6:.0f} GB/s', flush=True)
