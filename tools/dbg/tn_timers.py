"""Phase timers of the scene weight-gradient kernel k_gemm_tn_tr (library built with -DTN_TIMERS): ticks per work-group and phase."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from poseprobe_amd import bg_nerf, _lib
R, S = 1023, 128
opt = bg_nerf.default_options()
net = bg_nerf.NeRF(opt, device='cuda'); net.progress.data.fill_(0.6)
eng = bg_nerf.SceneEngine(net, lr=1e-3)
g = torch.Generator().manual_seed(0)
center = (torch.randn(R, 3, generator=g) * 0.3).cuda(); ray = torch.randn(R, 3, generator=g).cuda()
depth = ((torch.rand(R, S, generator=g) + torch.arange(S)) / S * 2.0 + 0.4).cuda(); image = torch.rand(R, 3, generator=g).cuda()
L = _lib.lib()
buf = (ctypes.c_ulonglong * 8)()
for _ in range(3): eng.step(center, ray, depth, image)
torch.cuda.synchronize(); L.pp_debug_read_tn_timers(buf, 1)
n = 10
for _ in range(n): eng.step(center, ray, depth, image)
torch.cuda.synchronize(); L.pp_debug_read_tn_timers(buf, 1)
names = ['wait for rows', 'convert + store', 'barrier 1', 'matrix phase', 'barrier 2', 'flush', '-', '-']
t = [buf[i] / n / 9 / 512 for i in range(8)]          # per launch (9 per step, a few with fewer work-groups) and work-group
tot = sum(t)
print(f'{tot:.0f} ticks per work-group and launch (16 chunks)')
for a, v in zip(names, t): print(f'  {a:16s} {v:9.0f}  {100 * v / max(tot, 1):5.1f} %')
