"""Phase timers of the scene trunk kernels (library built with -DTR_TIMERS): cycles per work-group and phase, wave 0 of each."""
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
buf = (ctypes.c_ulonglong * 16)()
for _ in range(3): eng.step(center, ray, depth, image)
torch.cuda.synchronize(); L.pp_debug_read_trunk_timers(buf, 1)
n = 10
for _ in range(n): eng.step(center, ray, depth, image)
torch.cuda.synchronize(); L.pp_debug_read_trunk_timers(buf, 1)
names = ['streamed steps', 'resident steps', 'epilogue', 'wait at A', 'convert', 'wait at B', 'stage top', '-']
for d, nm in ((0, 'forward'), (8, 'backward')):
    t = [buf[d + i] / n / 256 for i in range(8)]
    tot = sum(t)
    print(f'{nm}: {tot:.0f} ticks per work-group')
    for a, v in zip(names, t): print(f'  {a:16s} {v:9.0f}  {100 * v / max(tot, 1):5.1f} %')
