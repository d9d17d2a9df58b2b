"""Times one scene-branch optimisation step (SceneEngine.step) at the reference's training size and prints per-kernel-free
wall numbers; run under rocprofv3 for the kernel table."""
import sys, time, json, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poseprobe_amd import bg_nerf
R = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
S = int(sys.argv[2]) if len(sys.argv) > 2 else 128
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
opt = bg_nerf.default_options()
chain = int(sys.argv[4]) if len(sys.argv) > 4 else 3
nw = int(sys.argv[5]) if len(sys.argv) > 5 else 4
extra = {k: int(v) for k, v in (a.split('=') for a in sys.argv[6:])}
net = bg_nerf.NeRF(opt, device='cuda', options={'nerf_chain': chain, 'nerf_chain_nw': nw, **extra}); net.progress.data.fill_(0.6)
eng = bg_nerf.SceneEngine(net, lr=1e-3)
g = torch.Generator().manual_seed(0)
center = (torch.randn(R, 3, generator=g) * 0.3).cuda()
ray = torch.randn(R, 3, generator=g).cuda()
depth = ((torch.rand(R, S, generator=g) + torch.arange(S)) / S * 2.0 + 0.4).cuda()
image = torch.rand(R, 3, generator=g).cuda()
for _ in range(3):
    eng.step(center, ray, depth, image)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    loss, gc, gr = eng.step(center, ray, depth, image)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
M = R * S
fl_fwd = 2 * M * (64 * 256 + 256 * 256 * 6 + 320 * 256 + 256 + 288 * 128 + 128 * 3)
print(json.dumps(dict(nerf_chain=chain, nw=nw, **extra, rays=R, samples=S, ms_per_step=dt * 1e3, rays_per_s=R / dt, tflops=3 * fl_fwd / dt / 1e12, loss=float(loss))))
