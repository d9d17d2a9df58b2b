"""Phase timers of the split weight-gradient kernel (library built with -DMS_TIMERS)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from poseprobe_amd import ops, _lib
dev = torch.device('cuda:0'); g = torch.Generator(device='cpu').manual_seed(1)
M, cap = 55000, 1024 * 186
rnd = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).to(dev)
count = torch.tensor([M], dtype=torch.int32, device=dev)
warp_p = torch.zeros(50564 + 60, device=dev); warp_p[:50564] = rnd(50564, scale=0.09)
pts = rnd(cap, 3, scale=0.5); acts = torch.zeros(4 * cap * 4 * 128, device=dev); out = torch.zeros(cap, 16, device=dev)
g_out = rnd(cap, 16); scratch = torch.zeros(3 * cap * 4 * 128 + 49152, device=dev)
wgrad = torch.zeros_like(warp_p); pgrad = torch.zeros(cap, 3, device=dev)
_lib.set_option('mlp_split', 31)
ops.warp_fwd(warp_p, pts, count, cap, 1.5, acts, out)
stage2 = ops.warp_bwd_data(warp_p, pts, acts, g_out, count, cap, 1.5, scratch, wgrad, pgrad)
L = _lib.lib(); buf = (ctypes.c_ulonglong * 16)()
for _ in range(3): ops.warp_bwd_weights(acts, scratch, count, cap, wgrad, stage2)
torch.cuda.synchronize(); L.pp_debug_read_timers(buf, 1)
n = 10
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n): ops.warp_bwd_weights(acts, scratch, count, cap, wgrad, stage2)
e1.record(); torch.cuda.synchronize(); L.pp_debug_read_timers(buf, 1)
t = [buf[i] / n / 255 for i in range(16)]
tot = sum(t)
print(f'kernel {e0.elapsed_time(e1) / n * 1e3:.1f} us; timer total {tot:.0f} ticks per work-group (~{3440 * 3 / 255:.1f} tile-layers each)')
for nm, v in zip(['wait+barrier', 'prepare', 'issue', 'compute', 'loop'], t): print(f'  {nm:14s} {v:9.0f}  {100 * v / tot:5.1f} %')
