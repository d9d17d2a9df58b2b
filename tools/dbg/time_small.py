import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from poseprobe_amd import ops, _lib
dev = torch.device('cuda:0'); g = torch.Generator(device='cpu').manual_seed(1)
cap = 1024 * 186
rnd = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).to(dev)
warp_p = torch.zeros(50564 + 60, device=dev); warp_p[:50564] = rnd(50564, scale=0.09)
rgb_p = torch.zeros(41731 + 60, device=dev); rgb_p[:41731] = rnd(41731, scale=0.09)
pts = rnd(cap, 3, scale=0.5); acts = torch.zeros(4 * cap * 4 * 128, device=dev); out = torch.zeros(cap, 16, device=dev)
g_out = rnd(cap, 16); scratch = torch.zeros(3 * cap * 4 * 128 + 49152, device=dev)
wgrad = torch.zeros_like(warp_p); pgrad = torch.zeros(cap, 3, device=dev)
feat = rnd(cap, 64); racts = torch.zeros(3 * cap * 128, device=dev); rgb = torch.zeros(cap, 3, device=dev)
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for M in (16, 4096, 8192, 55000):
    count = torch.tensor([M], dtype=torch.int32, device=dev)
    for mode in (0, 15):
        _lib.set_option('mlp_split', mode)
        tf = timeit(lambda: ops.warp_fwd(warp_p, pts, count, cap, 1.5, acts, out))
        tb = timeit(lambda: ops.warp_bwd_data(warp_p, pts, acts, g_out, count, cap, 1.5, scratch, wgrad, pgrad))
        tr = timeit(lambda: ops.rgbnet_fwd(rgb_p, feat, count, cap, racts, rgb))
        print(f'M={M:6d} mlp_split={mode:2d}: warp_fwd {tf:7.1f}  warp_bwd_data {tb:7.1f}  rgb_fwd {tr:7.1f} us', flush=True)
