import os, sys, torch
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
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for mode in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else '0,15').split(',')]:
    _lib.set_option('mlp_split', mode)
    tf = timeit(lambda: ops.warp_fwd(warp_p, pts, count, cap, 1.5, acts, out))
    tb = timeit(lambda: ops.warp_bwd(warp_p, pts, acts, g_out, count, cap, 1.5, scratch, wgrad, pgrad))
    print(f'mlp_split={mode}: warp_fwd {tf:7.1f} us   warp_bwd {tb:7.1f} us', flush=True)
