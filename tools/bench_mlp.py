"""Micro-benchmark + A/B check of the MLP chains (fused vs layer-by-layer kernels).

    PP_MLP_FUSED=0 python tools/bench_mlp.py --save gpurun_out/mlp_ref.pt
    PP_MLP_FUSED=1 python tools/bench_mlp.py --check gpurun_out/mlp_ref.pt
"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poseprobe_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument('--m', type=int, default=55000)
ap.add_argument('--cap', type=int, default=1024 * 186)
ap.add_argument('--save'); ap.add_argument('--check')
ap.add_argument('--iters', type=int, default=20)
a = ap.parse_args()
dev = torch.device('cuda:0')
g = torch.Generator(device='cpu').manual_seed(1)
M, cap = a.m, a.cap

def rnd(*s, scale=1.0):
    return (torch.randn(*s, generator=g) * scale).to(dev)

def timeit(fn, n):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

count = torch.tensor([M], dtype=torch.int32, device=dev)
res = {}
# ---- warp
warp_p = torch.zeros(50564 + 60, device=dev)
warp_p[:50564] = rnd(50564, scale=0.09)
pts = rnd(cap, 3, scale=0.5)
acts = torch.zeros(4 * cap * 4 * 128, device=dev)
out = torch.zeros(cap, 16, device=dev)
f = lambda: ops.warp_fwd(warp_p, pts, count, cap, 1.5, acts, out)
t = timeit(f, a.iters)
print(f'warp_fwd  {t:8.1f} us'); res['warp_out'] = out[:M].clone().cpu()
res['warp_act3'] = acts.view(4, cap * 4, 128)[3, :4 * M:97].clone().cpu()
g_out = rnd(cap, 16)
scratch = torch.zeros(3 * cap * 4 * 128 + 49152, device=dev)
wgrad = torch.zeros_like(warp_p); pgrad = torch.zeros(cap, 3, device=dev)
def fb():
    wgrad.zero_(); pgrad.zero_()
    ops.warp_bwd(warp_p, pts, acts, g_out, count, cap, 1.5, scratch, wgrad, pgrad)
t = timeit(fb, a.iters)
print(f'warp_bwd  {t:8.1f} us (incl. 2 zero fills)'); res['warp_wgrad'] = wgrad.clone().cpu(); res['warp_pgrad'] = pgrad[:M].clone().cpu()
# ---- rgbnet
rgb_p = torch.zeros(41731 + 60, device=dev)
rgb_p[:41731] = rnd(41731, scale=0.09)
feat = rnd(cap, 64); feat[:, 57:] = 0
racts = torch.zeros(3 * cap * 128, device=dev); rgb = torch.zeros(cap, 3, device=dev)
f = lambda: ops.rgbnet_fwd(rgb_p, feat, count, cap, racts, rgb)
t = timeit(f, a.iters)
print(f'rgb_fwd   {t:8.1f} us'); res['rgb'] = rgb[:M].clone().cpu()
g_rgb = rnd(cap, 3)
rscr = torch.zeros(3 * cap * 128 + 49152, device=dev)
rgrad = torch.zeros_like(rgb_p); fgrad = torch.zeros(cap, 64, device=dev)
def rb():
    rgrad.zero_()
    ops.rgbnet_bwd(rgb_p, feat, racts, rgb, g_rgb, count, cap, rscr, rgrad, fgrad)
t = timeit(rb, a.iters)
print(f'rgb_bwd   {t:8.1f} us (incl. zero fill)'); res['rgb_wgrad'] = rgrad.clone().cpu(); res['rgb_fgrad'] = fgrad[:M].clone().cpu()
if a.save:
    torch.save(res, a.save)
if a.check:
    ref = torch.load(a.check, weights_only=True)
    bad = 0
    for k, v in res.items():
        r = ref[k]
        err = (v - r).abs().max().item(); sc = r.abs().max().item()
        flag = '' if err <= 2e-4 * sc + 1e-6 else '  <-- MISMATCH'
        bad += bool(flag)
        print(f'  {k:12s} max|diff| {err:.3e}  (max|ref| {sc:.3e}){flag}')
    sys.exit(1 if bad else 0)
