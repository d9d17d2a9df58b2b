"""Debug: forward / backward of the fused warp + rgb MLPs in fp32 and split mode inside one process, elementwise comparison."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from poseprobe_amd import ops, _lib
dev = torch.device('cuda:0')
g = torch.Generator(device='cpu').manual_seed(1)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
cap = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
bits = int(sys.argv[3]) if len(sys.argv) > 3 else 15
rnd = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).to(dev)
count = torch.tensor([M], dtype=torch.int32, device=dev)
warp_p = torch.zeros(50564 + 60, device=dev); warp_p[:50564] = rnd(50564, scale=0.09)
pts = rnd(cap, 3, scale=0.5)
g_out = rnd(cap, 16)

def run(mode):
    _lib.set_option('mlp_split', mode)
    acts = torch.zeros(4 * cap * 4 * 128, device=dev); out = torch.zeros(cap, 16, device=dev)
    ops.warp_fwd(warp_p, pts, count, cap, 1.5, acts, out)
    scratch = torch.zeros(3 * cap * 4 * 128 + 49152, device=dev)
    wgrad = torch.zeros_like(warp_p); pgrad = torch.zeros(cap, 3, device=dev)
    ops.warp_bwd(warp_p, pts, acts, g_out, count, cap, 1.5, scratch, wgrad, pgrad)
    torch.cuda.synchronize()
    return dict(acts=acts.view(4, cap * 4, 128)[:, :4 * M].clone(), out=out[:M].clone(), ybar=scratch[:3 * cap * 4 * 128].view(3, cap * 4, 128)[:, :4 * M].clone(),
                wgrad=wgrad.clone(), pgrad=pgrad[:M].clone())

a, b = run(0), run(bits)
for k in a:
    d = (a[k] - b[k]).abs()
    print(f'{k:6s} max|diff| {float(d.max()):.3e}  max|ref| {float(a[k].abs().max()):.3e}  nan {int(torch.isnan(b[k]).sum())}')
    if k in ('acts', 'ybar'):
        for l in range(a[k].shape[0]):
            dl = d[l]; bad = dl > 1e-4 * float(a[k][l].abs().max())
            rows = bad.any(1).nonzero().flatten()
            print(f'   layer {l}: max {float(dl.max()):.3e}, bad rows {len(rows)} {rows[:12].tolist()} cols {bad.any(0).nonzero().flatten()[:12].tolist()}')
    if k == 'out':
        bad = (d > 1e-4).any(1).nonzero().flatten()
        print('   bad samples', len(bad), bad[:20].tolist())

# ---- rgbnet
rgb_p = torch.zeros(41731 + 60, device=dev); rgb_p[:41731] = rnd(41731, scale=0.09)
feat = rnd(cap, 64); feat[:, 57:] = 0
g_rgb = rnd(cap, 3)
def run_rgb(mode):
    _lib.set_option('mlp_split', mode)
    racts = torch.zeros(3 * cap * 128, device=dev); rgb = torch.zeros(cap, 3, device=dev)
    ops.rgbnet_fwd(rgb_p, feat, count, cap, racts, rgb)
    rscr = torch.zeros(3 * cap * 128 + 49152, device=dev)
    rgrad = torch.zeros_like(rgb_p); fgrad = torch.zeros(cap, 64, device=dev)
    ops.rgbnet_bwd(rgb_p, feat, racts, rgb, g_rgb, count, cap, rscr, rgrad, fgrad)
    torch.cuda.synchronize()
    return dict(acts=racts.view(3, cap, 128)[:, :M].clone(), rgb=rgb[:M].clone(), ybar=rscr[:3 * cap * 128].view(3, cap, 128)[:, :M].clone(),
                wgrad=rgrad.clone(), fgrad=fgrad[:M].clone())
a, b = run_rgb(0), run_rgb(bits)
for k in a:
    d = (a[k] - b[k]).abs()
    print(f'rgb {k:6s} max|diff| {float(d.max()):.3e}  max|ref| {float(a[k].abs().max()):.3e}  nan {int(torch.isnan(b[k]).sum())}')
    if k in ('acts', 'ybar'):
        for l in range(3):
            bad = d[l] > 1e-4 * float(a[k][l].abs().max())
            print(f'   layer {l}: max {float(d[l].max()):.3e}, bad elements {int(bad.sum())} in rows {bad.any(1).nonzero().flatten()[:8].tolist()}')
