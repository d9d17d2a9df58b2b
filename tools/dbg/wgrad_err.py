"""Debug: error of the weight-gradient kernels (fp32 chain / split chain) against float64 products of the same Ybar and X."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from poseprobe_amd import ops, _lib
dev = torch.device('cuda:0'); g = torch.Generator(device='cpu').manual_seed(1)
M, cap = 55000, 1024 * 186
gscale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
rnd = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).to(dev)
count = torch.tensor([M], dtype=torch.int32, device=dev)
warp_p = torch.zeros(50564 + 60, device=dev); warp_p[:50564] = rnd(50564, scale=0.09)
pts = rnd(cap, 3, scale=0.5); acts = torch.zeros(4 * cap * 4 * 128, device=dev); out = torch.zeros(cap, 16, device=dev)
# gradients with a wide dynamic range over the samples (as in training: most samples contribute little)
g_out = rnd(cap, 16) * gscale * torch.exp(rnd(cap, 1) * 3.0)
scratch = torch.zeros(3 * cap * 4 * 128 + 49152, device=dev)
wgrad = torch.zeros_like(warp_p); pgrad = torch.zeros(cap, 3, device=dev)
_lib.set_option('mlp_split', 15)
ops.warp_fwd(warp_p, pts, count, cap, 1.5, acts, out)
stage2 = ops.warp_bwd_data(warp_p, pts, acts, g_out, count, cap, 1.5, scratch, wgrad, pgrad)
R = 4 * M
X = acts.view(4, cap * 4, 128)[:, :R].double()
Y = scratch[:3 * cap * 4 * 128].view(3, cap * 4, 128)[:, :R].double()
# layers: (Ybar3, X2 -> W3), (Ybar2, X1 -> W2), (Ybar1, X0 -> W1); parameter block: W0[128x3] b0 | W1 b1 | W2 b2 | W3 b3 | ...
off = {1: 128 * 3 + 128, 2: 128 * 3 + 128 + 128 * 128 + 128, 3: 128 * 3 + 128 + 2 * (128 * 128 + 128)}
ref = {3: Y[0].T @ X[2], 2: Y[1].T @ X[1], 1: Y[2].T @ X[0]}
for mode in (15, 31):
    _lib.set_option('mlp_split', mode)
    wg = torch.zeros_like(warp_p)
    ops.warp_bwd_weights(acts, scratch, count, cap, wg, stage2)
    torch.cuda.synchronize()
    for l in (3, 2, 1):
        got = wg[off[l]:off[l] + 128 * 128].view(128, 128).double()
        err = (got - ref[l]).abs()
        print(f'mode {mode} W{l}: rms err {float((err ** 2).mean().sqrt()):.3e}  max err {float(err.max()):.3e}  rms ref {float((ref[l] ** 2).mean().sqrt()):.3e}  '
              f'rel rms {float((err ** 2).mean().sqrt() / (ref[l] ** 2).mean().sqrt()):.3e}', flush=True)
