"""Debug: error of the fused warp forward (fp32 / split) against a float64 torch evaluation of the 4-row form, flipped samples excluded."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from poseprobe_amd import ops, _lib
from tests.test_hip_mlp import _warp_params, _warp_ref, _pack, OUT_RANGE
dev = 'cuda'
M = cap = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
scale_w = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
layers = [(W * scale_w, b * scale_w) for W, b in _warp_params(3)]
g = torch.Generator().manual_seed(5)
pts_h = torch.randn(cap, 3, generator=g) * 0.5
ref = _warp_ref([(W.double(), b.double()) for W, b in layers], pts_h.double()).reshape(M, 16)
P = _pack(layers)
params = torch.zeros(P.numel() + 60, device=dev); params[:P.numel()] = P.to(dev)
count = torch.tensor([M], dtype=torch.int32, device=dev)
for mode in (0, 1):
    _lib.set_option('mlp_split', mode)
    acts = torch.zeros(4 * cap * 4 * 128, device=dev); out = torch.zeros(cap, 16, device=dev)
    ops.warp_fwd(params, pts_h.to(dev), count, cap, OUT_RANGE, acts, out)
    err = (out.cpu().double() - ref).abs()
    rowbad = (err > 1e-4 * ref.abs().max()).any(1)
    e = err[~rowbad]
    r = ref[~rowbad]
    print(f'mode {mode}: flipped samples {int(rowbad.sum())}; rms err {float((e ** 2).mean().sqrt()):.3e}, max err {float(e.max()):.3e}, rms ref {float((r ** 2).mean().sqrt()):.3e};'
          f' value cols rms err {float((e[:, ::4] ** 2).mean().sqrt()):.3e}, jacobian cols rms err {float((e.reshape(-1, 4, 4)[:, :, 1:] ** 2).mean().sqrt()):.3e}')
