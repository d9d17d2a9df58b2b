"""Time pp_color_feat_fwd at the train step's shape with a given build of the library (experiments: CF_DBG variants).
    python tools/dbg/time_cf.py [path/to/lib.so]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from poseprobe_amd import _lib
if len(sys.argv) > 1:
    _lib.SO_PATH = os.path.abspath(sys.argv[1])
from poseprobe_amd import ops, synthetic as syn
from poseprobe_amd.engine import SceneConfig
cfg = SceneConfig(syn.XYZ_MIN, syn.XYZ_MAX, 160 ** 3, out_range=1.0)
g = torch.Generator().manual_seed(3)
M, cap, R = 55000, 190464, 1024
X, Y, Z = cfg.world_size
k0 = (torch.randn(X, Y, Z, 12, generator=g) * 0.1).cuda()
lo, hi = torch.tensor(syn.XYZ_MIN), torch.tensor(syn.XYZ_MAX)
pts = (lo + (hi - lo) * torch.rand(cap, 3, generator=g)).cuda()
vd = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1).cuda()
rid = torch.sort(torch.randint(0, R, (cap,), generator=g, dtype=torch.int32)).values.cuda()
grad = torch.randn(cap, 3, generator=g).cuda()
pe_w = torch.rand(6, generator=g).cuda()
count = torch.tensor([M], dtype=torch.int32).cuda()
feat = torch.empty(cap, 64).cuda()
big = torch.empty(64 * 1024 * 1024, device='cuda')
ts = []
for i in range(12):
    big.fill_(1.0)                               # evict k0 from the caches between launches
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.color_feat_fwd(cfg.pp, k0, pts, vd, rid, grad, pe_w, count, cap, feat); e1.record()
    torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
print(os.path.basename(_lib.SO_PATH), 'color_feat_fwd %.1f us (min %.1f)' % (sum(ts[2:]) / len(ts[2:]), min(ts[2:])))
