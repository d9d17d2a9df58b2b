"""Bitwise A/B of one entry point between two builds of the library (GPU box).
    python tools/dbg/ab_old_new.py save tools/_build/libold.so   ->  gpurun_out/ab_ref.pt
    python tools/dbg/ab_old_new.py check                          (the in-tree build against the saved outputs)
Cases: pp_color_feat_fwd at the bench workload's shape (160^3, C = 12, 5 / 1 bands) and a generic one (C = 8, 3 / 2 bands)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from poseprobe_amd import _lib
if len(sys.argv) > 2:
    _lib.SO_PATH = os.path.abspath(sys.argv[2])
from poseprobe_amd import ops, synthetic as syn
from poseprobe_amd.engine import SceneConfig

out = {}
for tag, kw in (('bench', {}), ('generic', dict(k0_dim=8, posbase_pe=3, viewbase_pe=2))):
    G = 160 if tag == 'bench' else 40
    try:
        cfg = SceneConfig(syn.XYZ_MIN, syn.XYZ_MAX, G ** 3, out_range=1.0, **kw)
    except TypeError as e:
        print('skip', tag, e); continue
    g = torch.Generator().manual_seed(3)
    M, cap, R = 50001, 60000, 1024
    X, Y, Z = cfg.world_size
    k0 = (torch.randn(X, Y, Z, cfg.k0_dim, generator=g) * 0.1).cuda()
    lo, hi = torch.tensor(syn.XYZ_MIN), torch.tensor(syn.XYZ_MAX)
    pts = (lo + (hi - lo) * (torch.rand(cap, 3, generator=g) * 1.1 - 0.05)).cuda()          # some samples outside the box
    vd = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1).cuda()
    rid = torch.randint(0, R, (cap,), generator=g, dtype=torch.int32).cuda()
    grad = torch.randn(cap, 3, generator=g).cuda()
    pe_w = torch.rand(cfg.posbase_pe + cfg.viewbase_pe, generator=g).cuda()
    count = torch.tensor([M], dtype=torch.int32).cuda()
    feat = torch.full((cap, 64), 7.0).cuda()
    ops.color_feat_fwd(cfg.pp, k0, pts, vd, rid, grad, pe_w, count, cap, feat)
    torch.cuda.synchronize()
    out[tag] = feat.cpu()
    fg = torch.randn(cap, 64, generator=g).cuda()
    pg, gg, vg = torch.zeros(cap, 3).cuda(), torch.zeros(cap, 3).cuda(), torch.zeros(cap, 3).cuda()
    ops.color_feat_bwd(cfg.pp, k0, pts, vd, rid, grad, pe_w, count, cap, fg, None, pg, gg, vg)
    torch.cuda.synchronize()
    out[tag + '_bwd'] = torch.cat([pg[:M], gg[:M], vg[:M]], 1).cpu()
if sys.argv[1] == 'save':
    torch.save(out, 'gpurun_out/ab_ref.pt')
    print('saved', list(out))
else:
    ref = torch.load('gpurun_out/ab_ref.pt', weights_only=True)
    bad = 0
    for k, v in out.items():
        same = torch.equal(v, ref[k])
        nd = int((v != ref[k]).sum())
        print(k, 'bit-identical' if same else f'{nd} elements differ, max |d| {float((v - ref[k]).abs().max()):.3e}')
        bad += not same
    sys.exit(bad)
