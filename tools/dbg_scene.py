import sys, torch, numpy as np
sys.path.insert(0, '.')
from oracle import scene_nerf as SN
from poseprobe_amd import bg_nerf
torch.manual_seed(0)
opt = bg_nerf.default_options()
net = bg_nerf.NeRF(opt, device='cuda'); net.progress.data.fill_(0.61)
P = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in net.state_dict().items() if k != 'progress'}
for R, S in [(96, 40), (96, 64), (67, 40), (64, 40), (30, 128), (128, 30)]:
    g = torch.Generator().manual_seed(1)
    center = (torch.randn(R, 3, generator=g) * 0.3).requires_grad_(True)
    ray = torch.randn(R, 3, generator=g).requires_grad_(True)
    depth = ((torch.rand(R, S, generator=g) + torch.arange(S)) / S * 2.0 + 0.4)
    coef = torch.randn(R, 3, generator=g)
    for p in P.values(): p.grad = None
    out = SN.render(P, center, ray, depth, 0.61, tuple(opt.barf_c2f))
    (out['rgb'] * coef).sum().backward()
    c, r = center.detach().cuda().requires_grad_(True), ray.detach().cuda().requires_grad_(True)
    dd = depth.cuda()[None, :, :, None]
    for p in net.parameters(): p.grad = None
    pred = net.composite(opt, r[None], net.forward_samples(opt, c[None], r[None], dd), dd)
    (pred['rgb'][0] * coef.cuda()).sum().backward()
    ec = (c.grad.cpu() - center.grad).abs().max(1).values
    er = (r.grad.cpu() - ray.grad).abs().max(1).values
    bad = (ec > 1e-2).nonzero().flatten().tolist()
    print(R, S, 'M', R * S, 'bad center rays', bad[:5], '...', bad[-3:], len(bad), 'max ec', float(ec.max()), 'max er', float(er.max()),
          'ref max', float(center.grad.abs().max()))
    for n, p in net.named_parameters():
        if n != 'progress':
            e = float((p.grad.cpu() - P[n].grad).abs().max()); m = float(P[n].grad.abs().max())
            if e > 1e-3 * m: print('   ', n, 'err', e, 'max', m)
