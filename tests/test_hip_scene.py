"""Scene branch (poseprobe_amd.bg_nerf -> csrc/pp_nerf.hip through the C ABI) against
  * tests/golden/scene_b2.npz, produced by executing the reference's frequency_nerf.NeRF (oracle/make_golden.py gen_scene),
  * the CPU oracle (oracle/scene_nerf.py) on larger seeded problems,
  * size-independent properties at the full training size (3072 rays x 128 samples).
fp32 tolerances: outputs 2e-5 relative (+2e-6 absolute); gradients 1e-4 relative + 2e-5 of the tensor's largest magnitude
(sums over up to 4e5 samples in a different order than torch's)."""
import numpy as np
import pytest
import torch

from tests.helpers import assert_close, assert_mostly_close, load

pytestmark = pytest.mark.gpu


def _net(d=None, progress=0.565, **kw):
    from poseprobe_amd import bg_nerf
    opt = bg_nerf.default_options(**kw)
    net = bg_nerf.NeRF(opt, device='cuda')
    net.progress.data.fill_(progress)
    if d is not None:
        sd = {k[6:]: torch.tensor(v) for k, v in d.items() if k.startswith('param.')}
        sd['progress'] = torch.tensor(float(d['progress']))
        net.load_state_dict(sd)
    return net, opt


def _oracle_params(net):
    return {k: v.detach().cpu().clone().requires_grad_(True) for k, v in net.state_dict().items() if k != 'progress'}


def test_scene_matches_reference_outputs_and_backward():
    d = load('scene_b2.npz')
    net, opt = _net(d)
    B, N, S = int(d['B']), int(d['N']), int(d['S'])
    center = torch.tensor(d['center']).cuda().requires_grad_(True)
    ray = torch.tensor(d['ray']).cuda().requires_grad_(True)
    depth = torch.tensor(d['depth_samples']).cuda()
    pred = net.forward_samples(opt, center, ray, depth, mode='train')
    pred = net.composite(opt, ray, pred, depth)
    for k in ('rgb_samples', 'density_samples', 'rgb', 'rgb_var', 'depth', 'depth_var', 'opacity', 'weights', 'all_cumulated'):
        assert tuple(pred[k].shape) == tuple(d['out.' + k].shape), k
        assert_close(pred[k], d['out.' + k], rtol=2e-5, atol=2e-6, name=k)
    total = 0.
    for k in ('rgb', 'depth', 'opacity', 'weights', 'rgb_samples', 'density_samples'):
        total = total + (torch.tensor(d['lf_coef_' + k]).cuda() * pred[k]).sum()
    assert_close(total, d['lf_value'], rtol=1e-5, atol=1e-4, name='lf_value')
    total.backward()
    assert_close(center.grad, d['lf_g_center'], rtol=1e-4, scaled=2e-5, name='g_center')
    assert_close(ray.grad, d['lf_g_ray'], rtol=1e-4, scaled=2e-5, name='g_ray')
    for name, p in net.named_parameters():
        if name == 'progress':
            continue
        assert_close(p.grad, d['lf_g.' + name], rtol=1e-4, scaled=2e-5, name='g.' + name)


def test_scene_pose_gradient_through_torch_camera_chain():
    """d loss / d pose: the kernel's ray gradients continue through the (tiny, per-ray) torch camera algebra."""
    d = load('scene_b2.npz')
    net, opt = _net(d)
    pose = torch.tensor(d['pose']).cuda().requires_grad_(True)
    intr, pix = torch.tensor(d['intr']).cuda(), torch.tensor(d['pixels']).cuda()

    def rays_of(pose):
        hom = torch.cat([pix, torch.ones_like(pix[..., :1])], -1)
        cam = hom @ intr.inverse().transpose(-1, -2)
        Rinv = pose[:, :, :3].transpose(-1, -2)
        c2w = torch.cat([Rinv, -Rinv @ pose[:, :, 3:]], -1)
        to_world = lambda X: torch.cat([X, torch.ones_like(X[..., :1])], -1) @ c2w.transpose(-1, -2)
        center = to_world(torch.zeros_like(cam))
        return center, to_world(cam) - center

    center, ray = rays_of(pose)
    assert_close(center, d['center'], rtol=1e-5, atol=1e-6, name='center')
    assert_close(ray, d['ray'], rtol=1e-5, atol=1e-6, name='ray')
    depth = torch.tensor(d['depth_samples']).cuda()
    pred = net.composite(opt, ray, net.forward_samples(opt, center, ray, depth), depth)
    total = 0.
    for k in ('rgb', 'depth', 'opacity', 'weights', 'rgb_samples', 'density_samples'):
        total = total + (torch.tensor(d['lf_coef_' + k]).cuda() * pred[k]).sum()
    total.backward()
    # 24 rays' gradients (|g_ray| up to ~1e3 through the 512*pi band) cancel down to |g_pose| ~ 1e2: their 2e-5 relative
    # rounding differences are amplified accordingly, hence the budget relative to the largest entry
    assert_close(pose.grad, d['lf_g_pose'], rtol=1e-3, scaled=1e-2, name='g_pose')
    # the camera chain itself, fed with the reference's own ray gradients, reproduces the reference's pose gradient
    pose.grad = None
    center2, ray2 = rays_of(pose)
    ((center2 * torch.tensor(d['lf_g_center']).cuda()).sum() + (ray2 * torch.tensor(d['lf_g_ray']).cuda()).sum()).backward()
    assert_close(pose.grad, d['lf_g_pose'], rtol=1e-4, scaled=1e-5, name='g_pose (chain)')


OUT_LD = (256, 256, 256, 320, 256, 256, 256, 288)          # row strides of the stored activations (csrc/pp_nerf.hip nerf_acts)


def hip_relu_states(net, R, S):
    """The ReLU state (activation > 0) of the nine hidden layers of the LAST no_grad forward pass of shape (R, S), read from
    the activation block the kernels keep for their backward pass (layout: csrc/pp_nerf.hip nerf_acts)."""
    acts = net._workspace(R, S).acts
    M = R * S
    off, states = M * 64, []
    for ld in OUT_LD:
        states.append(acts[off:off + M * ld].view(M, ld)[:, :256] > 0)
        off += M * ld
    states.append(acts[off:off + M * 128].view(M, 128) > 0)
    return [t.cpu() for t in states]


@pytest.mark.parametrize('R,S,white', [(96, 40, False), (64, 128, True), (37, 200, False)])
def test_scene_matches_oracle_on_seeded_rays(R, S, white):
    """Outputs against the oracle, then gradients TIGHTLY on the common linear piece: two fp32 implementations of a ReLU
    network differ in the state of the few pre-activations that lie within rounding distance of zero (counted here: the
    one-bit states of both paths are compared), and each such flip changes one sample's contribution discretely.  The oracle
    is therefore differentiated a second time with the kernels' own ReLU states imposed (oracle/scene_nerf.py `masks`):
    every ray gradient and every weight gradient then has to agree to 1e-4 relative + 1e-3 of the tensor's largest entry -
    no outlier allowance."""
    from oracle import scene_nerf as SN
    from poseprobe_amd import bg_nerf
    net, opt = _net(progress=0.61, white_bg=white)
    g = torch.Generator().manual_seed(R * 1000 + S)
    with torch.no_grad():
        for lin in list(net.mlp_feat) + list(net.mlp_rgb):
            lin.bias.copy_(0.05 * torch.randn(lin.bias.shape, generator=g))
    center = (torch.randn(R, 3, generator=g) * 0.3).requires_grad_(True)
    ray = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1) * (0.7 + torch.rand(R, 1, generator=g))
    ray.requires_grad_(True)
    depth = ((torch.rand(R, S, generator=g) + torch.arange(S)) / S * 2.0 + 0.4)
    image = torch.rand(R, 3, generator=g)
    P = _oracle_params(net)
    hidden = []
    out = SN.render(P, center, ray, depth, 0.61, tuple(opt.barf_c2f), white_bg=white, hidden=hidden)
    functional = lambda o, dev: (SN.photometric_loss(o['rgb'].reshape(R, 3), image.to(dev)) + 0.1 * o['depth'].mean()
                                 + 0.05 * (o['weights'] ** 2).sum())
    loss = functional(out, 'cpu')

    c, r = center.detach().cuda().requires_grad_(True), ray.detach().cuda().requires_grad_(True)
    dd = depth.cuda()[None, :, :, None]
    with torch.no_grad():
        net.forward_samples(opt, c[None], r[None], dd)
    states = hip_relu_states(net, R, S)
    flips = [int((st.reshape(-1) != (h.detach().reshape(-1, h.shape[-1]) > 0).reshape(-1)).sum()) for st, h in zip(states, hidden)]
    n_act = sum(st.numel() for st in states)
    assert sum(flips) <= max(4, 4e-6 * n_act), f'{flips} ReLU states differ out of {n_act}'      # expected ~6e-7 x activations
    pred = net.composite(opt, r[None], net.forward_samples(opt, c[None], r[None], dd), dd)
    for k in ('rgb', 'depth', 'opacity', 'weights', 'all_cumulated', 'rgb_var', 'depth_var'):
        assert_close(pred[k].reshape(-1), out[k].reshape(-1), rtol=5e-5, atol=5e-6, name=k)
    l2 = functional(pred, 'cuda')
    assert_close(l2, loss, rtol=2e-5, name='loss')
    l2.backward()
    # the oracle on the kernels' linear piece
    for t in list(P.values()) + [center, ray]:
        t.grad = None
    out_m = SN.render(P, center, ray, depth, 0.61, tuple(opt.barf_c2f), white_bg=white, masks=states)
    functional(out_m, 'cpu').backward()
    assert_close(c.grad, center.grad, rtol=1e-4, scaled=1e-3, name='g_center')
    assert_close(r.grad, ray.grad, rtol=1e-4, scaled=1e-3, name='g_ray')
    for name, p in net.named_parameters():
        if name != 'progress':
            assert_close(p.grad, P[name].grad, rtol=1e-4, scaled=1e-3, name='g.' + name)


def test_scene_split_products_keep_small_magnitude_operands():
    """The three-product scheme scales a tensor by ONE power of two (largest magnitude -> [2^14, 2^15)): entries far below the
    maximum keep fewer bits in the fp16 lo part.  Activations spanning more than 2^18 in magnitude: one row of rays is scaled
    so that its encoded points are ~2^-22 of the others' - its outputs and gradients must still match the oracle to the
    tolerance of the other rays (the dropped bits sit below the fp32 rounding of the sums they enter)."""
    from oracle import scene_nerf as SN
    net, opt = _net(progress=0.9)
    R, S = 48, 32
    g = torch.Generator().manual_seed(77)
    center = torch.randn(R, 3, generator=g) * 0.3
    ray = torch.randn(R, 3, generator=g)
    center[::4] *= 2.0 ** -22                                  # a quarter of the rays: positions (and the linear part of their encoding)
    ray[::4] *= 2.0 ** -22                                     # 2^22 times smaller than the rest
    depth = (torch.rand(R, S, generator=g) + torch.arange(S)) / S * 2.0 + 0.4
    with torch.no_grad():                                      # first-layer weights that let the raw coordinates matter
        net.mlp_feat[0].weight[:, :3] *= 64.0
    P = _oracle_params(net)
    out = SN.render(P, center, ray, depth, 0.9, tuple(opt.barf_c2f))
    dd = depth.cuda()[None, :, :, None]
    pred = net.composite(opt, ray.cuda()[None], net.forward_samples(opt, center.cuda()[None], ray.cuda()[None], dd), dd)
    small = torch.zeros(R, dtype=torch.bool)
    small[::4] = True
    for k in ('rgb', 'depth', 'opacity'):
        a, b = pred[k].reshape(R, -1).detach().cpu(), out[k].reshape(R, -1).detach()
        assert_close(a[small], b[small], rtol=5e-5, atol=5e-6, name=k + ' (small-magnitude rays)')
        assert_close(a[~small], b[~small], rtol=5e-5, atol=5e-6, name=k)
    a, b = pred['rgb_samples'].reshape(R, S, 3).detach().cpu(), out['rgb_samples'].detach()
    assert_close(a[small], b[small], rtol=5e-5, atol=5e-6, name='rgb_samples (small-magnitude rays)')
    # the dynamic range the GEMM operands really span
    enc_small = float(center[small].abs().max()) + float(ray[small].abs().max()) * 2.4
    assert enc_small * 2 ** 18 < 1.0


def test_scene_forward_on_points_equals_forward_samples():
    net, opt = _net(progress=0.9)
    g = torch.Generator().manual_seed(5)
    center, ray = torch.randn(1, 20, 3, generator=g).cuda() * 0.2, torch.randn(1, 20, 3, generator=g).cuda()
    depth = (torch.rand(1, 20, 8, 1, generator=g) * 2 + 0.3).cuda()
    a = net.forward_samples(opt, center, ray, depth)
    pts = center[:, :, None] + ray[:, :, None] * depth
    b = net.forward(opt, pts, ray)
    assert_close(b['rgb_samples'], a['rgb_samples'].detach().cpu(), rtol=1e-5, atol=1e-6, name='rgb_samples')
    assert_close(b['density_samples'], a['density_samples'].detach().cpu(), rtol=1e-5, atol=1e-6, name='density')


def test_scene_engine_step_equals_oracle_adam():
    """forward + 2*huber loss + backward + fused Adam == the oracle followed by torch.optim.Adam (lib/utils.py:294-296)."""
    from oracle import scene_nerf as SN
    from poseprobe_amd import bg_nerf
    net, opt = _net(progress=0.5)
    R, S = 128, 32
    g = torch.Generator().manual_seed(11)
    center, ray = torch.randn(R, 3, generator=g) * 0.3, torch.randn(R, 3, generator=g)
    depth = (torch.rand(R, S, generator=g) + torch.arange(S)) / S * 2.0 + 0.4
    image = torch.rand(R, 3, generator=g)
    P = _oracle_params(net)
    # eps = 1e-4 keeps the update a smooth function of the gradient where |g| ~ 1e-8 (with the default eps = 1e-8 an entry
    # whose gradient is rounding noise moves by a full +-lr, which no two fp32 implementations agree on)
    optim = torch.optim.Adam(list(P.values()), lr=1e-3, betas=(0.9, 0.999), eps=1e-4)
    eng = bg_nerf.SceneEngine(net, lr=1e-3, eps=1e-4)
    for it in range(3):
        optim.zero_grad()
        loss = SN.photometric_loss(SN.render(P, center, ray, depth, 0.5, tuple(opt.barf_c2f))['rgb'], image)
        loss.backward()
        optim.step()
        l2, g_center, g_ray = eng.step(center.cuda(), ray.cuda(), depth.cuda(), image.cuda())
        assert_close(l2, loss, rtol=2e-5 if it == 0 else 5e-4, name=f'loss[{it}]')   # later steps inherit Adam's sensitivity
    for name, p in net.named_parameters():
        if name != 'progress':
            assert_close(p, P[name], rtol=1e-4, atol=6e-4, name='adam.' + name)          # 20 % of the three lr-sized steps taken
    # padding of the packed block never moves
    o = net._off
    assert float(net.flat[o[0]:o[0] + 256 * 64].view(256, 64)[:, 63].abs().max()) == 0.0
    assert float(net.flat[o[18]:o[18] + 128 * 288].view(128, 288)[:, 283:].abs().max()) == 0.0


def test_scene_full_size_properties():
    """3072 rays x 128 samples (3 views x 1024 rays, default_config.py:114,:256): compositing identities and linearity of
    the backward pass in the upstream gradient."""
    from poseprobe_amd import bg_nerf
    net, opt = _net(progress=0.8)
    R, S = 3072, 128
    g = torch.Generator().manual_seed(3)
    center = (torch.randn(R, 3, generator=g) * 0.3).cuda()
    ray = torch.randn(R, 3, generator=g).cuda()
    depth = ((torch.rand(R, S, generator=g) + torch.arange(S)) / S * 2.0 + 0.4).cuda()
    c, r = center[None].clone().requires_grad_(True), ray[None].clone().requires_grad_(True)
    dd = depth[None, :, :, None]
    pred = net.composite(opt, r, net.forward_samples(opt, c, r, dd), dd)
    w = pred['weights'][0, :, :, 0]
    assert torch.isfinite(pred['rgb']).all() and (w >= 0).all()
    assert_close(w.sum(-1), pred['opacity'].reshape(-1).detach().cpu(), rtol=1e-5, atol=1e-6, name='sum(w) = opacity')
    assert float(pred['opacity'].detach().max()) <= 1 + 1e-5            # the last interval is 1e10 long: everything left is absorbed
    assert_close(pred['opacity'].reshape(-1), np.ones(R, np.float32), rtol=1e-5, name='opaque last bin')
    assert_close((w * depth).sum(-1), pred['depth'].reshape(-1).detach().cpu(), rtol=1e-5, atol=1e-6, name='depth')
    coef = torch.randn(R, 3, generator=g).cuda()
    (pred['rgb'][0] * coef).sum().backward()
    g1 = [p.grad.clone() for n, p in net.named_parameters() if n != 'progress'] + [c.grad.clone(), r.grad.clone()]
    for p in net.parameters():
        p.grad = None
    c.grad = r.grad = None
    pred = net.composite(opt, r, net.forward_samples(opt, c, r, dd), dd)
    (pred['rgb'][0] * (-2.0 * coef)).sum().backward()
    g2 = [p.grad for n, p in net.named_parameters() if n != 'progress'] + [c.grad, r.grad]
    for a, b in zip(g1, g2):
        assert_close(b, (-2.0 * a).cpu(), rtol=1e-3, scaled=1e-4, name='backward linearity')   # atomics: summation order differs


def test_scene_pending_forward_passes_of_one_shape_keep_their_activations():
    """Any number of forward passes of the same (R, S) may be pending before backward, as with the reference's autograd
    (each differentiable pass owns its activation block); no_grad passes in between do not disturb them."""
    net, opt = _net()
    g = torch.Generator().manual_seed(3)
    c1, c2 = (torch.randn(1, 8, 3, generator=g) * 0.2).cuda(), (torch.randn(1, 8, 3, generator=g) * 0.2).cuda()
    r = torch.randn(1, 8, 3, generator=g).cuda()
    d = torch.linspace(0.5, 2, 4).cuda().reshape(1, 1, 4, 1).repeat(1, 8, 1, 1)
    params = [p for n, p in net.named_parameters() if n != 'progress']
    singles = []
    for c in (c1, c2):
        net.zero_grad()
        net.forward_samples(opt, c, r, d)['rgb_samples'].sum().backward()
        singles.append([p.grad.clone() for p in params])
    net.zero_grad()
    a = net.forward_samples(opt, c1, r, d)
    b = net.forward_samples(opt, c2, r, d)                       # same shape, pending together
    with torch.no_grad():
        net.forward_samples(opt, c2 * 3.0, r, d)                 # shares the per-shape scratch block, owns nothing
    (a['rgb_samples'].sum() + b['rgb_samples'].sum()).backward()
    for p, g1, g2 in zip(params, *singles):
        assert_close(p.grad, (g1 + g2).cpu(), rtol=1e-4, scaled=1e-5, name='sum of two pending passes')


def test_scene_engine_pass_between_autograd_forward_and_backward():
    """SceneEngine's own forward / backward pair at the same (R, S) between an autograd forward and its backward."""
    from poseprobe_amd import bg_nerf
    net, opt = _net()
    g = torch.Generator().manual_seed(4)
    R, S = 16, 8
    c, r = (torch.randn(R, 3, generator=g) * 0.2).cuda(), torch.randn(R, 3, generator=g).cuda()
    d = ((torch.rand(R, S, generator=g) + torch.arange(S)) / S * 2 + 0.4).cuda()
    params = [p for n, p in net.named_parameters() if n != 'progress']
    net.zero_grad()
    net.forward_samples(opt, c[None], r[None], d[None, :, :, None])['rgb_samples'].sum().backward()
    ref = [p.grad.clone() for p in params]
    net.zero_grad()
    a = net.forward_samples(opt, c[None], r[None], d[None, :, :, None])
    eng = bg_nerf.SceneEngine(net, lr=0.0)
    eng.forward_backward(c * 2.0, r, d, torch.rand(R, 3, generator=g).cuda())
    a['rgb_samples'].sum().backward()
    for p, g0 in zip(params, ref):
        assert_close(p.grad, g0.cpu(), rtol=1e-4, scaled=1e-5, name='autograd pass around an engine pass')


def test_scene_unsupported_architecture_is_refused():
    from poseprobe_amd import bg_nerf
    opt = bg_nerf.default_options()
    opt.arch.layers_feat = [None] + [128] * 8
    with pytest.raises(NotImplementedError):
        bg_nerf.NeRF(opt, device='cuda')


@pytest.mark.parametrize('label,it', [('early', 100), ('late', 500)])
def test_scene_renderer_matches_reference_render(label, it):
    """SceneRenderer.render == the reference's Graph.render (executed by oracle/make_golden.py: gen_scene_render) with the
    recorded uniform draws replayed: coarse pass only before ratio_start_fine_sampling_at_x, coarse + fine after it.  The
    fine depth samples come from the coarse weights through a searchsorted, so they are compared like everything else."""
    from poseprobe_amd import bg_nerf
    d = load('scene_render.npz')
    opt = bg_nerf.default_options(sample_intvs=int(d['n_coarse']))
    opt.nerf.sample_intvs_fine, opt.nerf.fine_sampling = int(d['n_fine']), True
    opt.nerf.ratio_start_fine_sampling_at_x, opt.max_iter = 0.3, 1000
    sr = bg_nerf.SceneRenderer(opt, device='cuda')
    for ni, net in enumerate((sr.nerf, sr.nerf_fine)):
        sd = {k[len(f'param{ni}.'):]: torch.tensor(v.astype(np.float32)) for k, v in d.items() if k.startswith(f'param{ni}.')}
        sd['progress'] = torch.tensor(float(d['progress']))
        net.load_state_dict(sd)
    rand = [torch.tensor(d[f'{label}.rand{i}']) for i in range(2 if label == 'late' else 1)]
    ret = sr.render(opt, torch.tensor(d['pose']).cuda(), 32, 32, torch.tensor(d['intr']).cuda(),
                    pixels=torch.tensor(d['pixels']).cuda(), depth_range=[float(x) for x in d['depth_range']], iter=it,
                    mode='train', rand=rand)
    keys = [k[len(label) + 1:] for k in d if k.startswith(label + '.') and 'rand' not in k]
    assert ('rgb_fine' in keys) == (label == 'late') and ('rgb_fine' in ret) == (label == 'late')
    for k in keys:
        assert tuple(ret[k].shape) == tuple(d[f'{label}.{k}'].shape), k
        if k.endswith('_fine') and k != 't_fine':
            # fine sample positions inherit ~1e-6 differences of the coarse weights; the active 128*pi band turns them into
            # ~1e-4 differences of the network outputs at those samples
            assert_close(ret[k], d[f'{label}.{k}'], rtol=2e-3, atol=5e-4, name=k)
        else:
            assert_close(ret[k], d[f'{label}.{k}'], rtol=5e-5, atol=5e-6, name=k)


def test_scene_band_schedule_follows_in_place_progress_updates():
    """The trainer moves the coarse-to-fine window with `progress.data.fill_` (renderer.py:399-402): the next forward must
    see it (no stale cache), and the weights equal the oracle's window."""
    from oracle import scene_nerf as SN
    net, opt = _net(progress=0.45)
    for p in (0.45, 0.58, 0.9):
        net.progress.data.fill_(p)
        w = net.band_weights().cpu()
        ref = torch.cat([SN.band_weights(np.float32(p).item(), tuple(opt.barf_c2f), 10), SN.band_weights(np.float32(p).item(), tuple(opt.barf_c2f), 4)])
        assert_close(w, ref, rtol=1e-5, atol=1e-6, name=f'bands at {p}')


def test_scene_engine_hierarchical_step_equals_autograd_render():
    """SceneEngine(fine=True) == SceneRenderer.render (coarse + fine, draws replayed) + huber(rgb) + huber(rgb_fine) by
    autograd: loss, ray gradients, both networks' gradients; Adam skips the fine network until it has gradients."""
    from poseprobe_amd import bg_nerf
    opt = bg_nerf.default_options(sample_intvs=32)
    opt.nerf.sample_intvs_fine, opt.nerf.fine_sampling = 24, True
    opt.nerf.ratio_start_fine_sampling_at_x, opt.max_iter = 0.3, 1000
    torch.manual_seed(21)
    sr = bg_nerf.SceneRenderer(opt, device='cuda')
    g = torch.Generator().manual_seed(4)
    for net in (sr.nerf, sr.nerf_fine):
        net.progress.data.fill_(0.66)
        with torch.no_grad():
            net.mlp_feat[-1].bias[0] += 1.0
    B, N, S = 2, 48, 32
    center = (torch.randn(B, N, 3, generator=g) * 0.2).cuda().requires_grad_(True)
    ray = torch.randn(B, N, 3, generator=g).cuda().requires_grad_(True)
    image = torch.rand(B, N, 3, generator=g).cuda()
    rand = [torch.rand(B, N, S, 1, generator=g), torch.rand(25, generator=g)]
    lo, hi = 0.5, 2.5
    depth = (rand[0].cuda() + torch.arange(S).cuda()[None, None, :, None]) / S * (hi - lo) + lo

    eng = bg_nerf.SceneEngine(sr.nerf, lr=1e-3, net_fine=sr.nerf_fine)
    l0, _, _ = eng.forward_backward(center.detach().reshape(-1, 3), ray.detach().reshape(-1, 3), depth.reshape(B * N, S), image.reshape(-1, 3))
    before = sr.nerf_fine.flat.clone()
    eng.optimizer_step()
    assert eng.states[0].steps == 1 and eng.states[1].steps == 0 and torch.equal(sr.nerf_fine.flat, before)
    sr.nerf.flat.copy_(sr.nerf.flat)                                  # (parameters moved by one step; both paths below use them)

    loss, g_center, g_ray = eng.forward_backward(center.detach().reshape(-1, 3).contiguous(), ray.detach().reshape(-1, 3).contiguous(),
                                                 depth.reshape(B * N, S).contiguous(), image.reshape(-1, 3),
                                                 fine=True, depth_range=(lo, hi), fine_grid=rand[1])
    grads = [st.grad.clone() for st in eng.states]

    # autograd path on the same (already stepped) parameters, same draws
    pred_c = sr.nerf.forward_samples(opt, center, ray, depth, mode='train')
    pred_c = sr.nerf.composite(opt, ray, pred_c, depth)
    fine_t = bg_nerf.sample_depth_from_pdf(pred_c['weights'][..., 0].detach(), S, 24, (lo, hi), det=False, grid=rand[1])
    depth_f = torch.cat([depth, fine_t], dim=2).sort(dim=2).values
    pred_f = sr.nerf_fine.composite(opt, ray, sr.nerf_fine.forward_samples(opt, center, ray, depth_f, mode='train'), depth_f)
    ref = bg_nerf.photometric_loss(pred_c['rgb'], image) + bg_nerf.photometric_loss(pred_f['rgb'], image)
    ref.backward()
    assert_close(loss, ref, rtol=2e-5, name='loss (coarse + fine)')
    assert_mostly_close(g_center, center.grad.reshape(-1, 3), rtol=1e-4, scaled=5e-5, name='g_center', outlier_frac=0.2)
    assert_mostly_close(g_ray, ray.grad.reshape(-1, 3), rtol=1e-4, scaled=5e-5, name='g_ray', outlier_frac=0.2)
    for net, gflat in zip((sr.nerf, sr.nerf_fine), grads):
        for (name, p), gv in zip([(n, p) for n, p in net.named_parameters() if n != 'progress'], net._views(gflat)):
            assert_close(gv, p.grad, rtol=1e-3, scaled=1e-2, name='g.' + name)
    eng.optimizer_step()
    assert eng.states[0].steps == 2 and eng.states[1].steps == 1 and not torch.equal(sr.nerf_fine.flat, before)


def test_scene_render_by_slices_equals_one_pass():
    """Full-image rendering in ray slices (renderer.py:629-663) == one pass over all pixels (deterministic mid-point samples in
    eval mode); pixel centres are (x + 0.5, y + 0.5), row-major (utils/camera.py:365-368)."""
    from poseprobe_amd import bg_nerf, synthetic as syn
    opt = bg_nerf.default_options(sample_intvs=24)
    opt.nerf.rand_rays = 100                                  # 16 x 12 = 192 pixels -> two slices, the second one ragged
    torch.manual_seed(9)
    sr = bg_nerf.SceneRenderer(opt, device='cuda')
    sr.nerf.progress.data.fill_(1.0)
    H, W = 12, 16
    views = syn.make_views(2, H, W, seed=5)
    pose = torch.tensor(views['w2c'][:, :3, :4]).float().cuda()
    intr = torch.tensor(views['Ks']).float().cuda()
    a = sr.render_by_slices(opt, pose, H, W, intr, depth_range=(0.5, 3.0), mode='eval')
    b = sr.render(opt, pose, H, W, intr, depth_range=(0.5, 3.0), mode='eval')
    assert a['rgb'].shape == (2, H * W, 3)
    for k in ('rgb', 'depth', 'opacity'):
        assert_close(a[k], b[k].detach().cpu(), rtol=1e-6, atol=1e-7, name=k)
    c, r = bg_nerf.get_center_and_ray(pose, H, W, intr)
    c2, r2 = bg_nerf.get_center_and_ray_at_pixels(pose, torch.tensor([[0.5, 0.5], [W - 0.5, H - 0.5]]).cuda(), intr)
    assert_close(r[:, [0, H * W - 1]], r2.cpu(), rtol=1e-6, atol=1e-7, name='corner rays')


@pytest.mark.parametrize('R,S', [(1, 2), (3, 512), (5, 65)])
def test_scene_edge_shapes_match_oracle(R, S):
    """Smallest ray / sample counts, the maximum S the compositing kernels accept (512) and S just above one wave chunk."""
    from oracle import scene_nerf as SN
    net, opt = _net(progress=0.7)
    g = torch.Generator().manual_seed(S)
    center, ray = torch.randn(R, 3, generator=g) * 0.2, torch.randn(R, 3, generator=g)
    depth = ((torch.rand(R, S, generator=g) + torch.arange(S)) / S * 2.0 + 0.4)
    ref = SN.render(_oracle_params(net), center, ray, depth, 0.7, tuple(opt.barf_c2f))
    dd = depth.cuda()[None, :, :, None]
    pred = net.composite(opt, ray.cuda()[None], net.forward_samples(opt, center.cuda()[None], ray.cuda()[None], dd), dd)
    for k in ('rgb', 'depth', 'opacity', 'weights', 'all_cumulated'):
        assert_close(pred[k].reshape(-1), ref[k].reshape(-1), rtol=5e-5, atol=5e-6, name=k)


def test_scene_sample_count_above_the_kernel_limit_is_refused():
    from poseprobe_amd._lib import PoseProbeError
    net, opt = _net()
    c, r = torch.zeros(1, 2, 3).cuda(), torch.ones(1, 2, 3).cuda()
    d = torch.linspace(0.5, 2, 513).cuda().reshape(1, 1, 513, 1).repeat(1, 2, 1, 1)
    with pytest.raises(PoseProbeError, match='bad sizes'):
        net.composite(opt, r, net.forward_samples(opt, c, r, d), d)


def test_scene_inverse_depth_and_empty_medium():
    """`nerf.depth.param = 'inverse'` (default_config.py:111-112) yields 1 / (t + 1e-8) samples; a medium with (numerically)
    tiny density sends (almost) all weight to the last, 1e10-long interval: opacity 1, colour ~ the last sample's, finite gradients."""
    from poseprobe_amd import bg_nerf
    net, opt = _net(progress=1.0)
    opt.nerf.depth.param = 'inverse'
    dep = bg_nerf.sample_depth(opt, 1, 4, 8, (1, 0), mode='eval', device='cuda')
    assert_close(dep[0, 0, :, 0], 1.0 / ((np.arange(8) + 0.5) / 8 * (0 - 1) + 1 + 1e-8), rtol=1e-5, name='inverse depth')
    with torch.no_grad():
        net.mlp_feat[-1].weight[0].zero_()
        net.mlp_feat[-1].bias[0] = -14.0                    # softplus(-14) ~ 8e-7: x 1e10 still saturates the last interval
    c = torch.zeros(1, 4, 3).cuda().requires_grad_(True)
    r = torch.tensor([[[0., 0., 1.]] * 4]).cuda().requires_grad_(True)
    pred = net.composite(opt, r, net.forward_samples(opt, c, r, dep), dep)
    w = pred['weights'][0, :, :, 0]
    assert float(w[:, :-1].abs().max()) < 1e-4 and float((w[:, -1] - 1).abs().max()) < 1e-3
    assert_close(pred['rgb'][0], pred['rgb_samples'][0, :, -1].detach().cpu(), rtol=0, atol=2e-3, name='colour of the last sample')
    assert_close(pred['opacity'].reshape(-1), np.ones(4, np.float32), rtol=1e-6, name='opacity')
    pred['rgb'].sum().backward()
    assert torch.isfinite(c.grad).all() and torch.isfinite(r.grad).all()


def test_fp32_instruction_mode_passes_the_reference_fixture():
    """The default runs the forward / data-gradient products as three fp16 products (csrc/pp_gemm_split.h); PP_NERF_SPLIT=0
    puts them on the fp32 matrix instructions.  The switch is read when the library is loaded, so the reference-fixture tests
    are re-run in a child process with it: same tolerances in both modes."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, PP_NERF_SPLIT='0')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, '-m', 'pytest', '-q', '-x', '-m', 'gpu', 'tests/test_hip_scene.py', '-k',
                        'matches_reference_outputs_and_backward or renderer_matches_reference_render or edge_shapes'],
                       cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert ' passed' in r.stdout


def test_correspondence_loss_through_the_hip_render_path():
    """bg_losses.correspondence_loss: depths rendered by the kernels at matched pixels of a pair (coarse + fine), both
    re-projection directions; equals the explicit composition of its (reference-pinned) pieces and sends gradients to both
    networks and both poses."""
    from poseprobe_amd import bg_losses, bg_nerf, synthetic as syn
    opt = bg_nerf.default_options(sample_intvs=24)
    opt.nerf.fine_sampling, opt.nerf.sample_intvs_fine = True, 16
    opt.nerf.ratio_start_fine_sampling_at_x, opt.max_iter = None, 1000
    opt.update(renderrepro_do_pixel_reprojection_check=False, renderrepro_do_depth_reprojection_check=False,
               renderrepro_pixel_reprojection_thresh=20., renderrepro_depth_reprojection_thresh=0.1, diff_loss_type='huber')
    torch.manual_seed(31)
    sr = bg_nerf.SceneRenderer(opt, device='cuda')
    for net in (sr.nerf, sr.nerf_fine):
        net.progress.data.fill_(0.7)
        with torch.no_grad():
            net.mlp_feat[-1].bias[0] += 2.0
    H, W, N = 32, 48, 40
    views = syn.make_views(2, H, W, seed=4)
    poses = torch.tensor(views['w2c'][:, :3, :4]).float().cuda().requires_grad_(True)
    intr = torch.tensor(views['Ks']).float().cuda()
    g = torch.Generator().manual_seed(8)
    pix_self = (torch.rand(N, 2, generator=g) * torch.tensor([W - 1., H - 1.])).cuda()
    pix_other = (pix_self.cpu() + torch.randn(N, 2, generator=g) * 3).cuda()
    conf = torch.rand(N, 1, generator=g).cuda()
    rand = [torch.rand(2, N, 24, 1, generator=g), torch.rand(17, generator=g)]
    loss, stats, rets = bg_losses.correspondence_loss(sr, opt, poses, intr, pix_self, pix_other, conf, H, W, (0.5, 3.0),
                                                      iteration=500, rand=rand)
    assert 'depth_fine' in rets and torch.isfinite(loss)
    bottom = torch.tensor([[0., 0., 0., 1.]]).cuda()
    T = torch.cat([poses[1], bottom]) @ bg_losses.pose_inverse_4x4(torch.cat([poses[0], bottom]))
    ref = 0.
    for key in ('depth', 'depth_fine'):
        ds, do = rets[key][0].squeeze(-1), rets[key][1].squeeze(-1)
        ref = ref + bg_losses.reprojection_loss(opt, pix_self, ds, intr[0], pix_other, do, intr[1], T, conf)[0]
        ref = ref + bg_losses.reprojection_loss(opt, pix_other, do, intr[1], pix_self, ds, intr[0], bg_losses.pose_inverse_4x4(T), conf)[0]
    assert_close(loss, (ref / 4.).detach().cpu(), rtol=1e-6, name='loss composition')
    loss.backward()
    assert torch.isfinite(poses.grad).all() and float(poses.grad.abs().max()) > 0
    for net in (sr.nerf, sr.nerf_fine):
        gmax = max(float(p.grad.abs().max()) for n, p in net.named_parameters() if n != 'progress')
        assert np.isfinite(gmax) and gmax > 0


def test_scene_state_dict_is_compact_and_round_trips(tmp_path):
    """`state_dict()` hands out compact copies (not views of the 2 MB packed block) under the reference's names and shapes;
    saving and loading it reproduces the network bit for bit."""
    import os
    from poseprobe_amd import bg_nerf
    net, opt = _net(progress=0.42)
    sd = net.state_dict()
    assert sd['mlp_feat.0.weight'].shape == (256, 63) and sd['mlp_feat.4.weight'].shape == (256, 319)
    assert sd['mlp_feat.7.weight'].shape == (257, 256) and sd['mlp_rgb.0.weight'].shape == (128, 283)
    assert all(v.is_contiguous() for v in sd.values())
    path = str(tmp_path / 'nerf.pt')
    torch.save(sd, path)
    assert os.path.getsize(path) < 3.5e6                      # 0.6 M parameters, not 22 copies of the block
    net2 = bg_nerf.NeRF(opt, device='cuda')
    net2.load_state_dict(torch.load(path, map_location='cpu', weights_only=True))
    assert torch.equal(net2.flat, net.flat) and float(net2.progress) == float(net.progress)


def test_depth_consistency_loss_through_the_hip_render_path():
    """bg_losses.depth_consistency_loss (depth_cons_loss.py:128-330): pseudo ground truth from a training view, visibility by
    render_up_to_maxdepth in a virtual view, weighted Huber on the depths rendered there; gradients reach both networks, the
    poses are detached as in the reference; points projected outside the virtual image give the zero loss."""
    from poseprobe_amd import bg_losses, bg_nerf, synthetic as syn
    opt = bg_nerf.default_options(sample_intvs=24)
    opt.nerf.fine_sampling, opt.nerf.sample_intvs_fine = True, 16
    opt.nerf.ratio_start_fine_sampling_at_x, opt.max_iter = 0.3, 1000
    opt.update(diff_loss_type='huber')
    torch.manual_seed(17)
    sr = bg_nerf.SceneRenderer(opt, device='cuda')
    for net in (sr.nerf, sr.nerf_fine):
        net.progress.data.fill_(0.8)
    H, W, N = 32, 48, 64
    views = syn.make_views(3, H, W, seed=9)
    poses = torch.tensor(views['w2c'][:, :3, :4]).float().cuda().requires_grad_(True)
    intr = torch.tensor(views['Ks']).float().cuda()
    g = torch.Generator().manual_seed(2)
    pix = (torch.rand(N, 2, generator=g) * torch.tensor([W - 1., H - 1.])).cuda()
    # render_up_to_maxdepth: S deterministic samples ending exactly at each ray's own far bound
    dmax = (1.0 + torch.rand(1, N, generator=g)).cuda()
    ret = sr.render_up_to_maxdepth(opt, poses[:1].detach(), H, W, intr[:1], dmax, 0.5, pix[None], iter=900, mode='train')
    t = ret['t'][0, :, :, 0]
    assert_close(t[:, -1], dmax[0].cpu(), rtol=1e-6, name='last sample = per-ray far bound')
    assert_close(t[:, 0], (0.5 + (dmax[0] - 0.5) / 24).cpu(), rtol=1e-6, name='first sample')
    assert 'all_cumulated_fine' in ret and float(ret['all_cumulated'].max()) <= 1.0
    loss, stats = bg_losses.depth_consistency_loss(sr, opt, poses, intr, H, W, (0.5, 3.0), iteration=900, id_self=1,
                                                   pixels_ref=pix, w=0.6)
    assert torch.isfinite(loss) and float(loss) > 0 and stats['nbr_px_sampling'] == N
    loss.backward()
    assert poses.grad is None                                   # pseudo ground truth and virtual pose use detached poses
    for net in (sr.nerf, sr.nerf_fine):
        gmax = max(float(p.grad.abs().max()) for n, p in net.named_parameters() if n != 'progress')
        assert np.isfinite(gmax) and gmax > 0
    # w = 1.0: the virtual view IS the reference view, every point stays in bounds, so the reference-view render (grad
    # enabled, depth not detached - as in the reference), the no_grad visibility render and the virtual-view render all run
    # at the same (R, S) before one backward
    for net in (sr.nerf, sr.nerf_fine):
        net.zero_grad()
    loss1, stats1 = bg_losses.depth_consistency_loss(sr, opt, poses, intr, H, W, (0.5, 3.0), iteration=900, id_self=1,
                                                     pixels_ref=pix, w=1.0)
    assert torch.isfinite(loss1) and stats1['nbr_px_sampling'] == N
    # ... summed with a photometric render of the same number of rays and samples (one backward for both)
    ret = sr.render(opt, poses[1:2].detach(), H, W, intr[1:2], pixels=pix, depth_range=(0.5, 3.0), iter=900, mode='train')
    photo = bg_nerf.photometric_loss(ret['rgb'], torch.rand(1, N, 3, generator=g).cuda())
    (loss1 + photo).backward()
    for net in (sr.nerf, sr.nerf_fine):
        gmax = max(float(p.grad.abs().max()) for n, p in net.named_parameters() if n != 'progress')
        assert np.isfinite(gmax) and gmax > 0
    far = torch.tensor([[50., 50., -5.]] * 8).cuda()            # behind / outside the virtual camera: nothing survives
    bottom = torch.tensor([[0., 0., 0., 1.]]).cuda()
    z, _ = bg_losses.depth_consistency_loss_at_pose(sr, opt, torch.cat([poses[0].detach(), bottom]), intr[0], far, H, W, 0.5, 900)
    assert float(z) == 0.0 and z.requires_grad


def test_photometric_loss_kernel_equals_torch_huber():
    """pp_nerf_huber_loss = 2 * F.huber_loss(delta = 0.5, mean) and its gradient (base_losses.py:155-156), both branches, odd sizes."""
    from poseprobe_amd import ops
    g = torch.Generator().manual_seed(5)
    for n in (3, 3069, 40000):
        pred = (torch.rand(n, generator=g) * 2.0 - 0.5).cuda().requires_grad_(True)       # residuals on both sides of delta
        label = torch.rand(n, generator=g).cuda()
        ref = torch.nn.functional.huber_loss(pred, label, reduction='mean', delta=0.5) * 2.
        ref.backward()
        loss, gp = torch.zeros(1, device='cuda'), torch.empty(n, device='cuda')
        ops.nerf_huber_loss(pred.detach(), label, 0.5, 2.0, loss, gp)
        assert_close(loss[0], ref.detach(), rtol=2e-6, name=f'huber[{n}]')
        assert_close(gp, pred.grad, rtol=1e-6, atol=1e-12, name=f'huber grad[{n}]')


def test_band_weight_kernel_equals_the_torch_expression():
    """pp_nerf_band_weights against frequency_nerf.py:250-253 evaluated by torch on the device, before / inside / after the window."""
    import math
    from poseprobe_amd import ops
    out = torch.empty(14, device='cuda')
    for p in (0.0, 0.4, 0.4375, 0.565, 0.7, 1.0):
        prog = torch.tensor(p, device='cuda')
        ops.nerf_band_weights(prog, 0.4, 0.7, 10, 4, out)
        ref = []
        for L in (10, 4):
            alpha = (prog - 0.4) / (0.7 - 0.4) * L
            k = torch.arange(L, dtype=torch.float32, device='cuda')
            ref.append((1 - (alpha - k).clamp_(min=0, max=1).mul_(math.pi).cos_()) / 2)
        assert_close(out, torch.cat(ref), rtol=0, atol=1.2e-7, name=f'bands at {p}')


@pytest.mark.parametrize('nerf_split', [1, 0])
def test_scene_kernels_stay_inside_their_buffers(nerf_split):
    """Fence around the scene GEMM chain ("the memory-access fault of 4 Oct"): pp_nerf_fwd / pp_nerf_bwd
    at 1023 x 128 (the reference batch: 1023 row tiles of 128, every persistent work-group walks four tiles), 1 x 2 (less than
    one tile) and 37 x 50 (1850 rows: a ragged last tile, more tiles than nothing else exercises at this size) with every
    caller-owned buffer the kernels WRITE - the activation block (whose tail holds the ReLU mask planes and the split weight
    images), the backward scratch, both sample outputs, the parameter-gradient block, both ray gradients - embedded in a
    sentinel-filled arena: 64 KB of sentinels in front of and behind each must survive both passes, in both arithmetic modes.
    Run once per build, as every other test (no repetition: an out-of-bounds access is deterministic in the indexing)."""
    from poseprobe_amd import bg_nerf, ops
    dev = 'cuda'
    PAD = 16384                                                     # floats = 64 KB
    SENT = 0x7FC0DEAD                                               # a quiet-NaN payload no kernel produces

    def fenced(n, fill=None):
        arena = torch.empty(n + 2 * PAD, dtype=torch.int32, device=dev).fill_(SENT).view(torch.float32)
        body = arena[PAD:PAD + n]
        if fill is not None:
            body.copy_(fill) if isinstance(fill, torch.Tensor) else body.fill_(fill)
        return arena, body

    def intact(arena, n, what):
        a = arena.view(torch.int32)
        assert bool((a[:PAD] == SENT).all()), f'{what}: sentinels IN FRONT of the buffer were overwritten'
        assert bool((a[PAD + n:] == SENT).all()), f'{what}: sentinels BEHIND the buffer were overwritten'

    opt = bg_nerf.default_options(sample_intvs=128)
    torch.manual_seed(3)
    net = bg_nerf.NeRF(opt, device=dev, options={'nerf_split': nerf_split, 'nerf_split_tn': nerf_split})
    net.progress.data.fill_(0.7)
    g = torch.Generator().manual_seed(21)
    for R, S in ((1023, 128), (1, 2), (37, 50)):
        M = R * S
        center = (torch.randn(R, 3, generator=g) * 0.3).to(dev)
        ray = torch.randn(R, 3, generator=g).to(dev)
        depth = ((torch.rand(R, S, generator=g) + torch.arange(S)) / S * 2.0 + 0.4).to(dev).contiguous()
        count = torch.tensor([M], dtype=torch.int32, device=dev)
        n_acts, n_scr = ops.nerf_workspace(M, R)
        acts_a, acts = fenced(n_acts)
        scr_a, scr = fenced(n_scr, 0.0)
        rgb_a, rgb_s = fenced(M * 3)
        den_a, dens = fenced(M)
        pg_a, pgrad = fenced(net.flat.numel(), 0.0)
        gc_a, g_center = fenced(R * 3)
        gr_a, g_ray = fenced(R * 3)
        ops.nerf_fwd(net.flat, center, ray, depth, net.band_weights(), count, R, S, acts, rgb_s.view(M, 3), dens, net.ctx)
        torch.cuda.synchronize()
        for arena, n, what in ((acts_a, n_acts, 'acts'), (rgb_a, M * 3, 'rgb_samples'), (den_a, M, 'density_samples')):
            intact(arena, n, f'{R}x{S} forward, {what}')
        assert bool(torch.isfinite(rgb_s).all()) and bool(torch.isfinite(dens).all())
        g_rgb = torch.randn(M, 3, generator=g).to(dev)
        g_den = torch.randn(M, generator=g).to(dev)
        ops.nerf_bwd(net.flat, ray, depth, count, R, S, acts, rgb_s.view(M, 3), g_rgb, g_den, scr, pgrad, g_center.view(R, 3),
                     g_ray.view(R, 3), net.ctx)
        torch.cuda.synchronize()
        for arena, n, what in ((acts_a, n_acts, 'acts'), (scr_a, n_scr, 'scratch'), (pg_a, net.flat.numel(), 'params_grad'),
                               (gc_a, R * 3, 'g_center'), (gr_a, R * 3, 'g_ray'), (rgb_a, M * 3, 'rgb_samples'),
                               (den_a, M, 'density_samples')):
            intact(arena, n, f'{R}x{S} backward, {what}')
        assert bool(torch.isfinite(pgrad).all()) and bool(torch.isfinite(g_center).all()) and bool(torch.isfinite(g_ray).all())
        assert float(pgrad.abs().max()) > 0
        del acts_a, scr_a, acts, scr
        torch.cuda.empty_cache()


@pytest.mark.parametrize('R,S,chain,nw,head', [(1023, 128, 3, 4, 1), (1023, 128, 3, 8, 1), (1023, 128, 1, 4, 0), (37, 50, 3, 4, 0), (37, 50, 1, 8, 1),
                                                (1, 2, 3, 4, 1)])
def test_scene_trunk_kernel_equals_layer_by_layer_path(R, S, chain, nw, head):
    """Option nerf_chain (csrc/pp_nerf_trunk.h: the eight feature layers + density head as one kernel, tile resident in LDS)
    against the layer-by-layer GEMMs on the same inputs: every stored activation, the raw density and the sample outputs
    agree to fp32 rounding of a 256..320-term sum (the two paths scale their fp16 operand pairs by different powers of two -
    tile maximum vs tensor maximum - so they are not bit-identical), the one-bit ReLU masks the data-gradient kernels read
    agree wherever the activations of both paths have the same state (and ARE that state), and the tensor maxima recorded
    for the backward pass agree.  Sizes: four tiles per work-group, a ragged last tile, less than one tile.  chain = 1: fused
    forward writing the masks in the layer-by-layer layout; chain = 3: in the fused chains' own layout ([row][wavefront][lane
    half], bit j <-> column 32 w + 4 half + (j & 3) + 8 (j >> 2)), followed by a backward pass of both paths on the same
    upstream gradients (parameter, centre and ray gradients).  nw: wavefronts per work-group of the fused kernels (8 on a
    128-sample tile, one work-group per CU; 4 on a 64-sample tile, two per CU); head: the colour head's hidden layer as the
    forward chain's ninth stage (compared through the sample colours)."""
    from poseprobe_amd import bg_nerf, ops
    dev = 'cuda'
    opt = bg_nerf.default_options(sample_intvs=S)
    torch.manual_seed(5)
    nets = [bg_nerf.NeRF(opt, device=dev, options={'nerf_chain': c, 'nerf_chain_nw': nw, 'nerf_chain_head': head}) for c in (chain, 0)]
    g = torch.Generator().manual_seed(R + S)
    with torch.no_grad():
        for lin in list(nets[0].mlp_feat) + list(nets[0].mlp_rgb):
            lin.bias.copy_(0.05 * torch.randn(lin.bias.shape, generator=g))
        nets[1].flat.data.copy_(nets[0].flat.data)
    for n in nets:
        n.progress.data.fill_(0.65)
    M = R * S
    center = (torch.randn(R, 3, generator=g) * 0.3).to(dev)
    ray = torch.randn(R, 3, generator=g).to(dev)
    depth = ((torch.rand(R, S, generator=g) + torch.arange(S)) / S * 2.0 + 0.4).to(dev).contiguous()
    count = torch.tensor([M], dtype=torch.int32, device=dev)
    n_acts, n_scr = ops.nerf_workspace(M, R)
    res = []
    for n in nets:
        acts = torch.zeros(n_acts, device=dev)
        rgb, dens = torch.empty(M, 3, device=dev), torch.empty(M, device=dev)
        ops.nerf_fwd(n.flat, center, ray, depth, n.band_weights(), count, R, S, acts, rgb, dens, n.ctx)
        torch.cuda.synchronize()
        res.append((acts, rgb, dens))
    (a1, rgb1, d1), (a0, rgb0, d0) = res
    OUT_LD = [256, 256, 256, 320, 256, 256, 256, 288]
    off = M * 64
    assert torch.equal(a1[:off], a0[:off])                          # the encoding is the same kernel
    acts_l = []
    for l, ld in enumerate(OUT_LD):
        x1, x0 = a1[off:off + M * ld].view(M, ld)[:, :256], a0[off:off + M * ld].view(M, ld)[:, :256]
        assert_close(x1, x0, rtol=2e-5, scaled=2e-6, name=f'layer {l} output')
        acts_l.append((x1, x0))
        off += M * ld
    off += M * 128                                                   # colour head hidden layer (compared through rgb)
    assert_close(a1[off:off + M], a0[off:off + M], rtol=2e-5, scaled=2e-6, name='raw density')
    off += M
    mx1, mx0 = a1[off:off + 64], a0[off:off + 64]
    assert_close(mx1[1:9], mx0[1:9], rtol=2e-5, name='recorded layer maxima')
    off += 64
    Mp = (M + 127) // 128 * 128
    for l in range(8):
        b1 = a1[off:off + Mp * 8].view(torch.int32)
        b0 = a0[off:off + Mp * 8].view(torch.int32)
        off += Mp * 8
        # unpack: word [row group of 32][column][half] (16 bits), bit j <-> row 32 g + (j & 3) + 8 (j >> 2) + 4 half
        def states(b):
            wds = torch.stack([b & 0xFFFF, (b >> 16) & 0xFFFF], -1).view(Mp // 32, 256, 2)
            j = torch.arange(16, device=dev)
            bits = ((wds[..., None] >> j) & 1).bool()                 # [g, col, half, j]
            rows = ((j & 3) + 8 * (j >> 2))[None, :] + 4 * torch.arange(2, device=dev)[:, None]      # [half, j]
            out = torch.zeros(Mp // 32, 32, 256, dtype=torch.bool, device=dev)
            out[:, rows.reshape(-1), :] = bits.permute(0, 2, 3, 1).reshape(Mp // 32, 32, 256)
            return out.view(Mp, 256)[:M]
        def states_rows(b):
            wds = torch.stack([b & 0xFFFF, (b >> 16) & 0xFFFF], -1).view(Mp, 8, 2)                   # [row][w][half]
            j = torch.arange(16, device=dev)
            bits = ((wds[..., None] >> j) & 1).bool()                                                  # [row, w, half, j]
            col = (32 * torch.arange(8, device=dev)[:, None, None] + 4 * torch.arange(2, device=dev)[None, :, None]
                   + ((j & 3) + 8 * (j >> 2))[None, None, :]).reshape(-1)
            out = torch.zeros(Mp, 256, dtype=torch.bool, device=dev)
            out[:, col] = bits.reshape(Mp, 256)
            return out[:M]
        s1, s0 = (states_rows(b1) if chain == 3 else states(b1)), states(b0)
        x1, x0 = acts_l[l]
        assert torch.equal(s1, x1 > 0), f'layer {l}: mask bits of the fused kernel are not the states of its own activations'
        assert torch.equal(s0, x0 > 0)
        assert int((s1 != s0).sum()) <= max(2, 2e-6 * s1.numel()), f'layer {l}: {int((s1 != s0).sum())} states differ'
    assert_close(d1, d0, rtol=2e-5, scaled=2e-6, name='density')
    assert_close(rgb1, rgb0, rtol=0, atol=2e-6, name='rgb samples')
    if chain != 3:
        return
    # backward of both paths on the SAME forward state: the fused forward's activation block, with its mask planes re-packed
    # into the layer-by-layer layout for the layer-by-layer backward - identical ReLU states, so the gradients must agree
    # to rounding (no allowance for flipped states)
    def pack_cols(st):                                                # [M,256] bool -> int32 words of the layer-by-layer layout
        full = torch.zeros(Mp, 256, dtype=torch.bool, device=dev)
        full[:M] = st
        j = torch.arange(16, device=dev)
        rows = ((j & 3) + 8 * (j >> 2))[None, :] + 4 * torch.arange(2, device=dev)[:, None]          # [half, j]
        x = full.view(Mp // 32, 32, 256)[:, rows.reshape(-1), :].view(Mp // 32, 2, 16, 256)          # [g, half, j, col]
        wds = (x.to(torch.int64) << j[None, None, :, None]).sum(2)                                    # [g, half, col]
        return (wds[:, 0] | (wds[:, 1] << 16)).to(torch.int64).view(-1)                              # [g * 256 + col]
    a_conv = a1.clone()
    off_b = M * (64 + sum(OUT_LD) + 128 + 1) + 64
    for l in range(8):
        words = pack_cols(states_rows(a1[off_b:off_b + Mp * 8].view(torch.int32)))
        a_conv[off_b:off_b + Mp * 8] = ((words + 2 ** 31) % 2 ** 32 - 2 ** 31).to(torch.int32).view(torch.float32)   # same 32 bits
        off_b += Mp * 8
    g_rgb = torch.randn(M, 3, generator=g).to(dev)
    g_den = torch.randn(M, generator=g).to(dev)
    grads = []
    for n, acts in ((nets[0], a1), (nets[1], a_conv)):
        scr = torch.zeros(n_scr, device=dev)
        pg = torch.zeros_like(n.flat)
        gc, gr = torch.empty(R, 3, device=dev), torch.empty(R, 3, device=dev)
        ops.nerf_bwd(n.flat, ray, depth, count, R, S, acts, rgb1, g_rgb, g_den, scr, pg, gc, gr, n.ctx)
        torch.cuda.synchronize()
        grads.append((pg, gc, gr))
    (pg1, gc1, gr1), (pg0, gc0, gr0) = grads
    assert_close(pg1, pg0, rtol=2e-5, scaled=2e-5, name='parameter gradients')
    assert_close(gc1, gc0, rtol=2e-5, scaled=2e-5, name='g_center')
    assert_close(gr1, gr0, rtol=2e-5, scaled=2e-5, name='g_ray')
