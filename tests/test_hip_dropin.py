"""GPU parity tests through the DROP-IN module surface (poseprobe_amd.voxurf_coarse mirrors lib/voxurf_coarse.py):
constructed with the reference's kwargs, loaded through the reference's state_dict names, driven like
recon_scene.optimize_increamental drives it, compared with golden vectors produced by the reference."""
import numpy as np
import pytest
import torch

from tests.helpers import assert_close, load

pytestmark = pytest.mark.gpu


def make_model(d):
    from poseprobe_amd import synthetic as syn
    from poseprobe_amd import voxurf_coarse as Model
    G, H, W = int(d['G']), int(d.get('H', 32)), int(d.get('W', 32))
    rs = syn.range_shape()
    m = Model.Voxurf(syn.XYZ_MIN, syn.XYZ_MAX, num_voxels=G ** 3, num_voxels_base=G ** 3, alpha_init=1e-2,
                     rgbnet_dim=12, rgbnet_direct=True, rgbnet_depth=4, rgbnet_width=128, posbase_pe=5, viewbase_pe=1,
                     geo_rgb_dim=3, s_ratio=50, s_start=0.2, barf_c2f=[0.6, 1], i_train=np.arange(3), N_iters=10000,
                     HW=np.array([[H, W]] * 3), range_shape=rs, rect_size=rs.tolist(), camera_noise=0.,
                     some_unknown_cfg_key=1)
    sd = m.state_dict()
    # same names as the reference's state_dict
    expect = {'progress', 'sdf_alpha', 'sdf_beta', 'xyz_min', 'xyz_max', 'posfreq', 'viewfreq', 'warp_network.progress',
              'sdf.grid', 'sdf.xyz_min', 'sdf.xyz_max', 'k0.grid', 'k0.xyz_min', 'k0.xyz_max', 'rgbnet.0.weight',
              'rgbnet.0.bias', 'rgbnet.2.0.weight', 'rgbnet.2.0.bias', 'rgbnet.3.0.weight', 'rgbnet.3.0.bias',
              'rgbnet.4.weight', 'rgbnet.4.bias', 'grad_conv.weight', 'grad_conv.bias', 'tv_smooth_conv.weight',
              'tv_smooth_conv.bias'} | {f'warp_network.deform_net.net.net.{i}.0.{k}' for i in range(5) for k in ('weight', 'bias')}
    assert set(sd.keys()) == expect, set(sd.keys()) ^ expect
    assert np.array_equal(sd['sdf.grid'].numpy(), d['P.sdf']), 'cube-init template differs from the reference'
    sd['k0.grid'] = torch.tensor(d['P.k0'])
    sd['sdf_alpha'], sd['sdf_beta'] = torch.tensor(d['P.sdf_alpha']), torch.tensor(d['P.sdf_beta'])
    for li, key in enumerate(['rgbnet.0', 'rgbnet.2.0', 'rgbnet.3.0', 'rgbnet.4']):
        sd[key + '.weight'], sd[key + '.bias'] = torch.tensor(d[f'P.rgbnet.{li}.weight']), torch.tensor(d[f'P.rgbnet.{li}.bias'])
    for li in range(5):
        sd[f'warp_network.deform_net.net.net.{li}.0.weight'] = torch.tensor(d[f'P.warp.{li}.weight'])
        sd[f'warp_network.deform_net.net.net.{li}.0.bias'] = torch.tensor(d[f'P.warp.{li}.bias'])
    m.load_state_dict(sd)
    return m.cuda()


@pytest.mark.parametrize('tag', ['g8_s10', 'g24_s7000'])
def test_voxurf_forward_backward_like_the_reference_trainer(tag):
    from poseprobe_amd import camera
    from poseprobe_amd import voxurf_coarse as Model
    from poseprobe_amd.losses import _AttrDict, object_losses
    d = load(f'forward_{tag}.npz')
    m = make_model(d)
    pm = Model.pose_model(i_train=np.arange(3), camera_noise=0.).cuda()
    pm.se3_refine.data.copy_(torch.tensor(d['se3']))
    w2c, c2w = camera.current_pose_c2w(pm.se3_refine, torch.tensor(d['w2c_init']).cuda())
    assert_close(w2c.detach().cpu(), d['w2c'], rtol=1e-6, atol=1e-6, name='w2c')
    H, W = int(d['H']), int(d['W'])
    imgs, msks = torch.tensor(d['images']).cuda(), torch.tensor(d['masks']).cuda()
    target, mask, ro, rd, vd = Model.select_training_rays(torch.tensor(d['ray_idx']), imgs, msks, c2w,
                                                          np.array([[H, W]] * 3), d['Ks'])
    gs = int(d['global_step'])
    out = m(ro, rd, vd, use_deform=True, global_step=gs, near=0.24, far=4.8, bg=0, stepsize=1.5, inverse_y=True,
            flip_x=False, flip_y=False, jitter=torch.tensor(d['jitter']))
    c = lambda t: t.detach().cpu().numpy()
    assert np.array_equal(c(out['mask']), d['out.mask'])
    for k in ('alphainv_cum', 'weights', 'cum_weights', 'rgb_marched', 'raw_alpha', 'raw_rgb', 'depth', 'disp',
              'gradient', 'k0_tv', 'sdf_deform', 'grad_deform', 'sdf_correct'):
        assert out[k].shape == d['out.' + k].shape, (k, out[k].shape, d['out.' + k].shape)
        assert_close(c(out[k]), d['out.' + k], rtol=1e-4, atol=1e-5, scaled=1e-6, name=k)
    assert abs(out['s_val'] - float(d['out.s_val'])) < 1e-12
    cfg_train = _AttrDict(weight_main=1., weight_tv_k0=.01, weight_mask=.1)
    S, Wt, loss = object_losses(out, cfg_train, target, mask, gs, 10000, True)
    assert_close(c(loss), d['loss'], rtol=1e-4, name='loss')
    (loss * 0.1).backward()
    tol = dict(rtol=1e-3, scaled=2e-5)
    assert_close(c(pm.se3_refine.grad), d['grad.se3'], atol=1e-6, name='g.se3', **tol)
    assert m.k0.grid.grad.shape == d['grad.k0'].shape
    assert_close(c(m.k0.grid.grad), d['grad.k0'], atol=1e-9, name='g.k0', **tol)
    assert_close(c(m.sdf_alpha.grad), d['grad.sdf_alpha'], atol=1e-7, name='g.sdf_alpha', **tol)
    assert_close(c(m.sdf_beta.grad), d['grad.sdf_beta'], atol=1e-7, name='g.sdf_beta', **tol)
    rg = [m.rgbnet[0], m.rgbnet[2][0], m.rgbnet[3][0], m.rgbnet[4]]
    for li, lin in enumerate(rg):
        assert_close(c(lin.weight.grad), d[f'grad.rgbnet.{li}.weight'], atol=1e-8, name=f'g.rgbnet{li}.W', **tol)
        assert_close(c(lin.bias.grad), d[f'grad.rgbnet.{li}.bias'], atol=1e-8, name=f'g.rgbnet{li}.b', **tol)
    for li, lin in enumerate(m.warp_network.linears()):
        assert_close(c(lin.weight.grad), d[f'grad.warp.{li}.weight'], atol=2e-7, name=f'g.warp{li}.W', **tol)
        assert_close(c(lin.bias.grad), d[f'grad.warp.{li}.bias'], atol=2e-7, name=f'g.warp{li}.b', **tol)
    assert m.sdf.grid.grad is None      # frozen template


def test_voxurf_inference_matches_reference():
    d = load('inference_g24.npz')
    m = make_model(d)
    ro, rd, vd = (torch.tensor(d[k]).cuda() for k in ('rays_o', 'rays_d', 'viewdirs'))
    out = m.inference(ro, rd, vd, global_step=None, near=0.24, far=4.8, bg=0, stepsize=1.5, inverse_y=True,
                      flip_x=False, flip_y=False)
    c = lambda t: t.detach().cpu().numpy()
    M = d['out.weights'].shape[0]
    assert out['weights'].shape[0] == M, 'variable-length sampler kept a different number of samples'
    for k in ('alphainv_cum', 'weights', 'cum_weights', 'rgb_marched', 'normal_marched', 'raw_alpha', 'raw_rgb',
              'depth', 'gradient', 'gradient_error'):
        assert_close(c(out[k]), d['out.' + k], rtol=1e-4, atol=1e-5, scaled=1e-6, name=k)


def test_rays_of_a_view_and_alphas2weights_api():
    from poseprobe_amd import voxurf_coarse as Model
    d = load('rays.npz')
    H, W = 8, 12
    for v in range(3):
        o, dd, vd = Model.get_rays_of_a_view(H, W, torch.tensor(d['Ks'][v]), torch.tensor(d['c2w'][v]).cuda(), ndc=False,
                                             inverse_y=True, flip_x=False, flip_y=False)
        assert np.array_equal(dd.cpu().numpy(), d[f'vox_d_v{v}_invy1'])
        assert np.array_equal(o.cpu().numpy(), d[f'vox_o_v{v}_invy1'])
    alpha = torch.tensor([0.5, 0.5, 0.9999, 0.3, 0.2, 0.25], device='cuda', requires_grad=True)
    ray_id = torch.tensor([0, 0, 2, 2, 2, 3], device='cuda')
    w, last = Model.Alphas2Weights.apply(alpha, ray_id, 5)
    assert_close(w.detach().cpu(), [0.5, 0.25, 0.9999, 0.0, 0.0, 0.25], rtol=1e-6, atol=1e-7)
    (w * torch.arange(1., 7., device='cuda')).sum().backward()
    assert float(alpha.grad[3]) == 0.0 and float(alpha.grad[4]) == 0.0


def test_dense_grid_lookup_matches_torch_grid_sample():
    """DenseGrid.forward == F.grid_sample(bilinear, align_corners, zeros) fwd and bwd (lib/grid.py:47-58)."""
    import torch.nn.functional as F
    from poseprobe_amd.grid import DenseGrid
    torch.manual_seed(0)
    lo, hi = [-1., -0.5, 0.], [1., 0.5, 2.]
    g = DenseGrid(channels=12, world_size=[7, 9, 5], xyz_min=lo, xyz_max=hi)
    g.grid.data.normal_()
    g = g.cuda()
    pts = (torch.rand(500, 3) * torch.tensor([2.4, 1.2, 2.4]) + torch.tensor([-1.2, -0.6, -0.2])).cuda().requires_grad_(True)
    out = g(pts)
    wgt = torch.randn_like(out)
    (out * wgt).sum().backward()
    ref_grid = g.grid.detach().cpu().contiguous().requires_grad_(True)
    p = pts.detach().cpu().requires_grad_(True)
    ind = ((p.reshape(1, 1, 1, -1, 3) - torch.tensor(lo)) / (torch.tensor(hi) - torch.tensor(lo))).flip((-1,)) * 2 - 1
    ref = F.grid_sample(ref_grid, ind, mode='bilinear', align_corners=True).reshape(12, -1).T
    (ref * wgt.cpu()).sum().backward()
    assert_close(out.detach().cpu(), ref.detach(), rtol=1e-5, atol=1e-6, name='value')
    assert_close(g.grid.grad.cpu(), ref_grid.grad, rtol=1e-4, atol=1e-6, name='grid grad')
    assert_close(pts.grad.cpu(), p.grad, rtol=1e-4, atol=1e-5, name='pts grad')


def test_surface_point_queries_match_reference():
    """query_sdf_point_wocuda / _wodeform / _render (lib/voxurf_coarse.py:734-920) against the reference's outputs."""
    d = load('query_g24.npz')
    m = make_model(d)
    ro, rd = torch.tensor(d['rays_o']).cuda(), torch.tensor(d['rays_d']).cuda()
    rk = dict(near=0.24, far=4.8, bg=0, stepsize=1.5, inverse_y=True, flip_x=False, flip_y=False)
    c = lambda t: t.detach().cpu().numpy()
    jit = torch.tensor(d['jitter'])
    for use_deform in (True, False):
        for gs, tag in ((None, 'eval'), (50, 'train')):
            key = f'q_{"deform" if use_deform else "plain"}_{tag}'
            pts, mask, sdf_d = m.query_sdf_point_wocuda(ro, rd, global_step=gs, keep_dim=True, use_deform=use_deform,
                                                        jitter=jit, **rk)
            assert_close(c(sdf_d), d[key + '_sdf'], rtol=1e-4, atol=1e-5, name=key + ' sdf')
            assert np.array_equal(c(mask), d[key + '_mask']), key
            assert_close(c(pts)[d[key + '_mask']], d[key + '_pts'][d[key + '_mask']], rtol=1e-4, atol=2e-5, name=key + ' pts')
    pts, mask, sdf_d = m.query_sdf_point_wocuda_wodeform(ro, rd, global_step=None, keep_dim=True, **rk)
    assert_close(c(sdf_d), d['q_wodeform_sdf'], rtol=1e-4, atol=1e-5, name='wodeform sdf')
    assert np.array_equal(c(mask), d['q_wodeform_mask'])
    assert_close(c(pts)[d['q_wodeform_mask']], d['q_wodeform_pts'][d['q_wodeform_mask']], rtol=1e-4, atol=2e-5, name='wodeform pts')
    ro_g = ro.clone().requires_grad_(True)
    pts, mask, depth = m.query_sdf_point_wocuda_render(ro_g, rd, global_step=50, keep_dim=True, use_deform=True, jitter=jit, **rk)
    assert np.array_equal(c(mask), d['q_render_mask'])
    assert_close(c(depth), d['q_render_depth'], rtol=1e-4, atol=1e-5, name='render depth')
    assert_close(c(pts), d['q_render_pts'], rtol=1e-4, atol=2e-5, name='render pts')
    pts.sum().backward()                 # differentiable w.r.t. the rays (-> pose)
    assert torch.isfinite(ro_g.grad).all() and float(ro_g.grad.abs().sum()) > 0


def test_directvoxgo_twin_matches_reference():
    """DirectVoxGO.forward (lib/dvgo_ori.py:289-379) + gradients of density / k0 / rgbnet vs the reference's golden."""
    from poseprobe_amd import synthetic as syn
    from poseprobe_amd.dvgo_ori import DirectVoxGO
    d = load('dvgo_g16.npz')
    G = int(d['G'])
    m = DirectVoxGO(syn.XYZ_MIN, syn.XYZ_MAX, num_voxels=G ** 3, num_voxels_base=G ** 3, alpha_init=1e-2, rgbnet_dim=12,
                    rgbnet_direct=True, rgbnet_depth=3, rgbnet_width=128, posbase_pe=5, viewbase_pe=4,
                    fast_color_thres=1e-4)
    sd = m.state_dict()
    assert {'xyz_min', 'xyz_max', 'density', 'k0', 'posfreq', 'viewfreq', 'rgbnet.0.weight', 'rgbnet.0.bias',
            'rgbnet.2.0.weight', 'rgbnet.2.0.bias', 'rgbnet.3.weight', 'rgbnet.3.bias'} == set(sd.keys())
    sd['density'], sd['k0'] = torch.tensor(d['density']), torch.tensor(d['k0'])
    for li, key in enumerate(['rgbnet.0', 'rgbnet.2.0', 'rgbnet.3']):
        sd[key + '.weight'], sd[key + '.bias'] = torch.tensor(d[f'rgbnet.{li}.weight']), torch.tensor(d[f'rgbnet.{li}.bias'])
    m.load_state_dict(sd)
    m = m.cuda()
    ro, rd, vd = (torch.tensor(d[k]).cuda() for k in ('rays_o', 'rays_d', 'viewdirs'))
    out = m(ro, rd, vd, global_step=5, near=0.24, far=4.8, bg=1, stepsize=0.5, inverse_y=True, flip_x=False, flip_y=False,
            jitter=torch.tensor(d['jitter']))
    c = lambda t: t.detach().cpu().numpy()
    assert np.array_equal(c(out['mask_outbbox']), d['out.mask_outbbox'])
    assert (c(out['mask']) != d['out.mask']).mean() < 1e-3       # weights within rounding of the 1e-4 threshold
    for k in ('alphainv_cum', 'weights', 'rgb_marched', 'raw_alpha', 'depth'):
        assert_close(c(out[k]), d['out.' + k], rtol=1e-4, atol=1e-5, name=k)
    same = c(out['mask']) == d['out.mask']
    assert_close(c(out['raw_rgb'])[same], d['out.raw_rgb'][same], rtol=1e-4, atol=1e-5, name='raw_rgb')
    loss = ((out['rgb_marched'] - torch.tensor(d['target']).cuda()) ** 2).mean()
    assert_close(c(loss), d['loss'], rtol=1e-4, name='loss')
    loss.backward()
    tol = dict(rtol=1e-3, scaled=2e-5)
    assert_close(c(m.density.grad), d['grad_density'], atol=1e-9, name='g.density', **tol)
    assert_close(c(m.k0.grad), d['grad_k0'], atol=1e-9, name='g.k0', **tol)
    for li, lin in enumerate([m.rgbnet[0], m.rgbnet[2][0], m.rgbnet[3]]):
        assert_close(c(lin.weight.grad), d[f'grad.rgbnet.{li}.weight'], atol=1e-9, name=f'g.rgbnet{li}.W', **tol)
        assert_close(c(lin.bias.grad), d[f'grad.rgbnet.{li}.bias'], atol=1e-9, name=f'g.rgbnet{li}.b', **tol)


def test_reference_style_training_loop_with_hip_adam(tmp_path):
    """create_optimizer_or_freeze_model + Adam.step (lib/utils.py) driving the drop-in Voxurf for 2 steps, checked
    against the oracle trainer; then a checkpoint round trip through load_model."""
    from oracle import voxurf_oracle as O
    from poseprobe_amd import camera, synthetic as syn, utils
    from poseprobe_amd import voxurf_coarse as Model
    from poseprobe_amd.config import ConfigDict
    from poseprobe_amd.losses import object_losses
    from tests.helpers import params_from_npz, scene_for
    d = load('forward_g8_s10.npz')
    m = make_model(d)
    pm = Model.pose_model(i_train=np.arange(3), camera_noise=0.).cuda()
    pm.se3_refine.data.copy_(torch.tensor(d['se3']))
    cfg_train = ConfigDict(lrate_decay=10, lrate_k0=1e-1, lrate_rgbnet=1e-3, lrate_warp_network=1e-3, lrate_sdf_alpha=1e-2,
                           lrate_sdf_beta=1e-2, lrate_sdf=0, weight_main=1., weight_tv_k0=.01, weight_mask=.1, lr_pose=1e-3,
                           lr_pose_end=1e-4, sched_pose='ExponentialLR')
    opt = utils.create_optimizer_or_freeze_model(m, cfg_train, global_step=0)
    assert sorted(g['name'] for g in opt.param_groups) == ['k0', 'rgbnet', 'sdf_alpha', 'sdf_beta', 'warp_network']
    opt_pose, sched = utils.create_optimizer_pose(pm, cfg_train, max_iter=1000)
    P = params_from_npz(d)
    st = O.TrainState(P, scene_for(d['G']), torch.tensor(d['w2c_init']), torch.tensor(d['Ks']), torch.tensor(d['images']),
                      torch.tensor(d['masks']), se3_refine=torch.tensor(d['se3']), pose_iters=1000)
    H, W = int(d['H']), int(d['W'])
    imgs, msks = torch.tensor(d['images']).cuda(), torch.tensor(d['masks']).cuda()
    decay = 0.1 ** (1 / 10000)
    for s in range(2):
        idx, jit = syn.step_randomness(3 * H * W, int(d['n_rand']), seed=60 + s)
        st.step(torch.tensor(idx), torch.tensor(jit), 10 + s)
        opt.zero_grad(set_to_none=True)
        opt_pose.zero_grad()
        w2c, c2w = camera.current_pose_c2w(pm.se3_refine, torch.tensor(d['w2c_init']).cuda())
        target, mask, ro, rd, vd = Model.select_training_rays(torch.tensor(idx), imgs, msks, c2w, np.array([[H, W]] * 3), d['Ks'])
        out = m(ro, rd, vd, use_deform=True, global_step=10 + s, near=0.24, far=4.8, bg=0, stepsize=1.5, inverse_y=True,
                flip_x=False, flip_y=False, jitter=torch.tensor(jit))
        loss = object_losses(out, cfg_train, target, mask, 10 + s, 10000, True)[2]
        (loss * 0.1).backward()
        for g in opt.param_groups:
            g['lr'] = g['lr'] * decay
        opt.step()
        opt_pose.step()
        sched.step()
    c = lambda t: t.detach().cpu().numpy()
    dev = np.abs(c(m.k0.grid) - c(P['k0']))
    assert dev.max() < 0.02 and (dev > 1e-4).mean() < 0.02
    assert_close(c(pm.se3_refine), c(st.se3), rtol=0, atol=2e-4, name='se3 after 2 steps')
    assert_close(c(m.sdf_alpha), c(P['sdf_alpha']), rtol=0, atol=2e-4, name='sdf_alpha')
    path = str(tmp_path / 'last_ckpt.tar')
    torch.save({'global_step': 2, 'model_kwargs': m.get_kwargs(), 'MaskCache_kwargs': m.get_MaskCache_kwargs(),
                'model_state_dict': m.state_dict(), 'optimizer_state_dict': opt.state_dict()}, path)
    m2 = utils.load_model(Model.Voxurf, path)
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a.cpu(), b.cpu()), k


def test_directvoxgo_twin_backward_is_complete():
    """Random linear functional over the twin's differentiable outputs (pixels, alphainv_cum, weights, raw_alpha, depth):
    value and the gradients of density, k0 and rgbnet against the reference."""
    from poseprobe_amd import synthetic as syn
    from poseprobe_amd.dvgo_ori import DirectVoxGO
    d = load('dvgo_g16.npz')
    G = int(d['G'])
    m = DirectVoxGO(syn.XYZ_MIN, syn.XYZ_MAX, num_voxels=G ** 3, num_voxels_base=G ** 3, alpha_init=1e-2, rgbnet_dim=12,
                    rgbnet_direct=True, rgbnet_depth=3, rgbnet_width=128, posbase_pe=5, viewbase_pe=4, fast_color_thres=1e-4)
    sd = m.state_dict()
    sd['density'], sd['k0'] = torch.tensor(d['density']), torch.tensor(d['k0'])
    for li, key in enumerate(['rgbnet.0', 'rgbnet.2.0', 'rgbnet.3']):
        sd[key + '.weight'], sd[key + '.bias'] = torch.tensor(d[f'rgbnet.{li}.weight']), torch.tensor(d[f'rgbnet.{li}.bias'])
    m.load_state_dict(sd)
    m = m.cuda()
    ro, rd, vd = (torch.tensor(d[k]).cuda() for k in ('rays_o', 'rays_d', 'viewdirs'))
    out = m(ro, rd, vd, global_step=5, near=0.24, far=4.8, bg=1, stepsize=0.5, inverse_y=True, flip_x=False, flip_y=False,
            jitter=torch.tensor(d['jitter']))
    total = 0.
    for k in ('alphainv_cum', 'weights', 'rgb_marched', 'raw_alpha', 'depth'):
        total = total + (torch.tensor(d['lf_coef_' + k]).cuda() * out[k]).sum()
    c = lambda t: t.detach().cpu().numpy()
    assert_close(c(total), d['lf_value'], rtol=2e-4, atol=1e-3, name='functional value')
    total.backward()
    tol = dict(rtol=2e-3, atol=1e-6, scaled=1e-3)
    assert_close(c(m.density.grad), d['lf_g_density'], name='g.density', **tol)
    assert_close(c(m.k0.grad), d['lf_g_k0'], name='g.k0', **tol)
    for li, lin in enumerate([m.rgbnet[0], m.rgbnet[2][0], m.rgbnet[3]]):
        assert_close(c(lin.weight.grad), d[f'lf_g.rgbnet.{li}.weight'], name=f'g.rgbnet{li}.W', **tol)
        assert_close(c(lin.bias.grad), d[f'lf_g.rgbnet.{li}.bias'], name=f'g.rgbnet{li}.b', **tol)


def test_render_viewpoints_matches_the_reference_driver(tmp_path):
    """Full-image inference driver (lib/nvs_fun.py:39-188): two whole 32 x 32 views in 4096-ray chunks through
    Voxurf.inference, assembled images and disparity maps against the reference's own render_viewpoints (fixture produced by
    executing its function body), the three PSNRs it logs (2 decimals in the log), and the files it writes."""
    import types
    import zlib
    from poseprobe_amd import nvs_fun
    d = load('viewpoints_g24.npz')
    m = make_model(d)
    H, W = int(d['H']), int(d['W'])
    rk = dict(near=0.24, far=4.8, bg=0, stepsize=1.5, inverse_y=True, flip_x=False, flip_y=False)
    cfg = types.SimpleNamespace(data=types.SimpleNamespace(flip_x=False, flip_y=False))
    nv = d['w2c'].shape[0]
    rgbs, disps = nvs_fun.render_viewpoints(m, torch.tensor(d['w2c']), cfg, np.array([[H, W]] * nv), d['Ks'], False, rk,
                                            gt_imgs=d['gt'], masks=d['masks'], savedir=str(tmp_path), step=7)
    assert rgbs.shape == d['rgbs'].shape and disps.shape == d['disps'].shape
    assert_close(rgbs, d['rgbs'], rtol=1e-4, atol=1e-5, name='images')
    fin = np.isfinite(d['disps'])                              # rays that hit nothing have depth 0, disparity inf - in both
    assert np.array_equal(np.isfinite(disps), fin)
    assert_close(disps[fin], d['disps'][fin], rtol=1e-4, atol=1e-5, name='disparity maps')
    last = nvs_fun.render_viewpoints.last['per_view']
    got = np.array([last['psnr'], last['psnr_fore'], last['psnr_back']]).T
    assert np.abs(got - d['psnr_logged']).max() <= 0.0051 + 1e-3, (got, d['psnr_logged'])
    for i in range(nv):
        for name in (f'7_{i:03d}.png', f'7_render_{i:03d}.png', f'7_gt_{i:03d}.png', f'7_{i:03d}_normal.png'):
            raw = open(tmp_path / name, 'rb').read()
            assert raw[:8] == b'\x89PNG\r\n\x1a\n'
        raw = open(tmp_path / f'7_render_{i:03d}.png', 'rb').read()
        idat = raw[raw.index(b'IDAT') + 4:raw.index(b'IEND') - 8]
        px = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(H, 1 + 3 * W)[:, 1:].reshape(H, W, 3)
        assert np.array_equal(px, nvs_fun.to8b(rgbs[i]))
    with pytest.raises(NotImplementedError):
        nvs_fun.render_viewpoints(m, torch.tensor(d['w2c']), cfg, np.array([[H, W]] * nv), d['Ks'], False, rk, eval_ssim=True)


def test_remaining_module_surface_names(tmp_path):
    """SURVEY 8b names that no shipped configuration reaches, on the HIP operators: Voxurf MaskCache (coarse-SDF free-space
    query), the semantic and mask-cache ray samplers, DirectVoxGO's MaskCache / extract_fields; mesh extraction refuses with
    the reference line in the message."""
    import torch.nn.functional as F
    from poseprobe_amd import dvgo_ori, synthetic as syn, utils
    from poseprobe_amd import voxurf_coarse as Model
    g = torch.Generator().manual_seed(0)
    lo, hi = np.array([-1., -0.5, 0.], np.float32), np.array([1., 0.5, 2.], np.float32)
    sdf = torch.randn(9, 7, 5, generator=g)
    np.savez(tmp_path / 'coarse.npz', sdf_grid_xyz=sdf.numpy(), xyz_min=lo, xyz_max=hi)
    mc = Model.MaskCache(str(tmp_path / 'coarse.npz'), mask_cache_thres=0.1).cuda()
    pts = (torch.rand(4, 50, 3, generator=g) * torch.tensor(hi - lo) + torch.tensor(lo))
    ind = ((pts.reshape(1, 1, 1, -1, 3) - torch.tensor(lo)) / torch.tensor(hi - lo)).flip((-1,)) * 2 - 1
    ref = F.grid_sample(sdf[None, None], ind, align_corners=True).reshape(4, 50)
    got = mc(pts.cuda()).cpu()
    sure = (ref - 0.1).abs() > 1e-5
    assert got.shape == (4, 50) and torch.equal(got[sure], (ref < 0.1)[sure])
    np.save(tmp_path / 'pickled.npy', {'sdf_grid_xyz': sdf.numpy()}, allow_pickle=True)
    with pytest.raises(ValueError, match='pickled'):
        Model.MaskCache(str(tmp_path / 'pickled.npy'), 0.1)

    # semantic sampler: 20 / 30 / 50 % of min(|boundary|, |object|) pixels from the three lists, rays at exactly those pixels
    d = load('rays.npz')
    H, W = 8, 12
    K, c2w = torch.tensor(d['Ks'][0]), torch.tensor(d['c2w'][0]).cuda()
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing='ij')
    bgs, obs, bds = (ys[:, :4].reshape(-1), xs[:, :4].reshape(-1)), (ys[:, 4:8].reshape(-1), xs[:, 4:8].reshape(-1)), \
        (ys[:, 8:].reshape(-1), xs[:, 8:].reshape(-1))
    o, dd, vd, (x, y) = Model.get_rays_of_a_view_semantic(H, W, K, c2w, False, True, False, False, bgs, obs, bds)
    n = min(len(bds[0]), len(obs[0]))
    assert len(x) == int(n * .2) + int(n * .3) + int(n * .5)
    assert bool((x[:int(n * .2)] < 4).all()) and bool((x[-int(n * .5):] >= 4).all()) and bool((x[-int(n * .5):] < 8).all())
    full_o, full_d, full_v = Model.get_rays(H, W, K, c2w, True, False, False, normalize=False)
    assert torch.equal(dd.cpu(), full_d[y, x].cpu()) and torch.equal(vd.cpu(), full_v[y, x].cpu())
    imgs, msks = [torch.rand(H, W, 3, generator=g).cuda()] * 2, [torch.rand(H, W, 1, generator=g).cuda()] * 2
    samplers = dict(background=[bgs] * 2, object=[obs] * 2, boundary=[bds] * 2)
    rgb, msk, ro, rd, vd2, imsz = Model.get_training_rays_semantic(imgs, msks, [c2w, c2w], [(H, W)] * 2, [K, K], False, True, False,
                                                                   False, samplers)
    assert rgb.shape == (2 * len(x), 3) and msk.shape == (2 * len(x), 1) and ro.shape == rd.shape == (2 * len(x), 3) and imsz == [len(x)] * 2

    # mask-cache sampler: keeps the rays with a sample that is inside the box and inside the cache's known space
    dm = load('inference_g24.npz')
    m = make_model(dm)
    box = np.savez(tmp_path / 'all.npz', sdf_grid_xyz=-np.ones((4, 4, 4), np.float32), xyz_min=syn.XYZ_MIN, xyz_max=syn.XYZ_MAX)
    m.mask_cache = Model.MaskCache(str(tmp_path / 'all.npz'), mask_cache_thres=0.0).cuda()
    Hs = Ws = 16
    views = syn.make_views(2, Hs, Ws)
    c2ws = torch.tensor(np.stack([np.linalg.inv(np.vstack([w, [0, 0, 0, 1]]))[:3] for w in views['w2c']])).float().cuda()
    rk = dict(near=0.24, far=4.8, bg=0, stepsize=1.5, inverse_y=True, flip_x=False, flip_y=False)
    ims = [torch.tensor(im).cuda() for im in views['images']]
    Ks = [torch.tensor(k) for k in views['Ks']]
    rgb, ro, rd, vd3, imsz = Model.get_training_rays_in_maskcache_sampling(ims, c2ws, [(Hs, Ws)] * 2, Ks, False, True, False, False,
                                                                         m, rk)
    assert sum(imsz) == rgb.shape[0] == ro.shape[0] and 0 < sum(imsz) <= 2 * Hs * Ws
    o_all, d_all, _ = Model.get_rays_of_a_view(Hs, Ws, Ks[0], c2ws[0], False, True, False, False)
    _, out, _, _, _ = m.sample_ray_ori(rays_o=o_all.reshape(-1, 3), rays_d=d_all.reshape(-1, 3), **rk)
    assert imsz[0] == int((~out).any(-1).sum())                 # an all-inside cache: exactly the rays that enter the box
    with pytest.raises(NotImplementedError, match='1224'):
        m.extract_deform_geometry(None, None)

    # DirectVoxGO side
    dens = torch.randn(1, 1, 6, 6, 6, generator=g)
    torch.save({'MaskCache_kwargs': {'xyz_min': lo.tolist(), 'xyz_max': hi.tolist(), 'act_shift': -4.0, 'voxel_size_ratio': 1.0},
                'model_state_dict': {'density': dens}}, tmp_path / 'coarse_last.tar')
    mc2 = dvgo_ori.MaskCache(str(tmp_path / 'coarse_last.tar'), mask_cache_thres=1e-3).cuda()
    pooled = F.max_pool3d(dens, 3, padding=1, stride=1)
    ref = 1 - torch.exp(-F.softplus(F.grid_sample(pooled, ind, align_corners=True).reshape(4, 50) - 4.0))
    got = mc2(pts.cuda()).cpu()
    sure = (ref - 1e-3).abs() > 1e-6
    assert torch.equal(got[sure], (ref >= 1e-3)[sure])
    u = dvgo_ori.extract_fields(torch.tensor(lo), torch.tensor(hi), 5, lambda p: p.sum(-1), N=2)
    ax = [np.linspace(lo[a], hi[a], 5) for a in range(3)]
    assert np.allclose(u, ax[0][:, None, None] + ax[1][None, :, None] + ax[2][None, None, :], atol=1e-6)
    with pytest.raises(NotImplementedError, match='mcubes'):
        dvgo_ori.extract_geometry(torch.tensor(lo), torch.tensor(hi), 5, 0.0, lambda p: p.sum(-1))


def test_fused_object_losses_equal_the_torch_expressions():
    """losses.object_losses on CUDA tensors runs the two HIP loss kernels (values + gradient of the weighted sum); it must agree with
    the op-by-op torch expressions (lib/losses.py:34-74) in every scalar, in the total and in the gradient of the total w.r.t. all
    seven render outputs - and a caller that differentiates an INDIVIDUAL scalar (not done by the reference loop) still gets the
    right gradient through the fallback."""
    from poseprobe_amd import losses
    from poseprobe_amd.config import ConfigDict
    g = torch.Generator().manual_seed(4)
    N, M = 300, 2500
    r = lambda *s: torch.rand(*s, generator=g)
    base = dict(rgb_marched=r(N, 3), alphainv_cum=r(N) * 0.98 + 0.01, cum_weights=r(N, 1) * 1.1 - 0.05, gradient=torch.randn(M, 3, generator=g),
                grad_deform=torch.randn(M, 3, 3, generator=g) * 0.1, sdf_correct=torch.randn(M, 1, generator=g) * 0.01,
                sdf_deform=torch.randn(M, generator=g) * 0.02)
    base['alphainv_cum'][:5] = torch.tensor([0., 1., 5e-7, 1 - 5e-7, 0.5])          # the clamps' dead zones
    base['cum_weights'][:4, 0] = torch.tensor([0., 1., 5e-4, 0.9995])
    base['sdf_correct'][:3, 0] = 0.
    target, mask = r(N, 3).cuda(), (r(N, 1) < 0.6).float().cuda()
    cfg = ConfigDict(weight_main=1.0, weight_tv_k0=0.01, weight_mask=0.1)
    k0_tv = torch.tensor(0.37, device='cuda', requires_grad=True)

    def run(fn, pick=None):
        mo = {k: v.clone().cuda().requires_grad_(True) for k, v in base.items()}
        mo['k0_tv'] = k0_tv
        S, Wt, loss = fn(mo, cfg, target, mask, 1234, 10000, True)
        obj = loss * 0.1 if pick is None else S[pick] * 2.0 + loss * 0.1
        grads = torch.autograd.grad(obj, [mo[k] for k in base] + [k0_tv], allow_unused=True)
        return S, Wt, loss, grads

    S0, W0, l0, g0 = run(losses._object_losses_torch)
    S1, W1, l1, g1 = run(losses.object_losses)
    assert list(S0.keys()) == list(S1.keys()) and dict(W0) == dict(W1)
    for k in S0:
        assert_close(S1[k].detach().cpu(), S0[k].detach().cpu().numpy(), rtol=2e-5, atol=1e-7, name='loss.' + k)
    assert_close(l1.detach().cpu(), l0.detach().cpu().numpy(), rtol=2e-5, name='loss')
    for k, a, b in zip(list(base) + ['k0_tv'], g1, g0):
        assert_close(a.cpu(), b.cpu().numpy(), rtol=1e-4, atol=1e-9, name='d loss / d ' + k)
    for pick in ('img_render', 'sdf_deform_constraint'):                  # individual scalar + total: the fallback path
        _, _, _, ga = run(losses._object_losses_torch, pick)
        _, _, _, gb = run(losses.object_losses, pick)
        for k, a, b in zip(list(base) + ['k0_tv'], gb, ga):
            assert_close(a.cpu(), b.cpu().numpy(), rtol=1e-4, atol=1e-9, name=f'd ({pick} + loss) / d {k}')


def test_adam_step_batches_small_tensors_into_one_launch():
    """utils.Adam.step sends the small tensors of a group through pp_adam_upd_multi (32 per launch): same numbers as one
    pp_adam_upd per tensor, over several steps and group settings."""
    from poseprobe_amd import render_utils, utils
    g = torch.Generator().manual_seed(2)
    shapes = [(128, 57), (128,), (128, 128), (128,), (3, 128), (3,), (1,)] * 6         # 42 tensors: two launches
    ps = [torch.randn(*s, generator=g).cuda().requires_grad_(True) for s in shapes]
    ref = [p.detach().clone() for p in ps]
    rm, rv = [torch.zeros_like(p) for p in ref], [torch.zeros_like(p) for p in ref]
    opt = utils.Adam([dict(params=ps, lr=1e-3, name='net')], betas=(0.9, 0.99))
    for step in range(1, 4):
        grads = [torch.randn(*s, generator=g).cuda() * 1e-3 for s in shapes]
        for p, gr in zip(ps, grads):
            p.grad = gr
        opt.step()
        eps = 1e-8 * np.sqrt(1 - 0.99 ** step)
        for p, gr, m, v in zip(ref, grads, rm, rv):
            render_utils.adam_upd(p, gr, m, v, step, 0.9, 0.99, 1e-3, eps)
    torch.cuda.synchronize()
    for a, b in zip(ps, ref):
        assert torch.equal(a.detach(), b)


def test_sync_free_ray_inference_equals_inference():
    """Voxurf.inference_rays (the whole-view driver's chunk call: no host round trip for the sample count, buffers sized for the
    sampler's worst case and kept on the module, normals composited by the marching kernel) returns the per-ray entries of
    Voxurf.inference (lib/voxurf_coarse.py:1094-1222) - also on a second call that reuses the cached buffers with a chunk
    whose sample count is SMALLER (stale rows past the count must not leak into any ray)."""
    d = load('inference_g24.npz')
    m = make_model(d)
    ro, rd, vd = (torch.tensor(d[k]).cuda() for k in ('rays_o', 'rays_d', 'viewdirs'))
    kw = dict(near=0.24, far=4.8, bg=0, stepsize=1.5, inverse_y=True, flip_x=False, flip_y=False)
    c = lambda t: t.detach().cpu().numpy()
    for sel in (slice(None), torch.arange(ro.shape[0] - 1, -1, -1, device='cuda')):     # all rays, then the same count in another order
        a = m.inference(ro[sel], rd[sel], vd[sel], global_step=None, **kw)
        b = m.inference_rays(ro[sel], rd[sel], vd[sel], global_step=None, **kw)
        for k in ('rgb_marched', 'alphainv_cum', 'cum_weights', 'depth', 'disp'):
            assert a[k].shape == b[k].shape, k
            assert torch.equal(a[k], b[k]), k                        # same kernels on the same samples: bit-identical
        assert_close(c(b['normal_marched']), c(a['normal_marched']), rtol=1e-5, atol=1e-6, name='normal_marched')
    # a chunk that mostly misses the box after a full one: nothing stale survives
    far = ro.clone()
    far[::2] += torch.tensor([50.0, 0.0, 0.0], device='cuda')
    a = m.inference(far, rd, vd, global_step=None, **kw)
    b = m.inference_rays(far, rd, vd, global_step=None, **kw)
    for k in ('rgb_marched', 'alphainv_cum', 'depth'):
        assert torch.equal(a[k], b[k]), k
    assert_close(c(b['normal_marched']), c(a['normal_marched']), rtol=1e-5, atol=1e-6, name='normal_marched (sparse chunk)')
    # the reference values of the fixture through the new entry point
    b = m.inference_rays(ro, rd, vd, global_step=None, **kw)
    for k in ('alphainv_cum', 'cum_weights', 'rgb_marched', 'normal_marched', 'depth'):
        assert_close(c(b[k]), d['out.' + k], rtol=1e-4, atol=1e-5, scaled=1e-6, name=k)
