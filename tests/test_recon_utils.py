"""poseprobe_amd.recon_utils against outputs of the reference's own function bodies (tests/golden/reproj_g24.npz, generated
by oracle/make_golden.py::gen_reproj): the pure-torch helpers on CPU, the full reprojection / near-surface loss with the
surface queries on the GPU."""
import numpy as np
import pytest
import torch

from tests.helpers import assert_close, load


def test_ray_helpers_match_reference_cpu():
    from poseprobe_amd import recon_utils as R
    from poseprobe_amd import camera
    d = load('reproj_g24.npz')
    Ks = torch.tensor(d['Ks'])
    it = d['i_train']
    c2w = camera.pose.invert(torch.tensor(d['w2c_init'])[it])
    pts = torch.tensor(d['coord0'])
    for inv in (1, 0):
        o, dd = R.get_ray_dir(pts.clone(), Ks[it], c2w, inverse_y=bool(inv), flip_x=False, flip_y=False, mode='no_center')
        assert_close(o.numpy(), d[f'raydir_o_inv{inv}'], rtol=1e-6, atol=1e-7, name='rays_o')
        assert_close(dd.numpy(), d[f'raydir_d_inv{inv}'], rtol=1e-5, atol=1e-6, name='rays_d')
    o, dd = R.get_ray_dir(pts.clone(), Ks[it], c2w, inverse_y=True, flip_x=False, flip_y=False, mode='center')
    assert_close(dd.numpy(), d['raydir_d_center'], rtol=1e-5, atol=1e-6, name='rays_d centre')
    dist = R.point_to_ray_distance(o.reshape(-1, 3), dd.reshape(-1, 3), torch.tensor(d['p2r_point']))
    assert_close(dist.numpy(), d['p2r_dist'], rtol=1e-5, atol=1e-6, name='point_to_ray_distance')
    # huber / masking arithmetic of compute_diff_loss on a hand-checkable case
    diff = torch.tensor([[0.5, 3.0, 2.0]])
    got = R.compute_diff_loss('huber', diff, weights=torch.tensor([[1., 2., 4.]]), mask=torch.tensor([[True, True, False]]))
    assert abs(float(got) - (0.125 + 2 * 2.5) / (2 + 1e-6)) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize('use_deform', [True, False])
def test_project_error_matches_reference(use_deform):
    from poseprobe_amd import camera
    from poseprobe_amd import recon_utils as R
    from tests.test_hip_dropin import make_model
    d = load('reproj_g24.npz')
    m = make_model(d)
    dev = 'cuda'
    se3 = torch.tensor(d['se3'], device=dev, requires_grad=True)
    init = torch.tensor(d['w2c_init'], device=dev)
    w2c = torch.cat([init[:1], camera.pose.compose([camera.lie.se3_to_SE3(se3), init])[1:]], 0)
    rk = dict(near=0.24, far=4.8, bg=0, stepsize=1.5, flip_x=False, flip_y=False,
              jitter=torch.tensor(d['jitter']))        # the reference draws it with rand_like; the fixture carries it
    err, near = R.get_project_error(m, torch.tensor(d['Ks'], device=dev), np.array([[int(d['H']), int(d['W'])]] * 3), float(d['nl']),
                                    50, w2c, torch.tensor(d['coord0'], device=dev), torch.tensor(d['coord1'], device=dev),
                                    d['i_train'], d['j_train'], torch.tensor(d['mconf'], device=dev), use_deform=use_deform,
                                    pixel_thre=200, **rk)
    tag = 'deform' if use_deform else 'plain'
    assert_close(np.float32(err.item()), d[f'err_{tag}'], rtol=2e-4, atol=1e-5, name='projection_dis_error')
    assert_close(np.float32(near.item()), d[f'near_{tag}'], rtol=1e-5, atol=1e-5, name='near_surface_loss')
    if use_deform:          # the rendered-depth query is differentiable end to end (pose gradient through the HIP backward)
        (err + near).backward()
        assert_close(se3.grad.cpu().numpy(), d['g_se3_deform'], rtol=2e-3, atol=1e-4, scaled=1e-3, name='d/d se3')


@pytest.mark.gpu
def test_training_depth_is_differentiable_through_the_entry_distance():
    """`depth = t_min / |d| + sum(w * step)` of the training forward (lib/voxurf_coarse.py:1050-1058): value and
    d(sum depth)/d(se3) against the reference - the t_min term reaches the pose explicitly, not only through the samples."""
    from poseprobe_amd import camera
    from poseprobe_amd import voxurf_coarse as Model
    from tests.test_hip_dropin import make_model
    d = load('reproj_g24.npz')
    m = make_model(d)
    dev = 'cuda'
    H, W = int(d['H']), int(d['W'])
    se3 = torch.tensor(d['se3'], device=dev, requires_grad=True)
    init = torch.tensor(d['w2c_init'], device=dev)
    w2c = torch.cat([init[:1], camera.pose.compose([camera.lie.se3_to_SE3(se3), init])[1:]], 0)
    c2w = camera.pose.invert(w2c)
    idx = torch.tensor(d['depth_ray_idx'], device=dev)
    zeros = torch.zeros(3, H, W, 3, device=dev)
    _, _, ro, rd, vd = Model.select_training_rays(idx, zeros, torch.ones(3, H, W, 1, device=dev), c2w,
                                                  np.array([[H, W]] * 3), torch.tensor(d['Ks'], device=dev))
    out = m(ro, rd, vd, use_deform=True, global_step=50, near=0.24, far=4.8, bg=0, stepsize=1.5, inverse_y=True, flip_x=False,
            flip_y=False, jitter=torch.tensor(d['depth_jitter']))
    assert_close(out['depth'].detach().cpu().numpy(), d['depth_vals'], rtol=1e-4, atol=1e-5, name='depth')
    out['depth'].sum().backward()
    assert_close(se3.grad.cpu().numpy(), d['depth_g_se3'], rtol=2e-3, atol=1e-4, scaled=1e-3, name='d depth / d se3')


@pytest.mark.gpu
def test_backward_is_complete_over_every_output_of_the_forward_dict():
    """A random linear functional over EVERY differentiable entry of Voxurf.forward's dict (pixels, weights, raw alpha /
    rgb, depth, disp, normals, deformation terms, k0 TV) differentiated by the reference and by the HIP autograd node:
    the value and the gradients of the pose, alpha/beta, both MLPs and the colour grid must agree - no output may be
    silently non-differentiable."""
    from poseprobe_amd import camera
    from poseprobe_amd import voxurf_coarse as Model
    from tests.test_hip_dropin import make_model
    d = load('reproj_g24.npz')
    m = make_model(d)
    dev = 'cuda'
    H, W = int(d['H']), int(d['W'])
    se3 = torch.tensor(d['se3'], device=dev, requires_grad=True)
    init = torch.tensor(d['w2c_init'], device=dev)
    c2w = camera.pose.invert(torch.cat([init[:1], camera.pose.compose([camera.lie.se3_to_SE3(se3), init])[1:]], 0))
    idx = torch.tensor(d['depth_ray_idx'], device=dev)
    _, _, ro, rd, vd = Model.select_training_rays(idx, torch.zeros(3, H, W, 3, device=dev), torch.ones(3, H, W, 1, device=dev),
                                                  c2w, np.array([[H, W]] * 3), torch.tensor(d['Ks'], device=dev))
    out = m(ro, rd, vd, use_deform=True, global_step=50, near=0.24, far=4.8, bg=0, stepsize=1.5, inverse_y=True, flip_x=False,
            flip_y=False, jitter=torch.tensor(d['depth_jitter']))
    total = 0.
    for k in [f[len('lf_coef_'):] for f in d if f.startswith('lf_coef_')]:
        c = torch.tensor(d['lf_coef_' + k], device=dev)
        assert out[k].shape == c.shape, (k, out[k].shape, c.shape)
        total = total + (c * out[k]).sum()
    assert_close(np.float32(total.item()), d['lf_value'], rtol=2e-4, atol=1e-3, name='functional value')
    total.backward()
    tol = dict(rtol=2e-3, atol=1e-5, scaled=1e-3)
    assert_close(se3.grad.cpu().numpy(), d['lf_g_se3'], name='d/d se3', **tol)
    for name, prm in m.named_parameters():
        key = 'lf_g.' + name
        if key in d:
            assert prm.grad is not None, name
            assert_close(prm.grad.detach().cpu().numpy(), d[key], name=name, **tol)
