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


class _OracleSurface:
    """The oracle's zero-crossing query behind the three attributes / one method get_project_error needs (CPU checker)."""

    def __init__(self, d):
        from oracle import voxurf_oracle as O
        from poseprobe_amd import synthetic as syn
        G = int(d['G'])
        rs = syn.range_shape()
        self.O = O
        self.scene = O.Scene(syn.XYZ_MIN, syn.XYZ_MAX, G ** 3, output_range=float(rs.max()), rect_size=rs.tolist())
        self.sdf = torch.tensor(d['P.sdf'])
        self.xyz_min, self.xyz_max = self.scene.xyz_min, self.scene.xyz_max
        self.diagonal_length = torch.sqrt(torch.sum(self.xyz_max - self.xyz_min ** 2))       # sic, lib/voxurf_coarse.py:102

    def query_sdf_point_wocuda_wodeform(self, o, dd, global_step=None, keep_dim=True, jitter=None, **_):
        return self.O.query_wodeform(self.scene, self.sdf, o, dd, jitter if global_step is not None else None)


def test_oracle_zero_crossing_query_reproduces_the_reference_pose_gradient_cpu():
    """oracle.query_wodeform (restating lib/voxurf_coarse.py:797-837) inside recon_utils.get_project_error against what the
    reference's own get_project_error(use_deform=False) produced: both losses and d/d se3 (fixture g_se3_plain)."""
    from oracle import voxurf_oracle as O
    from poseprobe_amd import recon_utils as R
    d = load('reproj_g24.npz')
    se3 = torch.tensor(d['se3'], requires_grad=True)
    init = torch.tensor(d['w2c_init'])
    w2c = torch.cat([init[:1], O.pose_compose_pair(O.se3_to_SE3(se3), init)[1:]], 0)    # the package's lie algebra is HIP-only
    err, near = R.get_project_error(_OracleSurface(d), torch.tensor(d['Ks']), np.array([[int(d['H']), int(d['W'])]] * 3),
                                    float(d['nl']), 50, w2c, torch.tensor(d['coord0']), torch.tensor(d['coord1']), d['i_train'],
                                    d['j_train'], torch.tensor(d['mconf']), use_deform=False, pixel_thre=200, near=0.24, far=4.8,
                                    bg=0, stepsize=1.5, flip_x=False, flip_y=False, jitter=torch.tensor(d['jitter']))
    assert_close(np.float32(err.item()), d['err_plain'], rtol=2e-4, atol=1e-5, name='projection_dis_error')
    assert_close(np.float32(near.item()), d['near_plain'], rtol=1e-5, atol=1e-5, name='near_surface_loss')
    (err + near).backward()
    assert np.abs(d['g_se3_plain']).max() > 1.0                  # the fixture's gradient is far from zero
    assert_close(se3.grad.numpy(), d['g_se3_plain'], rtol=1e-3, atol=1e-4, scaled=1e-4, name='d/d se3 (zero-crossing query)')


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
    # both queries are differentiable end to end: the rendered-depth query through the render backward, the zero-crossing query
    # (use_deform=False: what the live loop runs while <= 2 views are active, recon_scene.py:584, :624-631) through
    # pp_sdf_crossing_dense_bwd
    (err + near).backward()
    assert_close(se3.grad.cpu().numpy(), d[f'g_se3_{tag}'], rtol=2e-3, atol=1e-4, scaled=1e-3, name='d/d se3')


@pytest.mark.gpu
@pytest.mark.parametrize('train', [True, False])
def test_zero_crossing_query_backward_matches_the_oracle(train):
    """A random linear functional over BOTH differentiable outputs of query_sdf_point_wocuda_wodeform (surface points and
    the dense SDF row) differentiated w.r.t. rays_o / rays_d by the HIP node and by the oracle (autograd over its restatement
    of lib/voxurf_coarse.py:797-837), incl. rays that miss the box, rays without a crossing and un-normalised directions."""
    from oracle import voxurf_oracle as O
    from tests.test_hip_dropin import make_model
    d = load('reproj_g24.npz')
    m = make_model(d)
    surf = _OracleSurface(d)
    g = torch.Generator().manual_seed(5)
    N = 301
    o = torch.randn(N, 3, generator=g) * 0.15 + torch.tensor([0.0, 0.0, -1.6])
    tgt = (torch.rand(N, 3, generator=g) - 0.5) * torch.tensor([1.6, 1.6, 1.4]) + torch.tensor([0., 0., -0.1])
    dd = tgt - o
    dd = dd / dd.norm(dim=-1, keepdim=True) * (0.5 + torch.rand(N, 1, generator=g))       # |d| != 1: the norm path counts
    jit = torch.rand(N, generator=g) if train else None
    c_pts, c_sdf = torch.randn(N, 3, generator=g), torch.randn(N, surf.scene.n_samples(), generator=g) * 0.05
    rk = dict(near=0.24, far=4.8, stepsize=1.5, bg=0)
    res = {}
    for name, dev in (('oracle', 'cpu'), ('hip', 'cuda')):
        oo, dv = o.clone().to(dev).requires_grad_(True), dd.clone().to(dev).requires_grad_(True)
        if name == 'oracle':
            pts, hit, sdf_d = O.query_wodeform(surf.scene, surf.sdf, oo, dv, jit)
        else:
            pts, hit, sdf_d = m.query_sdf_point_wocuda_wodeform(oo, dv, global_step=7 if train else None, keep_dim=True,
                                                                jitter=None if jit is None else jit.cuda(), **rk)
        val = (pts * c_pts.to(dev)).sum() + (sdf_d * c_sdf.to(dev)).sum()
        val.backward()
        res[name] = (pts.detach().cpu(), hit.cpu(), sdf_d.detach().cpu(), oo.grad.cpu(), dv.grad.cpu())
    (p0, h0, s0, go0, gd0), (p1, h1, s1, go1, gd1) = res['oracle'], res['hip']
    assert 0.2 < h0.float().mean() < 0.98, 'the case must hold hits and misses'
    assert_close(s1.numpy(), s0.numpy(), rtol=1e-4, atol=2e-5, name='dense sdf row')
    same = (h0 == h1)
    assert same.float().mean() > 0.99
    assert_close(p1[same & h0].numpy(), p0[same & h0].numpy(), rtol=1e-4, atol=2e-5, name='surface points')
    # rays whose crossing sits within rounding distance of an interval end (z0 zeroed on one side only) are excluded
    stable = same & ((p0 - p1).abs().amax(-1) < 1e-3)
    assert stable.float().mean() > 0.98
    assert_close(go1[stable].numpy(), go0[stable].numpy(), rtol=2e-3, atol=1e-5, scaled=1e-3, name='d/d rays_o')
    assert_close(gd1[stable].numpy(), gd0[stable].numpy(), rtol=2e-3, atol=1e-5, scaled=1e-3, name='d/d rays_d')


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


@pytest.mark.gpu
def test_zero_crossing_query_on_a_long_ray_and_a_rough_template():
    """pp_sdf_first_crossing / pp_sdf_crossing_dense_bwd beyond one wavefront of samples per ray (64^3 grid: 76 samples, the
    first sign change is searched in two ballot rounds) on a ROUGH template (smooth random field: crossings at every depth,
    several per ray, some rays without any): value and gradient w.r.t. the rays against the oracle."""
    from oracle import voxurf_oracle as O
    from poseprobe_amd import synthetic as syn
    from poseprobe_amd import voxurf_coarse as Model
    G = 64
    rs = syn.range_shape()
    scene = O.Scene(syn.XYZ_MIN, syn.XYZ_MAX, G ** 3, output_range=float(rs.max()), rect_size=rs.tolist())
    assert scene.n_samples() == 76
    g = torch.Generator().manual_seed(9)
    # template: positive everywhere except behind the plane x + y + z = 1.35 near the FAR corner of the box (+ a little smooth noise):
    # rays along the box diagonal cross it ~67 samples in, rays towards random targets cross it anywhere or not at all
    lin = [torch.linspace(float(lo), float(hi), G) for lo, hi in zip(syn.XYZ_MIN, syn.XYZ_MAX)]
    gx, gy, gz = torch.meshgrid(*lin, indexing='ij')
    noise = torch.nn.functional.interpolate(torch.randn(1, 1, 9, 9, 9, generator=g), size=(G, G, G), mode='trilinear', align_corners=True)
    sdf = (0.3 - 0.6 * torch.sigmoid((gx + gy + gz - 1.35) / 0.05))[None, None] + 0.02 * noise
    m = Model.Voxurf(syn.XYZ_MIN, syn.XYZ_MAX, num_voxels=G ** 3, num_voxels_base=G ** 3, alpha_init=1e-2, rgbnet_dim=12, rgbnet_direct=True,
                     rgbnet_depth=4, rgbnet_width=128, posbase_pe=5, viewbase_pe=1, geo_rgb_dim=3, s_ratio=50, s_start=0.2, barf_c2f=[0.6, 1],
                     i_train=np.arange(3), N_iters=10000, HW=np.array([[32, 32]] * 3), range_shape=rs, rect_size=rs.tolist(), camera_noise=0.)
    sd = m.state_dict()
    sd['sdf.grid'] = sdf.clone()
    m.load_state_dict(sd)
    m = m.cuda()
    N = 257
    o = torch.randn(N, 3, generator=g) * 0.05 + torch.tensor([-1.2, -1.2, -1.4])          # outside the near corner
    tgt = (torch.rand(N, 3, generator=g) - 0.5) * torch.tensor([1.3, 1.3, 1.3]) + torch.tensor([0., 0., -0.1])
    tgt[::2] = torch.tensor([0.6, 0.6, 0.5]) + torch.randn((N + 1) // 2, 3, generator=g) * 0.03    # every other ray: along the diagonal
    dd = tgt - o
    dd = dd / dd.norm(dim=-1, keepdim=True)
    jit = torch.rand(N, generator=g)
    c_pts = torch.randn(N, 3, generator=g)
    res = {}
    for name, dev in (('oracle', 'cpu'), ('hip', 'cuda')):
        oo, dv = o.clone().to(dev).requires_grad_(True), dd.clone().to(dev).requires_grad_(True)
        if name == 'oracle':
            pts, hit, sdf_d = O.query_wodeform(scene, sdf, oo, dv, jit)
        else:
            pts, hit, sdf_d = m.query_sdf_point_wocuda_wodeform(oo, dv, global_step=3, keep_dim=True, jitter=jit.cuda(), near=0.24, far=4.8,
                                                                stepsize=1.5, bg=0)
        (pts * c_pts.to(dev)).sum().backward()
        res[name] = (pts.detach().cpu(), hit.cpu(), sdf_d.detach().cpu(), oo.grad.cpu(), dv.grad.cpu())
    (p0, h0, s0, go0, gd0), (p1, h1, s1, go1, gd1) = res['oracle'], res['hip']
    first = torch.argmax(((s0[:, :-1] * s0[:, 1:]) <= 0).float(), 1)
    assert 0.3 < h0.float().mean() < 0.99 and int((first[h0] >= 63).sum()) >= 20, 'need hits, misses and crossings past slot 63'
    assert_close(s1.numpy(), s0.numpy(), rtol=1e-4, atol=2e-5, name='dense sdf row')
    same = h0 == h1
    assert same.float().mean() > 0.98
    stable = same & ((p0 - p1).abs().amax(-1) < 1e-3)
    assert stable.float().mean() > 0.97
    assert_close(p1[stable & h0].numpy(), p0[stable & h0].numpy(), rtol=1e-4, atol=2e-5, name='surface points')
    assert_close(go1[stable].numpy(), go0[stable].numpy(), rtol=2e-3, atol=1e-5, scaled=1e-3, name='d/d rays_o')
    assert_close(gd1[stable].numpy(), gd0[stable].numpy(), rtol=2e-3, atol=1e-5, scaled=1e-3, name='d/d rays_d')
