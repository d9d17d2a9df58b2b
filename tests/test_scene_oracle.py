"""Pins the scene-branch oracle (oracle/scene_nerf.py) against the fixture produced by executing the reference's
frequency_nerf.NeRF (oracle/make_golden.py: gen_scene).  CPU only."""
import numpy as np
import torch

from oracle import scene_nerf as SN
from tests.helpers import assert_close, load


def scene_inputs(d, dtype=torch.float32):
    P = {k[6:]: torch.tensor(v, dtype=dtype).requires_grad_(True) for k, v in d.items() if k.startswith('param.')}
    B, N, S = int(d['B']), int(d['N']), int(d['S'])
    center = torch.tensor(d['center'], dtype=dtype).reshape(B * N, 3).requires_grad_(True)
    ray = torch.tensor(d['ray'], dtype=dtype).reshape(B * N, 3).requires_grad_(True)
    depth = torch.tensor(d['depth_samples'], dtype=dtype).reshape(B * N, S)
    return P, center, ray, depth


def linear_functional(d, out):
    total = 0.
    for k in ('rgb', 'depth', 'opacity', 'weights', 'rgb_samples', 'density_samples'):
        c = torch.tensor(d['lf_coef_' + k]).reshape(out[k].shape).to(out[k].dtype)
        total = total + (c * out[k]).sum()
    return total


def test_scene_oracle_matches_reference_outputs_and_gradients():
    d = load('scene_b2.npz')
    P, center, ray, depth = scene_inputs(d)
    out = SN.render(P, center, ray, depth, float(d['progress']), tuple(d['barf_c2f']))
    for k in ('rgb_samples', 'density_samples', 'rgb', 'rgb_var', 'depth', 'depth_var', 'opacity', 'weights', 'all_cumulated'):
        assert_close(out[k].reshape(-1), d['out.' + k].reshape(-1), rtol=2e-5, atol=2e-6, name=k)
    total = linear_functional(d, out)
    assert_close(total, d['lf_value'], rtol=1e-5, atol=1e-4, name='lf_value')
    total.backward()
    assert_close(center.grad.reshape(-1), d['lf_g_center'].reshape(-1), rtol=1e-4, scaled=1e-5, name='g_center')
    assert_close(ray.grad.reshape(-1), d['lf_g_ray'].reshape(-1), rtol=1e-4, scaled=1e-5, name='g_ray')
    for k, p in P.items():
        assert_close(p.grad, d['lf_g.' + k], rtol=1e-4, scaled=1e-5, name='g.' + k)


def test_scene_oracle_photometric_loss():
    d = load('scene_b2.npz')
    rgb = torch.tensor(d['out.rgb'])
    assert_close(SN.photometric_loss(rgb, torch.tensor(d['loss_image'])), d['loss_render'], rtol=1e-6, name='loss_render')


def test_band_weights_window():
    w = SN.band_weights(0.565, (0.4, 0.7), 10)
    assert np.all(w[:5].numpy() == 1.0) and np.all(w[6:].numpy() == 0.0) and 0.0 < float(w[5]) < 1.0
    assert torch.equal(SN.band_weights(0.1, None, 4), torch.ones(4))


def test_scene_host_helpers_match_reference_render():
    """Host-side pieces of SceneRenderer (per-ray torch algebra, no kernels): rays at pixels (utils/camera.py:384-416) and
    the coarse-to-fine inverse-transform sampler (renderer.py:702-738) against the recorded reference render."""
    from poseprobe_amd import bg_nerf
    d = load('scene_render.npz')
    center, ray = bg_nerf.get_center_and_ray_at_pixels(torch.tensor(d['pose']), torch.tensor(d['pixels']), torch.tensor(d['intr']))
    assert_close(center, d['late.origins'], rtol=1e-6, atol=1e-6, name='origins')
    assert_close(ray, d['late.viewdirs'], rtol=1e-6, atol=1e-6, name='viewdirs')
    lo, hi = (float(x) for x in d['depth_range'])
    fine = bg_nerf.sample_depth_from_pdf(torch.tensor(d['late.weights'])[..., 0], int(d['n_coarse']), int(d['n_fine']), (lo, hi),
                                         det=False, grid=torch.tensor(d['late.rand1']))
    t = torch.cat([torch.tensor(d['late.t']), fine], dim=2).sort(dim=2).values
    assert_close(t, d['late.t_fine'], rtol=1e-6, atol=1e-6, name='t_fine')
    opt = bg_nerf.default_options(sample_intvs=int(d['n_coarse']))
    dep = bg_nerf.sample_depth(opt, 2, 20, int(d['n_coarse']), (lo, hi), mode='val', device='cpu')
    step = (hi - lo) / int(d['n_coarse'])
    assert_close(dep[0, 0, :, 0], lo + step * (np.arange(int(d['n_coarse'])) + 0.5), rtol=1e-6, name='midpoints')
    assert bool(((torch.tensor(d['late.t'])[..., 0] >= lo) & (torch.tensor(d['late.t'])[..., 0] <= hi)).all())


def test_correspondence_loss_helpers_match_reference():
    """bg_losses.reprojection_loss / project_to_other_img / pose_inverse_4x4 / compute_diff_loss against the reference's own
    function bodies (oracle/make_golden.py: gen_scene_corres): loss, consistency-filter statistics and the gradients w.r.t.
    the rendered depths and the relative pose, for the plain Huber loss, the filtered variant and the end-point error."""
    from poseprobe_amd import bg_losses, bg_nerf
    d = load('scene_corres.npz')
    K = torch.tensor(d['K'])
    pix_self, pix_other, conf = torch.tensor(d['pix_self']), torch.tensor(d['pix_other']), torch.tensor(d['conf'])
    assert_close(bg_losses.pose_inverse_4x4(torch.tensor(d['T'])), d['T_inv'], rtol=1e-6, atol=1e-6, name='pose_inverse_4x4')
    assert_close(torch.tensor(d['P_other']) @ bg_losses.pose_inverse_4x4(torch.tensor(d['P_self'])), d['T'], rtol=1e-5, atol=1e-6,
                 name='T_self2other')
    opts = {'plain': dict(pix=False, dep=False, pt=20., dt=0.1, kind='huber'),
            'filtered': dict(pix=True, dep=True, pt=14., dt=0.5, kind='huber'),
            'epe': dict(pix=False, dep=False, pt=20., dt=0.1, kind='epe')}
    for label, o in opts.items():
        opt = bg_nerf.Options(renderrepro_do_pixel_reprojection_check=o['pix'], renderrepro_do_depth_reprojection_check=o['dep'],
                              renderrepro_pixel_reprojection_thresh=o['pt'], renderrepro_depth_reprojection_thresh=o['dt'],
                              diff_loss_type=o['kind'])
        T = torch.tensor(d['T'], requires_grad=True)
        d_self = torch.tensor(d['d_self'], requires_grad=True)
        d_other = torch.tensor(d['d_other'], requires_grad=True)
        loss, stats = bg_losses.reprojection_loss(opt, pix_self, d_self, K[0], pix_other, d_other, K[1], T, conf)
        loss.backward()
        assert_close(loss, d[label + '.loss'], rtol=1e-5, name=label + '.loss')
        assert_close(d_self.grad, d[label + '.g_d_self'], rtol=1e-4, atol=1e-6, name=label + '.g_d_self')
        assert_close(T.grad, d[label + '.g_T'], rtol=1e-4, atol=1e-5, name=label + '.g_T')
        for k in ('perc_val_pix_rep', 'perc_val_depth_rep'):
            if label + '.' + k in d:
                assert_close(stats[k], d[label + '.' + k], rtol=1e-6, name=label + '.' + k)
    assert 0.05 < float(d['filtered.perc_val_pix_rep']) < 0.95        # the pixel filter actually selects


def test_depth_consistency_geometry_matches_reference():
    """bg_losses.backproject_to_3d / project / nearest_pose_id / sample_virtual_pose against the reference's own function bodies
    (oracle/make_golden.py: gen_scene_depthcons)."""
    from poseprobe_amd import bg_losses
    d = load('scene_depthcons.npz')
    K, P, Pc2w = torch.tensor(d['K']), torch.tensor(d['P']), torch.tensor(d['P_c2w'])
    assert_close(bg_losses.pose_inverse_4x4(P), d['P_c2w'], rtol=1e-6, atol=1e-6, name='batched pose_inverse_4x4')
    pts = bg_losses.backproject_to_3d(torch.tensor(d['pix']), torch.tensor(d['depth']), K[1], Pc2w[1])
    assert_close(pts, d['pts3d'], rtol=1e-5, atol=1e-6, name='backproject_to_3d')
    px, dj = bg_losses.project(pts, P[3], K[3])
    assert_close(px, d['proj_px'], rtol=1e-5, atol=1e-4, name='project: pixels')
    assert_close(dj, d['proj_depth'], rtol=1e-5, atol=1e-6, name='project: depth')
    assert [bg_losses.nearest_pose_id(Pc2w, i) for i in range(int(d['V']))] == [int(x) for x in d['nearest']]
    virt = bg_losses.sample_virtual_pose(Pc2w, 2, P[2], float(d['w']))
    assert_close(virt, d['virtual_w2c'], rtol=1e-5, atol=1e-6, name='virtual pose')
