"""Index arithmetic with X != Y != Z: the golden fixtures are cubic grids, so a non-cubic bounding box (world_size
[24, 18, 30]-like, as real scenes produce, lib/voxurf_coarse.py:96-104) is checked here against the oracle directly: one full
step - sampler indices bit-exact, pixels / losses / gradients (incl. the dense k0 gradient) within the fp32 tolerances -
followed by the fused TV + Adam pass against the oracle's optimiser."""
import numpy as np
import pytest
import torch

from tests.helpers import assert_close

pytestmark = pytest.mark.gpu


def test_noncubic_grid_step_against_the_oracle():
    from oracle import voxurf_oracle as O
    from poseprobe_amd import synthetic as syn
    from poseprobe_amd.engine import SceneConfig, TrainEngine
    from poseprobe_amd.params_init import reference_like_params
    xyz_min = np.array([-0.6, -0.45, -0.7], dtype=np.float32)
    xyz_max = np.array([0.6, 0.45, 0.8], dtype=np.float32)
    nvox = 24 * 18 * 30
    H, W, V, N = 40, 48, 3, 192
    rs = syn.range_shape()
    cfg = SceneConfig(xyz_min, xyz_max, nvox, out_range=float(rs.max()))
    X, Y, Z = cfg.world_size
    assert len({X, Y, Z}) == 3, cfg.world_size
    views = syn.make_views(V, H, W)
    eng = TrainEngine(cfg, V, H, W, N, pose_iters=1000)
    eng.set_views(views['images'], views['masks'], views['Ks'], views['w2c'])
    P0 = reference_like_params(cfg, 3)
    eng.load_reference_params(P0['k0'], P0['sdf'], P0['sdf_alpha'], P0['sdf_beta'], P0['rgbnet'], P0['warp'],
                              se3=torch.tensor(syn.se3_perturbation(V)))
    eng.zero_grads()
    idx, jit = syn.step_randomness(V * H * W, N, seed=21)
    eng.render_and_grads(torch.tensor(idx, dtype=torch.int32, device='cuda'), torch.tensor(jit, device='cuda'), 10)
    torch.cuda.synchronize()

    scene = O.Scene(xyz_min, xyz_max, nvox, output_range=float(rs.max()), rect_size=rs.tolist())
    assert list(scene.world_size) == [X, Y, Z]
    P = O.params_require_grad(reference_like_params(cfg, 3))
    se3 = torch.tensor(syn.se3_perturbation(V), requires_grad=True)
    c2w = O.pose_invert(O.current_pose_pnp(se3, torch.tensor(views['w2c'])))
    ro, rd, vd, target, mask = O.select_training_rays(torch.tensor(idx).long(), torch.tensor(views['images']),
                                                      torch.tensor(views['masks']), torch.tensor(views['Ks']), c2w)
    out = O.voxurf_forward(P, scene, ro, rd, vd, jitter=torch.tensor(jit), global_step=10)
    S, Wt, loss = O.object_losses(out, target, mask, 10, scene.N_iters)
    (loss * 0.1).backward()
    c = lambda t: t.detach().cpu().numpy()
    M = int(eng.ws.count.item())
    assert M == out['weights'].shape[0] and M > 500
    assert np.array_equal(c(eng.ws.ray_id[:M]), c(out['_ray_id']))
    assert_close(c(eng.ws.rgb_marched), c(out['rgb_marched']), rtol=1e-4, atol=1e-5, name='rgb_marched')
    L = eng.losses()
    for k in ('img_render', 'grad_constraint', 'grad_deform_constraint', 'sdf_deform_constraint', 'mask_render'):
        assert_close(np.float32(L[k]), c(S[k]), rtol=2e-4, atol=1e-7, name='loss.' + k)
    tol = dict(rtol=1e-3, scaled=5e-5)
    assert_close(c(eng.se3_grad), c(se3.grad), atol=1e-6, name='g.se3', **tol)
    # dense colour-grid gradient (data part; the TV term is added by the optimiser kernel): reference layout [1,C,X,Y,Z]
    gk0 = c(eng.k0_grad.permute(3, 0, 1, 2)[None])
    ref = c(P['k0'].grad) - c(_tv_grad(P['k0'].detach(), 0.1 * 0.01))
    assert_close(gk0, ref, atol=1e-8, name='g.k0', **tol)
    marked = c(eng.k0_touched[eng.touch_par].view(X, Y, Z))
    assert ((np.abs(ref).sum(axis=(0, 1)) > 0) <= (marked != 0)).all(), 'a voxel with gradient was not marked'


def _tv_grad(k0, scale):
    from oracle import voxurf_oracle as O
    p = k0.clone().requires_grad_(True)
    (O.total_variation(p) * scale).backward()
    return p.grad
