"""Dual-branch step (poseprobe_amd.joint): both branches share the poses; se3 receives the sum of their gradients
(loss = 0.1 * L_obj + L_bg, lib/recon_scene.py:645-649)."""
import numpy as np
import pytest
import torch

from tests.helpers import assert_close, load
from tests.test_hip_step import build_engine

pytestmark = pytest.mark.gpu


def _scene_batch(d, V, N, S, seed=3):
    g = torch.Generator().manual_seed(seed)
    H, W = int(d['H']), int(d['W'])
    pixels = (torch.rand(N, 2, generator=g) * torch.tensor([W - 1., H - 1.])).cuda()
    image = torch.rand(V, N, 3, generator=g).cuda()
    rand = torch.rand(V, N, S, 1, generator=g).cuda()
    return pixels, image, rand


def test_joint_step_pose_gradient_is_the_sum_of_both_branches():
    from poseprobe_amd import bg_nerf, camera
    from poseprobe_amd.joint import DualBranchEngine
    d = load('forward_g24_s10.npz')
    ray_idx = torch.tensor(d['ray_idx'], dtype=torch.int32, device='cuda')
    jitter = torch.tensor(d['jitter'], device='cuda')
    gs = int(d['global_step'])
    opt = bg_nerf.default_options(sample_intvs=24)
    torch.manual_seed(5)
    net = bg_nerf.NeRF(opt, device='cuda')
    net.progress.data.fill_(0.6)
    V, N, S = 3, 40, 24
    pixels, image, rand = _scene_batch(d, V, N, S)

    ref, _ = build_engine(d)                       # object branch alone
    ref.zero_grads()
    ref.render_and_grads(ray_idx, jitter, gs)
    g_obj = ref.se3_grad.clone()
    flat_obj, k0_obj = ref.flat.grad.clone(), ref.k0_grad.clone()

    eng, _ = build_engine(d)
    joint = DualBranchEngine(eng, net, depth_range=(0.5, 3.0))
    eng.zero_grads()
    _, loss_bg = joint.forward_backward(ray_idx, jitter, gs, pixels, image, depth_rand=rand)
    assert_close(eng.flat.grad, flat_obj, rtol=1e-4, scaled=1e-5, name='object MLP grads untouched by the scene branch')
    assert_close(eng.k0_grad.sum(), k0_obj.sum(), rtol=1e-4, name='k0 grad')

    # the scene branch's share by autograd: se3 -> (HIP pose chain) -> c2w -> rays -> NeRF autograd nodes -> loss
    se3 = eng.se3.detach().clone().requires_grad_(True)
    _, c2w = camera.current_pose_c2w(se3, eng.w2c_init, fix_first=True)
    fx, fy, cx, cy = (eng.intr[:, i][:, None] for i in range(4))
    x, y = pixels[None, :, 0], pixels[None, :, 1]
    dir_cam = torch.stack([(x - cx) / fx, (y - cy) / fy, torch.ones(V, N, device='cuda')], -1)
    ray = dir_cam @ c2w[:, :, :3].transpose(-1, -2)
    center = c2w[:, None, :, 3].expand_as(ray)
    depth = (rand + torch.arange(S, device='cuda')[None, None, :, None]) / S * 2.5 + 0.5
    net2 = bg_nerf.NeRF(opt, device='cuda')
    net2.load_state_dict(net.state_dict())
    pred = net2.composite(opt, ray, net2.forward_samples(opt, center, ray, depth, mode='train'), depth)
    loss = bg_nerf.photometric_loss(pred['rgb'], image)
    loss.backward()
    assert_close(loss_bg, loss, rtol=1e-5, name='scene loss')
    assert_close(eng.se3_grad - g_obj, se3.grad, rtol=1e-3, scaled=2e-3, name='scene share of the pose gradient')
    assert float((eng.se3_grad - g_obj).abs().max()) > 0
    # the scene engine's parameter gradients equal the autograd node's
    views = net2._views(joint.scene.grad)
    for (name, p), gv in zip([(n, p) for n, p in net2.named_parameters() if n != 'progress'], views):
        assert_close(gv, p.grad, rtol=1e-4, scaled=2e-5, name='scene g.' + name)


def test_joint_train_steps_run_and_reduce_both_losses():
    from poseprobe_amd import bg_nerf
    from poseprobe_amd.joint import DualBranchEngine
    d = load('forward_g24_s10.npz')
    eng, _ = build_engine(d)
    opt = bg_nerf.default_options(sample_intvs=24)
    torch.manual_seed(6)
    net = bg_nerf.NeRF(opt, device='cuda')
    net.progress.data.fill_(0.3)
    joint = DualBranchEngine(eng, net, lr_scene=5e-4)
    ray_idx = torch.tensor(d['ray_idx'], dtype=torch.int32, device='cuda')
    jitter = torch.tensor(d['jitter'], device='cuda')
    pixels, image, rand = _scene_batch(d, 3, 64, 24)
    image = image * 0 + torch.tensor([0.2, 0.5, 0.7]).cuda()          # a learnable target: constant colour
    eng.zero_grads()
    se3_before = eng.se3.clone()
    losses = []
    for it in range(30):
        _, loss_bg = joint.train_step(ray_idx, jitter, int(d['global_step']) + it, pixels, image, depth_rand=rand)
        losses.append(float(loss_bg))
    assert np.isfinite(losses).all() and losses[-1] < 0.5 * losses[0], losses
    assert float((eng.se3 - se3_before).abs().max()) > 0           # poses moved (views other than the fixed first one)
    assert float((eng.se3 - se3_before)[0].abs().max()) == 0.0


def test_trainer_counterpart_schedules_and_snapshot(tmp_path):
    """DualBranchTrainer: the fine network and the end of pose refinement start where the schedule says, the coarse-to-fine
    window and the learning rate follow the iteration, and `model_last.pth.tar` round-trips (reference key layout)."""
    from poseprobe_amd import bg_nerf
    from poseprobe_amd.trainer import DualBranchTrainer, scene_lr
    d = load('forward_g24_s10.npz')
    eng, _ = build_engine(d)
    eng.zero_grads()
    opt = bg_nerf.default_options(sample_intvs=16)
    opt.nerf.fine_sampling, opt.nerf.sample_intvs_fine, opt.nerf.rand_rays = True, 16, 96
    torch.manual_seed(2)
    tr = DualBranchTrainer(eng, opt, max_iter=10, ratio_start_fine=0.3, ratio_end_pose=0.3)
    se3_hist = []
    for step in range(6):
        _, loss_bg = tr.train_step(step)
        se3_hist.append(eng.se3.clone())
        assert np.isfinite(float(loss_bg))
    st_c, st_f = tr.joint.scene.states
    assert st_c.steps == 6 and st_f.steps == 3                          # fine network from step 3 = 0.3 * max_iter on
    assert not torch.equal(se3_hist[2], se3_hist[1]) and torch.equal(se3_hist[5], se3_hist[2])   # poses frozen from step 3
    assert abs(float(tr.nerf.progress) - 0.6) < 1e-6 and abs(float(tr.nerf_fine.progress) - 0.6) < 1e-6
    assert abs(float(tr.joint.scene.seg_lr) - scene_lr(6, 1e-3, 1e-4, 10)) < 1e-9
    tr.save_snapshot(str(tmp_path))
    ck = torch.load(str(tmp_path / 'model_last.pth.tar'), map_location='cpu', weights_only=True)
    assert set(ck) >= {'current_pose', 'iteration', 'iteration_nerf', 'state_dict', 'optimizer', 'scheduler'}
    assert ck['state_dict']['nerf.mlp_feat.7.weight'].shape == (257, 256) and ck['state_dict']['nerf_fine.mlp_rgb.0.weight'].shape == (128, 283)
    ref_opt = torch.optim.Adam([torch.nn.Parameter(v.clone()) for k, v in ck['state_dict'].items() if k.startswith('nerf.')], lr=1e-3)
    ref_opt.add_param_group(dict(params=[torch.nn.Parameter(v.clone()) for k, v in ck['state_dict'].items() if k.startswith('nerf_fine.')]))
    ref_opt.load_state_dict(ck['optimizer'])                           # torch accepts it as an Adam state of those two groups

    eng2, _ = build_engine(d)
    tr2 = DualBranchTrainer(eng2, opt, max_iter=10)
    tr2.load_snapshot(str(tmp_path / 'model_last.pth.tar'))
    assert tr2.iteration == 6
    for a, b in zip(tr.joint.scene.states, tr2.joint.scene.states):
        assert torch.equal(a.net.flat, b.net.flat) and torch.equal(a.m, b.m) and torch.equal(a.v, b.v) and a.steps == b.steps
    assert abs(float(tr2.nerf.progress) - 0.6) < 1e-6


def test_joint_step_through_rccl_on_one_rank_matches_plain():
    """The dual-branch step with a DistContext (RCCL, world_size 1): same trajectory as without (atomics budget of the other
    trajectory tests); exercises the scene networks' all-reduce and the averaged Adam scale."""
    import os
    import socket
    import torch.distributed as dist
    from poseprobe_amd import bg_nerf
    from poseprobe_amd.dist import DistContext
    from poseprobe_amd.joint import DualBranchEngine
    if not dist.is_initialized():
        s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda:0'))
    d = load('forward_g8_s10.npz')
    ray_idx = torch.tensor(d['ray_idx'], dtype=torch.int32, device='cuda')
    jitter = torch.tensor(d['jitter'], device='cuda')
    opt = bg_nerf.default_options(sample_intvs=16)
    pixels, image, rand = _scene_batch(d, 3, 32, 16)
    finals = []
    for ctx in (None, DistContext(mode='samples', resync_every=2)):
        eng, _ = build_engine(d, pose_iters=1000, **({} if ctx is None else {'dist_ctx': ctx}))
        torch.manual_seed(12)
        net = bg_nerf.NeRF(opt, device='cuda')
        net.progress.data.fill_(0.5)
        joint = DualBranchEngine(eng, net, lr_scene=1e-3)
        eng.zero_grads()
        for it in range(3):
            joint.train_step(ray_idx, jitter, int(d['global_step']) + it, pixels, image, depth_rand=rand)
        torch.cuda.synchronize()
        finals.append((eng.se3.clone().cpu(), net.flat.clone().cpu()))
    assert_close(finals[1][0], finals[0][0], rtol=0, atol=2e-4, name='se3')
    assert float((finals[1][1] - finals[0][1]).abs().gt(3e-4).float().mean()) < 0.02


def test_trainer_incremental_views_pose_hand_off_and_loss_mixing():
    """The trainer counterpart's control decisions (lib/recon_scene.py:552-649): views join one by one at multiples of
    `incremental_step`, a joining view starts from a handed-over pose (default = the previous view's current pose, the
    reference's `use_identical`; a callback stands where cv2.solvePnPRansac does), batches only touch active views, and
    extra pose-only terms are mixed into the object loss (scaled by 0.1 with it) before the optimiser step."""
    from poseprobe_amd import bg_nerf
    from poseprobe_amd.trainer import DualBranchTrainer, active_views
    assert [active_views(s, 6, 100) for s in (0, 99, 100, 250, 400, 10 ** 6)] == [2, 2, 3, 4, 6, 6]
    assert active_views(0, 3, 0, incremental=False) == 3
    d = load('forward_g24_s10.npz')
    V, H, W = d['images'].shape[:3]
    opt = bg_nerf.default_options(sample_intvs=16)
    opt.nerf.rand_rays = 96
    calls = []

    def from_pnp(view, prev_w2c):
        calls.append((view, prev_w2c.clone()))
        out = prev_w2c.clone()
        out[:, 3] += torch.tensor([0.01, -0.02, 0.03])
        return out

    def pull_to_zero(se3, w2c_init, k):
        return 0.5, (se3[:k] ** 2).sum()

    for initialiser in (None, from_pnp):
        eng, _ = build_engine(d)
        eng.zero_grads()
        torch.manual_seed(3)
        tr = DualBranchTrainer(eng, opt, max_iter=20, incremental_step=2, pose_initialiser=initialiser, pose_terms=(pull_to_zero,))
        init_before, se3_before = eng.w2c_init.clone(), eng.se3.clone()
        ray_idx, _, pixels, image = tr.sample_batch(2)
        assert int(ray_idx.max()) < 2 * H * W and image.shape[0] == 2 and pixels.shape[0] == 96 // 2
        for step in range(2):
            tr.train_step(step)
        assert tr.n_active == 2 and torch.equal(eng.w2c_init[2], init_before[2]) and torch.equal(eng.se3[2], se3_before[2])
        assert not torch.equal(eng.se3[1], se3_before[1])         # an active, refined view moves
        assert 'pull_to_zero' in tr.last_pose_terms
        from poseprobe_amd import ops
        ops.pose_fwd(eng.se3, eng.w2c_init, eng.refine_mask, eng.w2c, eng.c2w, eng.jac)
        pose1 = eng.w2c[1].clone()
        tr.train_step(2)                                         # view 2 joins
        assert tr.n_active == 3
        if initialiser is None:
            assert_close(eng.w2c_init[2], pose1.cpu(), rtol=0, atol=1e-6, name='identical hand-off')
        else:
            assert calls and calls[-1][0] == 2
            assert_close(calls[-1][1], pose1.cpu(), rtol=0, atol=1e-6, name='previous pose handed to the initialiser')
            assert_close(eng.w2c_init[2][:, 3], (pose1[:, 3].cpu() + torch.tensor([0.01, -0.02, 0.03])), rtol=0, atol=1e-6, name='initialised pose')
        assert float(eng.se3[2].abs().max()) > 0.0               # ... and is refined from its first step on
        # the mixing itself: d(0.1 * 0.5 * |se3|^2) / d se3 on the refined views
        eng.se3_grad.zero_()
        tr._mix_pose_terms(3)
        expect = 0.1 * 0.5 * 2.0 * eng.se3 * eng.refine_mask[:, None]
        assert_close(eng.se3_grad, expect.cpu(), rtol=1e-6, atol=1e-9, name='mixed pose term')
        eng.se3_grad.zero_()


def test_reprojection_term_differentiates_the_zero_crossing_query_while_two_views_are_active():
    """trainer.ReprojectionTerm (lib/recon_scene.py:584, :624-637): with <= 2 active views the surface point is the zero
    crossing of the raw template and its pose gradient flows through pp_sdf_crossing_dense_bwd (BOTH views of the pair
    receive a gradient: the 'own' view through the query's rays, the other through world2cam); with 3 views it is the
    rendered depth.  The mixed gradient equals the term's autograd gradient times loss_scale on the refined views."""
    from poseprobe_amd import bg_nerf, camera, recon_utils
    from poseprobe_amd.trainer import DualBranchTrainer, ReprojectionTerm
    d = load('forward_g24_s10.npz')
    eng, _ = build_engine(d)
    eng.zero_grads()
    H, W = eng.H, eng.W
    g = torch.Generator().manual_seed(11)
    P = 40
    mk = lambda: (torch.rand(P, 2, generator=g) * torch.tensor([W - 1., H - 1.]) * 0.5 + torch.tensor([W, H]) * 0.25)
    pairs = [(0, 1, mk(), mk(), torch.rand(P, generator=g)), (1, 2, mk(), mk(), torch.rand(P, generator=g)),
             (0, 2, mk(), mk(), torch.rand(P, generator=g))]
    term = ReprojectionTerm(eng, pairs, nl=0.1, weight_projection=1.0, weight_near_surface=0.1, seed=0)
    with torch.no_grad():
        eng.refine_mask[0] = 1                  # let view 0 move too, so that BOTH views of the (0, 1) pair show their share
    se3 = eng.se3.detach().clone().requires_grad_(True)
    torch.manual_seed(7)                        # the query draws its per-ray jitter from the global generator (as the reference does)
    w, val = term(se3, eng.w2c_init, 2)
    assert term.last['use_deform'] is False and term.last['pair'] == (0, 1) and w == 1.0
    val.backward()
    g2 = se3.grad.clone()
    assert float(g2[0].abs().max()) > 0 and float(g2[1].abs().max()) > 0 and float(g2[2].abs().max()) == 0
    # the same loss with the surface points detached: only the world2cam half of the gradient is left (what round 2 had)
    i, j, ci, cj, conf = pairs[0]

    class Detached:
        xyz_min, xyz_max, diagonal_length = term.model.xyz_min, term.model.xyz_max, term.model.diagonal_length

        def query_sdf_point_wocuda_wodeform(self, o, dd, **kw):
            p, m, s = term.model.query_sdf_point_wocuda_wodeform(o.detach(), dd.detach(), **kw)
            return p.detach(), m, s

    se3b = eng.se3.detach().clone().requires_grad_(True)
    w2c, _ = camera.current_pose_c2w(se3b, eng.w2c_init, fix_first=False)
    Ks = torch.zeros(3, 3, 3, device='cuda')
    Ks[:, 0, 0], Ks[:, 1, 1], Ks[:, 0, 2], Ks[:, 1, 2], Ks[:, 2, 2] = eng.intr[:, 0], eng.intr[:, 1], eng.intr[:, 2], eng.intr[:, 3], 1.
    torch.manual_seed(7)
    err, near = recon_utils.get_project_error(Detached(), Ks, np.array([[H, W]] * 3), 0.1, 0, w2c, cj[None].cuda(), ci[None].cuda(),
                                              np.array([j]), np.array([i]), conf[None].cuda(), use_deform=False, pixel_thre=200,
                                              near=eng.cfg.near, far=eng.cfg.far, bg=0, stepsize=eng.cfg.stepsize)
    assert_close(np.float32((0.1 * near + 1.0 * err).item()), np.float32(val.item()), rtol=1e-5, atol=1e-7, name='term value')
    (0.1 * near + 1.0 * err).backward()
    assert int(term.last['hits']) >= 10, term.last
    assert float((g2 - se3b.grad).abs().max()) > 0.02 * float(g2.abs().max()), 'the query must contribute to the pose gradient'
    # three active views: rendered-depth query over the engine's live parameters
    se3c = eng.se3.detach().clone().requires_grad_(True)
    term.rng = np.random.RandomState(1)
    w, val3 = term(se3c, eng.w2c_init, 3)
    assert term.last['use_deform'] is True
    val3.backward()
    assert torch.isfinite(se3c.grad).all() and float(se3c.grad.abs().max()) > 0
    # mixing into the engine's pose gradient through the trainer
    opt = bg_nerf.default_options(sample_intvs=16)
    opt.nerf.rand_rays = 96
    term.rng = np.random.RandomState(0)
    tr = DualBranchTrainer(eng, opt, max_iter=20, incremental_step=4, pose_terms=(term,))
    tr.global_step = 0
    eng.se3_grad.zero_()
    torch.manual_seed(7)
    vals = tr._mix_pose_terms(2)
    assert 'reprojection' in vals
    assert_close(eng.se3_grad, (eng.loss_scale * g2 * eng.refine_mask[:, None]).cpu(), rtol=1e-4, atol=1e-7, name='mixed reprojection gradient')
    tr.train_step(0)                                            # and a whole joint step runs with it
    assert term.last['use_deform'] is False
