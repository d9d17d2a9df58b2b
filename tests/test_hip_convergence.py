"""End-to-end behaviour of the fused trainer (not a parity test): a "teacher" model renders three 96x96 views of the
cube-init SDF with smooth colours; a "student" with a different colour grid / colour MLP and perturbed camera poses is
trained on those images with poseprobe_amd.engine.TrainEngine.  The photometric loss must fall clearly (measured: 0.03 -> 0.002..0.008
per step), everything must stay finite and the free poses must move - i.e. forward, backward (incl. the 6-DoF pose gradient), TV + Adam and
the lr / c2f schedules work together over hundreds of steps, which no single-step comparison shows.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
LAST = None                 # (first, last) photometric loss of the most recent run, for tools/dbg/conv_loop.py


def _render_views(eng, V, H, W, step):
    """Full images through the training forward (no jitter), N rays at a time -> [V,H,W,3] and cum_weights [V,H,W]."""
    from poseprobe_amd import ops
    N = eng.N
    out = torch.zeros(V * H * W, 3, device='cuda')
    acc = torch.zeros(V * H * W, device='cuda')
    cfg, ws, sc = eng.cfg, eng.ws, eng.cfg.pp
    jitter = torch.zeros(N, device='cuda')
    ops.pose_fwd(eng.se3, eng.w2c_init, eng.refine_mask, eng.w2c, eng.c2w, eng.jac)
    for b in range(0, V * H * W, N):
        idx = torch.arange(b, b + N, dtype=torch.int32, device='cuda') % (V * H * W)
        ops.raygen_select_fwd(sc, idx, eng.c2w, eng.intr, H, W, cfg.inverse_y, True, eng.images, eng.masks, ws.rays_o,
                              ws.rays_d, ws.viewdirs, ws.target, ws.mask_px)
        eng.core.sample(ws, jitter)
        eng._upload_step_scalars(step / cfg.N_iters)
        inv_s = float(np.float32(1.0) / np.float32(cfg.s_val(step)))
        P = eng.flat
        eng.core.forward(ws, eng.k0_cl, eng.sdf, P.view('sdf_ab'), P.view('rgbnet'), P.view('warp'), inv_s, eng.pe_w)
        n = min(N, V * H * W - b)
        out[b:b + n] = ws.rgb_marched[:n]
        acc[b:b + n] = ws.cum_weights[:n]
    torch.cuda.synchronize()
    return out.view(V, H, W, 3), acc.view(V, H, W)


def test_student_converges_to_teacher_images_and_poses():
    from poseprobe_amd import synthetic as syn
    from poseprobe_amd.engine import SceneConfig, TrainEngine
    from poseprobe_amd.params_init import reference_like_params
    G, H, W, V, N = 48, 96, 96, 3, 1024
    rs = syn.range_shape()
    Ks = syn.intrinsics(V, H, W)
    w2c = syn.cameras(V)

    def engine(seed, se3):
        cfg = SceneConfig(syn.XYZ_MIN, syn.XYZ_MAX, G ** 3, out_range=float(rs.max()))
        e = TrainEngine(cfg, V, H, W, N, pose_iters=300)
        e.set_views(np.zeros((V, H, W, 3), np.float32), np.ones((V, H, W, 1), np.float32), Ks, w2c)
        P = reference_like_params(cfg, seed)
        # sdf_alpha = 0.637 instead of the reference's 10: the mapped cube SDF then has a gradient norm of O(1), i.e. the
        # eikonal prior agrees with the teacher's geometry (with 10 it is ~15 and the prior - correctly - reshapes the
        # surface away from the teacher within a few dozen steps)
        e.load_reference_params(P['k0'], P['sdf'], torch.tensor([0.637]), P['sdf_beta'], P['rgbnet'], P['warp'],
                                se3=torch.tensor(se3))
        return e

    teacher = engine(3, np.zeros((V, 6), np.float32))
    # a smooth colour field: low-frequency k0
    with torch.no_grad():
        x = torch.linspace(-1, 1, G, device='cuda')
        gx, gy, gz = torch.meshgrid(x, x, x, indexing='ij')
        for c in range(12):
            teacher.k0_cl[..., c] = 0.8 * torch.sin((c % 3 + 1) * gx + 0.5 * c) * torch.cos((c % 4) * gy) + 0.3 * gz
    step = 2000
    img, acc = _render_views(teacher, V, H, W, step)
    assert float(acc.max()) > 0.9 and float(img.std()) > 0.02, 'teacher views are empty'

    se3_0 = syn.se3_perturbation(V, std=2e-2, seed=5)
    student = engine(11, se3_0)
    student.set_views(img.cpu().numpy(), (acc > 0.5).float().unsqueeze(-1).cpu().numpy(), Ks, w2c)
    student.zero_grads()
    first, last = [], []
    for s in range(600):
        idx, jit = syn.step_randomness(V * H * W, N, seed=900 + s)
        student.train_step(torch.tensor(idx, dtype=torch.int32, device='cuda'), torch.tensor(jit, device='cuda'), step + s)
        if s < 3 or s >= 500:
            (first if s < 3 else last).append(student.losses()['img_render'])
    torch.cuda.synchronize()
    l0, l1 = float(np.median(first)), float(np.median(last))         # per-step batches differ: medians
    global LAST
    LAST = (l0, l1)
    # typical: 0.030 -> 0.0015..0.0045 (unordered float atomics make the trajectory run-to-run different); about 2 % of the runs -
    # with the fp32 and with the split-precision MLP kernels alike, tools/dbg/conv_loop.py: 2/80 and 1/80 - settle in a second
    # basin at 0.0150..0.0160
    assert np.isfinite(l1) and l1 < 0.6 * l0, f'photometric loss {l0:.4e} -> {l1:.4e}'
    # view 0 is never refined (recon_scene.py:68); the other two do move (their optimum is not unique: the deformation
    # network can absorb a rigid motion, so no claim about the direction is made here)
    se3 = student.se3.detach().cpu().numpy()
    assert np.allclose(se3[0], se3_0[0])
    assert np.isfinite(se3).all() and np.abs(se3[1:] - se3_0[1:]).max() > 1e-4
    assert bool(torch.isfinite(student.k0_cl).all()) and bool(torch.isfinite(student.flat.data).all())
