"""Scene-branch losses on top of the HIP render path (per-ray torch algebra on [N,2] / [N] tensors; every rendered quantity
comes from `bg_nerf.SceneRenderer`, i.e. from the kernels):

  * correspondence loss of SPARF as PoseProbe uses it (lib/bg_nerf/source/training/core/corres_loss.py:93-227): depths are
    rendered at matched pixels of an image pair, each pixel is re-projected into the other image with the current poses, and
    the confidence-weighted Huber distance to its match is averaged over both directions (and over the coarse / fine passes);
  * the helpers it needs - `pose_inverse_4x4` (utils/camera.py:37-66), `project_to_other_img`
    (utils/geometry/batched_geometry_utils.py:199-228), `compute_diff_loss` (training/core/base_losses.py:197-224).

  * depth-consistency loss (training/core/depth_cons_loss.py:128-330): pseudo ground-truth 3D points are back-projected from
    the depth rendered in a training view, projected into a virtual view interpolated between two training poses, kept
    where they are visible there (accumulated transmittance up to their depth >= 0.2), and the depth rendered in the virtual
    view is pulled towards theirs with a visibility-weighted Huber loss (coarse and fine pass).
"""
import torch

from . import bg_nerf


def pose_inverse_4x4(mat):
    """[4,4] or [B,4,4] rigid transform -> its inverse without a matrix inversion."""
    R, t = mat[..., :3, :3], mat[..., :3, 3:]
    Rinv = R.transpose(-1, -2)
    out = torch.zeros_like(mat)
    out[..., :3, :3] = Rinv
    out[..., :3, 3:] = -Rinv @ t
    out[..., 3, 3] = 1
    return out


def project_to_other_img(kpi, di, Ki, Kj, T_itoj):
    """Pixels kpi [N,2] with depths di [N] of image i -> their pixel positions [N,2] and depths [N] in image j."""
    hom = torch.cat([kpi, torch.ones_like(kpi[..., :1])], dim=-1)
    p_i = (hom @ torch.inverse(Ki).transpose(-1, -2)) * di[..., None]
    p_j4 = torch.cat([p_i, torch.ones_like(p_i[..., :1])], dim=-1) @ T_itoj.transpose(-1, -2)
    p_j = p_j4[..., :-1] / (p_j4[..., -1:] + 1e-6)
    px = p_j @ Kj.transpose(-1, -2)
    return px[..., :-1] / (px[..., -1:] + 1e-6), p_j[..., -1]


def compute_diff_loss(loss_type, diff, weights=None, mask=None, dim=-1):
    t = loss_type.lower()
    if t == 'epe':
        loss = torch.norm(diff, 2, dim, keepdim=True)
    elif t == 'l1':
        loss = diff.abs()
    elif t == 'mse':
        loss = diff ** 2
    elif t == 'huber':
        loss = torch.nn.functional.huber_loss(diff, torch.zeros_like(diff), reduction='none', delta=1.)
    else:
        raise ValueError('Wrong loss type: {}'.format(loss_type))
    if weights is not None:
        loss = loss * weights
    if mask is not None:
        loss = loss * mask.float()
        return loss.sum() / (mask.float().sum() + 1e-6)
    return loss.sum() / (loss.nelement() + 1e-6)


def reprojection_loss(opt, pixels_self, depth_self, K_self, pixels_other, depth_other, K_other, T_self2other, conf, stats=None):
    """One direction of the loss (corres_loss.py:93-138) with the optional pixel / depth consistency filters."""
    stats = {} if stats is None else stats
    proj, depth_in_other = project_to_other_img(pixels_self.float(), depth_self, K_self, K_other, T_self2other)
    err = torch.norm(proj - pixels_other, dim=-1, keepdim=True)
    valid = torch.ones_like(err).bool()
    if getattr(opt, 'renderrepro_do_pixel_reprojection_check', False):
        ok = err.detach().le(opt.renderrepro_pixel_reprojection_thresh)
        valid = valid & ok
        stats['perc_val_pix_rep'] = ok.sum().float() / (ok.nelement() + 1e-6)
    if getattr(opt, 'renderrepro_do_depth_reprojection_check', False):
        rel = (torch.abs(depth_other - depth_in_other) / (depth_other + 1e-6)).detach().le(opt.renderrepro_depth_reprojection_thresh)
        valid = valid & rel.unsqueeze(-1)
        stats['perc_val_depth_rep'] = rel.sum().float() / (rel.nelement() + 1e-6)
    loss = compute_diff_loss(getattr(opt, 'diff_loss_type', 'huber'), proj - pixels_other, weights=conf, mask=valid)
    return loss, stats


def correspondence_loss(renderer, opt, poses_w2c, intr, pixels_self, pixels_other, conf, H, W, depth_range, iteration=None,
                        rand=None):
    """poses_w2c [2,3,4] and intr [2,3,3] of the (self, other) pair; matched pixels [N,2] each, confidences [N,1].
    Returns (loss, stats, render dict) - corres_loss.py:140-227 for one pair."""
    rets = renderer.render(opt, poses_w2c, H, W, intr, pixels=torch.stack([pixels_self, pixels_other]), depth_range=depth_range,
                           iter=iteration, mode='train', rand=rand)
    bottom = poses_w2c.new_tensor([[0., 0., 0., 1.]])
    P_self, P_other = torch.cat([poses_w2c[0], bottom]), torch.cat([poses_w2c[1], bottom])
    T = P_other @ pose_inverse_4x4(P_self)
    Tinv = pose_inverse_4x4(T)
    stats = {'depth_in_corr_loss': rets['depth'][0].detach().mean()}
    total, passes = 0., 0
    for key in ('depth', 'depth_fine'):
        if key not in rets:
            continue
        d_self, d_other = rets[key][0].squeeze(-1), rets[key][1].squeeze(-1)
        a, stats = reprojection_loss(opt, pixels_self, d_self, intr[0], pixels_other, d_other, intr[1], T, conf, stats)
        b, stats = reprojection_loss(opt, pixels_other, d_other, intr[1], pixels_self, d_self, intr[0], Tinv, conf, stats)
        total = total + a + b
        passes += 2
    return total / passes, stats, rets


# ------------------------------------------------------------------------------------------------ depth consistency
def backproject_to_3d(kpi, di, Ki, T_itoj):
    """Pixels [N,2] with depths [N] of camera i -> 3D points [N,3] in frame j (batched_geometry_utils.py:231-248)."""
    hom = torch.cat([kpi, torch.ones_like(kpi[..., :1])], dim=-1)
    p_i = (hom @ torch.inverse(Ki).transpose(-1, -2)) * di[..., None]
    p_j = torch.cat([p_i, torch.ones_like(p_i[..., :1])], dim=-1) @ T_itoj.transpose(-1, -2)
    return p_j[..., :-1] / (p_j[..., -1:] + 1e-6)


def project(pts, T_itoj, Kj):
    """3D points [N,3] of frame i -> pixels [N,2] and depths [N] in camera j (batched_geometry_utils.py:251-266)."""
    p_j4 = torch.cat([pts, torch.ones_like(pts[..., :1])], dim=-1) @ T_itoj.transpose(-1, -2)
    p_j = p_j4[..., :-1] / (p_j4[..., -1:] + 1e-6)
    px = p_j @ Kj.transpose(-1, -2)
    return px[..., :-1] / (px[..., -1:] + 1e-6), p_j[..., -1]


def nearest_pose_id(poses_c2w, id_self, scene_center=(0., 0., 0.)):
    """Training view whose camera position subtends the smallest angle with view `id_self`'s, seen from the scene centre
    (datasets/data_utils.py:267-311, angular_dist_method 'vector')."""
    c = torch.as_tensor(scene_center, dtype=poses_c2w.dtype, device=poses_c2w.device)
    v = poses_c2w[:, :3, 3] - c
    v = v / (v.norm(dim=-1, keepdim=True) + 1e-6)           # TINY_NUMBER of the reference (datasets/data_utils.py:27)
    ang = torch.arccos((v * v[id_self]).sum(-1).clamp(-1.0, 1.0))
    ang[id_self] = 1e3
    return int(torch.argmin(ang))


def sample_virtual_pose(poses_c2w, id_self, pose_w2c_self, w):
    """Element-wise interpolation (weight w on the reference view) between the reference view's and its nearest neighbour's
    camera-to-world matrices, inverted like a rigid transform - exactly what the reference does (depth_cons_loss.py:45-63)."""
    id_other = nearest_pose_id(poses_c2w.detach(), id_self)
    c2w = w * pose_inverse_4x4(pose_w2c_self).detach() + (1 - w) * poses_c2w[id_other].detach()
    return pose_inverse_4x4(c2w)


def depth_consistency_loss_at_pose(renderer, opt, pose_w2c_unseen, intr_unseen, pts3d_w, H, W, depth_min, iteration=None):
    """Core of the loss (depth_cons_loss.py:232-330).  pose_w2c_unseen [4,4], pts3d_w [N,3] pseudo ground truth in the world
    frame.  Returns (loss, stats); a zero loss (with grad) when no point survives the image-bounds / visibility filters."""
    zero = lambda: (pts3d_w.new_zeros((), requires_grad=True), {})
    px, gt_depth = project(pts3d_w, pose_w2c_unseen, intr_unseen)
    ok = px[:, 0].ge(0.) & px[:, 1].ge(0.) & px[:, 0].le(W - 1) & px[:, 1].le(H - 1) & gt_depth.ge(depth_min)
    px, gt_depth = px[ok], gt_depth[ok]
    if gt_depth.shape[0] == 0:
        return zero()
    pose34, K = pose_w2c_unseen[None, :3], intr_unseen[None]
    with torch.no_grad():                                   # visibility: transmittance accumulated up to the pseudo depth
        vis = renderer.render_up_to_maxdepth(opt, pose34, H, W, K, gt_depth[None], depth_min, px[None], iter=iteration,
                                             mode='train')
        vis_w = (vis['all_cumulated_fine'] if 'all_cumulated_fine' in vis else vis['all_cumulated']).squeeze(0).unsqueeze(-1)
    keep = vis_w.ge(0.2).reshape(-1)
    px, gt_depth, vis_w = px[keep], gt_depth[keep], vis_w[keep]
    stats = {'nbr_px_sampling': int(pts3d_w.shape[0])}
    if gt_depth.shape[0] == 0:
        return zero()
    depth_range = (getattr(renderer, 'depth_range', None) or (depth_min, float(gt_depth.max()) * 1.5))
    ret = renderer.render(opt, pose34, H, W, K, pixels=px[None], depth_range=depth_range, iter=iteration, mode='train')
    loss = 0.
    for key_d, key_o in (('depth', 'opacity'), ('depth_fine', 'opacity_fine')):
        if key_d not in ret:
            continue
        w = vis_w * ret[key_o].squeeze(0).detach()
        loss = loss + compute_diff_loss(getattr(opt, 'diff_loss_type', 'huber'), gt_depth.view(-1) - ret[key_d].reshape(-1),
                                        weights=w.view(-1))
        stats['avg_vis_weight'] = w.sum() / (w.nelement() + 1e-6)
    return loss, stats


def depth_consistency_loss(renderer, opt, poses_w2c, intr, H, W, depth_range, iteration, id_self, pixels_ref, w):
    """One evaluation of the loss as the trainer runs it (depth_cons_loss.py:128-230): view `id_self` supplies the pseudo
    ground truth at `pixels_ref` [N,2] (poses are detached for it, as in the reference), `w` in [0,1] places the virtual view.
    poses_w2c [B,3,4], intr [B,3,3]."""
    renderer.depth_range = tuple(depth_range)
    B = poses_w2c.shape[0]
    bottom = poses_w2c.new_tensor([[[0., 0., 0., 1.]]]).repeat(B, 1, 1)
    P = torch.cat([poses_w2c.detach(), bottom], dim=1)
    P_c2w = pose_inverse_4x4(P)
    ret = renderer.render(opt, P[id_self][None, :3], H, W, intr[id_self][None], pixels=pixels_ref[None], depth_range=depth_range,
                          iter=iteration, mode='train')
    start = getattr(opt.nerf, 'ratio_start_fine_sampling_at_x', None)
    use_fine = 'depth_fine' in ret and not (start is not None and iteration < opt.max_iter * (start + 0.05))
    depth_ref = (ret['depth_fine'] if use_fine else ret['depth']).squeeze(0).squeeze(-1)
    pts3d = backproject_to_3d(pixels_ref, depth_ref, intr[id_self], P_c2w[id_self])
    pose_unseen = sample_virtual_pose(P_c2w, id_self, P[id_self], w)
    return depth_consistency_loss_at_pose(renderer, opt, pose_unseen, intr[id_self].clone(), pts3d, H, W, depth_range[0], iteration)
