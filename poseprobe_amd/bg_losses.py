"""Scene-branch losses on top of the HIP render path (per-ray torch algebra on [N,2] / [N] tensors; every rendered quantity
comes from `bg_nerf.SceneRenderer`, i.e. from the kernels):

  * correspondence loss of SPARF as PoseProbe uses it (lib/bg_nerf/source/training/core/corres_loss.py:93-227): depths are
    rendered at matched pixels of an image pair, each pixel is re-projected into the other image with the current poses, and
    the confidence-weighted Huber distance to its match is averaged over both directions (and over the coarse / fine passes);
  * the helpers it needs - `pose_inverse_4x4` (utils/camera.py:37-66), `project_to_other_img`
    (utils/geometry/batched_geometry_utils.py:199-228), `compute_diff_loss` (training/core/base_losses.py:197-224).

The depth-consistency loss (training/core/depth_cons_loss.py) needs renderings from virtual viewpoints chosen by the SPARF
trainer and is not part of this module.
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
