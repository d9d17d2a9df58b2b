"""Trainer-side geometry around the surface-point queries: pixel -> ray for arbitrary (sub-pixel) image points, point /
ray distances and the reprojection + near-surface losses that PoseProbe adds to the object loss while poses are being
optimised.  Pure torch on device tensors; the surface points themselves come from the HIP path
(`Voxurf.query_sdf_point_wocuda_render` / `_wodeform`).

Reference behaviour mirrored (not its code): lib/recon_scene.py:93-113 (get_ray_dir), :313-319 (point_to_ray_distance),
:321-369 (get_project_error); lib/camera.py:251-253 (world2cam); lib/common.py:450-465 (project_to_cam_real);
lib/losses.py:77-103 (compute_diff_loss).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import camera


def get_ray_dir(points, K, c2w, inverse_y, flip_x, flip_y, mode='center'):
    """points [B,N,2] pixel coordinates (x, y), K [B,3,3], c2w [B,3,4] -> rays_o, rays_d [B,N,3] (rays_d normalised).
    mode='center' shifts by half a pixel; flips reverse the point order along N (x) / B (y) like the reference does."""
    pts = points + 0.5 if mode == 'center' else points
    x, y = pts[..., 0], pts[..., 1]
    if flip_x:
        x = x.flip((1,))
    if flip_y:
        y = y.flip((0,))
    fx, fy, cx, cy = K[:, 0, 0:1], K[:, 1, 1:2], K[:, 0, 2:3], K[:, 1, 2:3]
    u, v = (x - cx) / fx, (y - cy) / fy
    one = torch.ones_like(u)
    cam = torch.stack([u, v, one], -1) if inverse_y else torch.stack([u, -v, -one], -1)
    d = (cam[..., None, :] * c2w[:, None, :3, :3]).sum(-1)
    d = d / d.norm(dim=-1, keepdim=True)
    o = c2w[:, None, :3, 3].expand(d.shape)
    return o, d


def point_to_ray_distance(ray_origins, ray_directions, point):
    """Distance of `point` [3] to each half-line o + t d, t >= 0  ([M,3] unit directions) -> [M]."""
    rel = point - ray_origins
    t = (rel * ray_directions).sum(1)
    foot = ray_origins + t[:, None] * ray_directions
    return torch.where(t < 0, rel.norm(dim=1), (point - foot).norm(dim=1))


def world2cam(X, pose):
    """X [B,N,3] world points, pose [B,3,4] w2c -> camera coordinates [B,N,3]."""
    return torch.cat([X, torch.ones_like(X[..., :1])], -1) @ pose.transpose(-1, -2)


def project_to_cam_real(points, camera_mat, HW=None):
    """Pinhole projection of camera-frame points [B,N,3] with intrinsics [B,3,3] -> pixels [B,N,2]."""
    uvw = (camera_mat @ points.transpose(1, 2)).transpose(1, 2)
    return uvw[..., :2] / uvw[..., 2:]


def compute_diff_loss(loss_type, diff, weights=None, var=None, mask=None, dim=-1, delta=1.):
    kind = loss_type.lower()
    if kind == 'epe':
        loss = torch.norm(diff, 2, dim, keepdim=True)
    elif kind == 'l1':
        loss = diff.abs()
    elif kind == 'mse':
        loss = diff ** 2
    elif kind == 'huber':
        loss = F.huber_loss(diff, torch.zeros_like(diff), reduction='none', delta=delta)
    else:
        raise ValueError('Wrong loss type: {}'.format(loss_type))
    if weights is not None:
        assert weights.dim() == loss.dim()
        loss = loss * weights
    if var is not None:
        v = torch.maximum(var, torch.tensor(1e-3, device=var.device))
        loss = loss / v + torch.log(v)
    if mask is not None:
        assert mask.dim() == loss.dim()
        m = mask.float()
        return (loss * m).sum() / (m.sum() + 1e-6)
    return loss.sum() / (loss.nelement() + 1e-6)


def get_project_error(model, Ks, HW, nl, global_step, current_pose, coord0, coord1, i_train, j_train, mconf, inverse_y=True,
                      flip_x=False, flip_y=False, use_deform=True, pixel_thre=None, **render_kwargs):
    """Symmetric reprojection error of matched pixels through the current surface, plus the near-surface prior.

    coord0 / coord1 [P,N,2]: matches between views i_train[p] and j_train[p]; mconf [P,N] match confidences; current_pose
    [V,3,4] w2c.  Every pixel of either view is lifted to its surface point (expected ray depth), moved into the OTHER view
    and projected; the Huber distance to the matched pixel is averaged over valid points (in front of the near plane `nl`,
    hit by the query, closer than pixel_thre).  near_surface: rays that pass the box centre* further away than half the
    (reference-defined) diagonal are pushed back.  (*: `xyz_min + xyz_max`, as the reference writes it.)"""
    dev = current_pose.device
    coord = torch.cat([coord0, coord1], 0)
    own = np.concatenate([i_train, j_train], 0)
    other = np.concatenate([j_train, i_train], 0)
    conf = torch.cat([mconf, mconf], 0)
    o, d = get_ray_dir(coord, Ks[own], c2w=camera.pose.invert(current_pose[own]), inverse_y=inverse_y, flip_x=flip_x,
                       flip_y=flip_y, mode='no_center')
    o, d = o.reshape(-1, 3), d.reshape(-1, 3)
    rk = dict(render_kwargs, inverse_y=inverse_y, flip_x=flip_x, flip_y=flip_y)
    if use_deform:
        pts, hit, _ = model.query_sdf_point_wocuda_render(o, d, global_step=global_step, use_deform=True, keep_dim=True, **rk)
    else:
        pts, hit, _ = model.query_sdf_point_wocuda_wodeform(o, d, global_step=global_step, keep_dim=True, **rk)
    centre = model.xyz_min + model.xyz_max
    off = point_to_ray_distance(o, d, point=centre.to(dev))
    near_surface = (torch.clamp(off - model.diagonal_length.to(dev) / 2., min=0.0) * (conf.flatten() > 0)).sum()

    P = len(own)
    pts, hit = pts.reshape(P, -1, 3), hit.reshape(P, -1)
    cam = world2cam(pts, current_pose[other])
    depth = cam[..., 2:] if inverse_y else -cam[..., 2:]
    behind = (depth < nl).expand_as(cam)
    cam = torch.where(behind, torch.full_like(cam, float(nl)), cam)
    px = project_to_cam_real(cam, Ks[other])
    if not inverse_y:
        px = torch.stack([float(np.asarray(HW)[0, 1]) - px[..., 0], px[..., 1]], -1)
    dist = torch.norm(px - torch.cat([coord1, coord0], 0), p=2, dim=-1)
    valid = (~behind[..., 0]) & hit.bool()
    if pixel_thre is not None:
        valid = valid & (dist.detach() <= pixel_thre)
    get_project_error.last_valid = valid            # diagnostics for callers / tests (how many matches took part)
    return compute_diff_loss('huber', dist, weights=conf, mask=valid, delta=1.), near_surface
