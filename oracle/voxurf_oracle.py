"""CPU restatement (torch-CPU fp32, autograd for the derivatives) of PoseProbe's object-branch hot path.

TEST INFRASTRUCTURE ONLY.  This module is the *checker*: only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import it; the product (poseprobe_amd/) never does.

It restates, in functional form with all randomness passed in explicitly (ray indices, per-ray
jitter), the algorithm of the following reference code (file:line relative to /root/reference):

  se3 / pose algebra          lib/camera.py:76-99 (Pose), :127-188 (Lie.se3_to_SE3, taylor_A/B/C)
  current pose (PnP mode)     lib/recon_scene.py:62-74
  ray generation              lib/voxurf_coarse.py:1339-1368 (get_rays), :1402-1407 (get_rays_of_a_view)
  dense sampler               lib/voxurf_coarse.py:697-719 (sample_ray_ori), :933-945 (compaction)
  variable-length sampler     lib/voxurf_coarse.py:661-695 (+ lib/cuda/render_utils_kernel.cu:12-242)
  sdf mapping                 lib/voxurf_coarse.py:946-949
  custom trilinear            lib/voxurf_coarse.py:522-543, :545-659
  deformation MLP             lib/deformation/deform_net.py:12-31, modules.py:43-124
  spatial gradients           lib/voxurf_coarse.py:964-984
  NeuS alpha                  lib/voxurf_coarse.py:483-519
  transmittance scan          lib/voxurf_coarse.py:1316-1332 (+ render_utils_kernel.cu:577-707)
  k0 lookup                   lib/grid.py:47-58
  BARF positional encoding    lib/voxurf_coarse.py:721-732, :1009-1025
  rgbnet + compositing        lib/voxurf_coarse.py:1026-1062 ; inference variant :1094-1222
  total variation             lib/voxurf_coarse.py:443-456, :1298-1313
  losses                      lib/losses.py:6-74
  Adam                        lib/utils.py:82-198 (+ group build :316-342)
  DirectVoxGO twin            lib/dvgo_ori.py:242-244, :263-379, :478-485

Parity status: pinned against golden vectors produced by the reference's own Python executed in the
build container (oracle/make_golden.py -> tests/golden/*.npz; tests/test_oracle_vs_golden.py).
Two pieces have no reference-side pin ("parity unpinned"): the compiled CUDA kernels (restated
in oracle/native_ops.c from the .cu text) and torch_scatter.segment_coo (third-party, unpinned;
restated as an ordered index_add_).
"""
import math
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

from . import native_ops


# ----------------------------------------------------------------------------------------------
# SE(3) / pose algebra  (lib/camera.py)
# ----------------------------------------------------------------------------------------------
def _taylor(x, kind, nth=10):
    """A: sin(x)/x, B: (1-cos x)/x^2, C: (x-sin x)/x^3 as 11-term series (camera.py:165-188)."""
    ans = torch.zeros_like(x)
    denom = 1.
    for i in range(nth + 1):
        if kind == 'A':
            if i > 0:
                denom *= (2 * i) * (2 * i + 1)
        elif kind == 'B':
            denom *= (2 * i + 1) * (2 * i + 2)
        else:
            denom *= (2 * i + 2) * (2 * i + 3)
        ans = ans + (-1) ** i * x ** (2 * i) / denom
    return ans


def skew(w):
    w0, w1, w2 = w.unbind(dim=-1)
    O = torch.zeros_like(w0)
    return torch.stack([torch.stack([O, -w2, w1], dim=-1),
                        torch.stack([w2, O, -w0], dim=-1),
                        torch.stack([-w1, w0, O], dim=-1)], dim=-2)


def se3_to_SE3(wu):
    """[...,6] -> [...,3,4]  (camera.py:127-142)."""
    w, u = wu.split([3, 3], dim=-1)
    wx = skew(w)
    theta = w.norm(dim=-1)[..., None, None]
    eye = torch.eye(3, dtype=torch.float32)
    A, B, C = _taylor(theta, 'A'), _taylor(theta, 'B'), _taylor(theta, 'C')
    R = eye + A * wx + B * wx @ wx
    Vm = eye + B * wx + C * wx @ wx
    return torch.cat([R, Vm @ u[..., None]], dim=-1)


def pose_from(R, t):
    return torch.cat([R.float(), t.float()[..., None]], dim=-1)


def pose_invert(pose):
    """camera.py:76-82 (transpose form)."""
    R, t = pose[..., :3], pose[..., 3:]
    R_inv = R.transpose(-1, -2)
    t_inv = (-R_inv @ t)[..., 0]
    return pose_from(R_inv, t_inv)


def pose_compose_pair(pose_a, pose_b):
    """pose_b o pose_a  (camera.py:92-99)."""
    R_a, t_a = pose_a[..., :3], pose_a[..., 3:]
    R_b, t_b = pose_b[..., :3], pose_b[..., 3:]
    return pose_from(R_b @ R_a, (R_b @ t_a + t_b)[..., 0])


def current_pose_pnp(se3_refine, pose_init, fix_first=True):
    """w2c poses [V,3,4]: refine o init, view 0 never refined in PnP mode (recon_scene.py:62-74)."""
    refine = se3_to_SE3(se3_refine)
    composed = pose_compose_pair(refine, pose_init)
    if not fix_first:
        return composed
    return torch.cat([pose_init[:1], composed[1:]], dim=0)


# ----------------------------------------------------------------------------------------------
# rays (lib/voxurf_coarse.py:1339-1407)
# ----------------------------------------------------------------------------------------------
def rays_at_pixels(px_i, px_j, K, c2w, inverse_y=True, mode='center', normalize=True):
    """Rays through pixel columns px_i / rows px_j (float tensors of equal shape) of one view.

    normalize=True is the Voxurf variant (rays_d = viewdirs = d/|d|, voxurf_coarse.py:1404);
    normalize=False the DVGO one (dvgo_ori.py:562-563: un-normalised rays_d, unit viewdirs).
    """
    i, j = px_i.float(), px_j.float()
    if mode == 'center':
        i, j = i + 0.5, j + 0.5
    elif mode != 'lefttop':
        raise NotImplementedError
    if inverse_y:
        dirs = torch.stack([(i - K[0][2]) / K[0][0], (j - K[1][2]) / K[1][1], torch.ones_like(i)], -1)
    else:
        dirs = torch.stack([(i - K[0][2]) / K[0][0], -(j - K[1][2]) / K[1][1], -torch.ones_like(i)], -1)
    rays_d = torch.sum(dirs[..., None, :] * c2w[:3, :3], -1)
    rays_o = c2w[:3, 3].expand(rays_d.shape)
    viewdirs = rays_d / rays_d.norm(dim=-1, keepdim=True)
    if normalize:
        rays_d = viewdirs
    return rays_o, rays_d, viewdirs


def rays_of_view(H, W, K, c2w, inverse_y=True, flip_x=False, flip_y=False, normalize=True):
    jj, ii = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32),
                            indexing='ij')
    if flip_x:
        ii = ii.flip((1,))
    if flip_y:
        jj = jj.flip((0,))
    return rays_at_pixels(ii, jj, K, c2w, inverse_y=inverse_y, normalize=normalize)


def select_training_rays(ray_idx, images, masks, Ks, c2w, inverse_y=True):
    """Rays + targets for flat indices into the [V,H,W] 'flatten' ordering
    (get_training_rays_flatten voxurf_coarse.py:1518-1549 followed by [indices], recon_scene.py:598-600)."""
    V, H, W = images.shape[:3]
    view = torch.div(ray_idx, H * W, rounding_mode='floor')
    rem = ray_idx - view * (H * W)
    pj = torch.div(rem, W, rounding_mode='floor')
    pi = rem - pj * W
    ro, rd, vd = [], [], []
    # evaluate per view so that the arithmetic is the per-view expression of the reference
    o_all = torch.zeros(len(ray_idx), 3)
    d_all = torch.zeros(len(ray_idx), 3)
    for v in range(V):
        sel = (view == v).nonzero()[:, 0]
        if len(sel) == 0:
            continue
        o, d, _ = rays_at_pixels(pi[sel], pj[sel], Ks[v], c2w[v], inverse_y=inverse_y)
        o_all = o_all.index_put((sel,), o)
        d_all = d_all.index_put((sel,), d)
    target = images.reshape(-1, 3)[ray_idx]
    mask = masks.reshape(-1, 1)[ray_idx]
    return o_all, d_all, d_all, target, mask


# ----------------------------------------------------------------------------------------------
# scene / parameters
# ----------------------------------------------------------------------------------------------
@dataclass
class Scene:
    xyz_min: torch.Tensor
    xyz_max: torch.Tensor
    num_voxels: int
    stepsize: float = 1.5
    near: float = 0.24
    far: float = 4.8
    bg: float = 0.
    N_iters: int = 10000
    s_ratio: float = 50.
    s_start: float = 0.2
    step_start: float = 0.
    barf_c2f: Optional[Sequence[float]] = (0.6, 1.)
    posbase_pe: int = 5
    viewbase_pe: int = 1
    k0_dim: int = 12
    output_range: float = 1.0  # DeformedImplicitField.output_range = range_shape.max()
    rect_size: Optional[Sequence[float]] = None
    voxel_size: torch.Tensor = field(init=False)
    world_size: torch.Tensor = field(init=False)

    def __post_init__(self):
        self.xyz_min = torch.as_tensor(self.xyz_min, dtype=torch.float32)
        self.xyz_max = torch.as_tensor(self.xyz_max, dtype=torch.float32)
        # voxurf_coarse.py:319-323
        self.voxel_size = ((self.xyz_max - self.xyz_min).prod() / self.num_voxels).pow(1 / 3)
        self.world_size = ((self.xyz_max - self.xyz_min) / self.voxel_size).long()

    def n_samples(self):
        # voxurf_coarse.py:700
        return int(np.linalg.norm(np.array(self.world_size.tolist()) + 1) / self.stepsize) + 1


def cube_sdf_init(scene: Scene):
    """Literal restatement of the 'cube_init' template (voxurf_coarse.py:153-170), incl. its quirk that it
    is only a box SDF for a centred box."""
    lo, hi, ws = scene.xyz_min, scene.xyz_max, scene.world_size
    x, y, z = np.mgrid[lo[0].item():hi[0].item():ws[0].item() * 1j,
                       lo[1].item():hi[1].item():ws[1].item() * 1j,
                       lo[2].item():hi[2].item():ws[2].item() * 1j]
    c = ((hi + lo) / 2).tolist()
    r = scene.rect_size
    d = []
    for ax, g in enumerate((x, y, z)):
        d.append(np.minimum(np.abs(g - (r[ax] / 2 - c[ax])), np.abs(g - (r[ax] / 2 + c[ax]))))
    sdf = torch.from_numpy((d[0] ** 2 + d[1] ** 2 + d[2] ** 2) ** 0.5)
    inside = np.ones_like(x, dtype=bool)
    for ax, g in enumerate((x, y, z)):
        inside &= (g >= (c[ax] - r[ax] / 2)) & (g <= (c[ax] + r[ax] / 2))
    sdf[torch.from_numpy(inside)] *= -1
    return sdf.float()[None, None]


def init_params(scene: Scene, seed=0, k0_std=0.1, warp_last_std=1e-2, rgbnet_width=128, rgbnet_depth=4,
                geo_rgb_dim=3):
    """Random-init parameter set with the reference's shapes (state_dict names of SURVEY §8b)."""
    g = torch.Generator().manual_seed(seed)
    ws = scene.world_size.tolist()
    P = {}
    P['sdf'] = cube_sdf_init(scene)
    P['k0'] = torch.randn(1, scene.k0_dim, *ws, generator=g) * k0_std
    P['sdf_alpha'] = torch.tensor([10.0])
    P['sdf_beta'] = torch.tensor([2.0])
    dim0 = (3 + 3 * scene.posbase_pe * 2) + (3 + 3 * scene.viewbase_pe * 2) + scene.k0_dim + geo_rgb_dim
    dims = [dim0] + [rgbnet_width] * (rgbnet_depth - 1) + [3]
    P['rgbnet'] = []
    for li in range(len(dims) - 1):
        bound = 1 / math.sqrt(dims[li])
        Wt = (torch.rand(dims[li + 1], dims[li], generator=g) * 2 - 1) * bound
        b = (torch.rand(dims[li + 1], generator=g) * 2 - 1) * bound
        if li == len(dims) - 2:
            b = torch.zeros_like(b)
        P['rgbnet'].append((Wt, b))
    wd = [3, 128, 128, 128, 128, 4]
    P['warp'] = []
    for li in range(5):
        std = math.sqrt(2.0 / wd[li])
        Wt = torch.randn(wd[li + 1], wd[li], generator=g) * std
        bound = 1 / math.sqrt(wd[li])
        b = (torch.rand(wd[li + 1], generator=g) * 2 - 1) * bound
        if li == 4:
            Wt = torch.randn(wd[li + 1], wd[li], generator=g) * warp_last_std
            b = torch.randn(wd[li + 1], generator=g) * warp_last_std
        P['warp'].append((Wt, b))
    return P


def params_require_grad(P, sdf=False):
    P['k0'].requires_grad_(True)
    P['sdf'].requires_grad_(sdf)
    P['sdf_alpha'].requires_grad_(True)
    P['sdf_beta'].requires_grad_(True)
    for Wt, b in P['rgbnet'] + P['warp']:
        Wt.requires_grad_(True)
        b.requires_grad_(True)
    return P


def flat_param_list(P):
    out = [('k0', P['k0']), ('sdf_alpha', P['sdf_alpha']), ('sdf_beta', P['sdf_beta'])]
    for li, (Wt, b) in enumerate(P['rgbnet']):
        out += [(f'rgbnet.{li}.weight', Wt), (f'rgbnet.{li}.bias', b)]
    for li, (Wt, b) in enumerate(P['warp']):
        out += [(f'warp.{li}.weight', Wt), (f'warp.{li}.bias', b)]
    return out


# ----------------------------------------------------------------------------------------------
# samplers
# ----------------------------------------------------------------------------------------------
def sample_dense(scene: Scene, rays_o, rays_d, jitter=None):
    """sample_ray_ori (voxurf_coarse.py:697-719) with the per-ray U[0,1) jitter passed in ([N] or None)."""
    S = scene.n_samples()
    vec = torch.where(rays_d == 0, torch.full_like(rays_d, 1e-6), rays_d)
    rate_a = (scene.xyz_max - rays_o) / vec
    rate_b = (scene.xyz_min - rays_o) / vec
    t_min = torch.minimum(rate_a, rate_b).amax(-1).clamp(min=scene.near, max=scene.far)
    t_max = torch.maximum(rate_a, rate_b).amin(-1).clamp(min=scene.near, max=scene.far)
    mask_out = (t_max <= t_min)
    rng = torch.arange(S)[None].float().repeat(rays_d.shape[-2], 1)
    if jitter is not None:
        rng = rng + jitter.reshape(-1, 1)
    step = scene.stepsize * scene.voxel_size * rng
    interpx = t_min[..., None] + step / rays_d.norm(dim=-1, keepdim=True)
    pts = rays_o[..., None, :] + rays_d[..., None, :] * interpx[..., None]
    mask_out = mask_out[..., None] | ((scene.xyz_min > pts) | (pts > scene.xyz_max)).any(dim=-1)
    return pts, mask_out, step, t_min, t_max


def compact_samples(pts, mask_out, step):
    """voxurf_coarse.py:936-945: ray-major boolean compaction."""
    N, S = step.shape
    ray_id = torch.arange(N * S) // S
    keep = ~mask_out.flatten()
    return pts.reshape(-1, 3)[keep], ray_id[keep], step.flatten()[keep], keep


def sample_variable(scene: Scene, rays_o, rays_d):
    """sample_ray_cuda (voxurf_coarse.py:661-695): far forced to 1e9, points recomputed in torch."""
    stepdist = scene.stepsize * scene.voxel_size
    (_, mask_out, ray_id, step_id, _, t_min, _, start, _) = native_ops.sample_pts_on_rays(
        rays_o, rays_d, scene.xyz_min, scene.xyz_max, scene.near, 1e9, float(stepdist))
    view = rays_d / rays_d.norm(dim=-1, keepdim=True)
    pts = start[ray_id] + view[ray_id] * step_id[..., None] * stepdist
    keep = ~mask_out
    return pts[keep], ray_id[keep], step_id[keep], mask_out, t_min


# ----------------------------------------------------------------------------------------------
# grid lookups
# ----------------------------------------------------------------------------------------------
def _norm_coords(scene, xyz):
    """xyz [M,3] world -> [1,1,1,M,3] in [-1,1] with the x<->z flip (voxurf_coarse.py:527-528)."""
    return ((xyz.reshape(1, 1, 1, -1, 3) - scene.xyz_min) / (scene.xyz_max - scene.xyz_min)).flip((-1,)) * 2 - 1


def trilinear_custom(grid, optical):
    """Twice-differentiable trilinear lookup with the reference's corner order, *unclamped* weights and
    clamped indices (grid_sample_3d, voxurf_coarse.py:545-659).  grid [1,C,D,H,W], optical [1,1,1,M,3]."""
    _, Cc, ID, IH, IW = grid.shape
    ix = ((optical[..., 0] + 1) / 2) * (IW - 1)
    iy = ((optical[..., 1] + 1) / 2) * (IH - 1)
    iz = ((optical[..., 2] + 1) / 2) * (ID - 1)
    with torch.no_grad():
        x0, y0, z0 = torch.floor(ix), torch.floor(iy), torch.floor(iz)
        x1, y1, z1 = x0 + 1, y0 + 1, z0 + 1
    flat = grid.view(1, Cc, ID * IH * IW)
    out = None
    # corner order tnw,tne,tsw,tse,bnw,bne,bsw,bse == (dz,dy,dx) lexicographic
    for dz in (0, 1):
        for dy in (0, 1):
            for dx in (0, 1):
                wx = (ix - x0) if dx else (x1 - ix)
                wy = (iy - y0) if dy else (y1 - iy)
                wz = (iz - z0) if dz else (z1 - iz)
                w = wx * wy * wz
                with torch.no_grad():
                    cx = torch.clamp(x1 if dx else x0, 0, IW - 1)
                    cy = torch.clamp(y1 if dy else y0, 0, IH - 1)
                    cz = torch.clamp(z1 if dz else z0, 0, ID - 1)
                    idx = (cz * IW * IH + cy * IW + cx).long().view(1, 1, -1).repeat(1, Cc, 1)
                val = torch.gather(flat, 2, idx)
                term = val.view(1, Cc, *ix.shape[1:]) * w.view(1, 1, *ix.shape[1:])
                out = term if out is None else out + term
    return out


def lookup_custom(scene, grid, xyz):
    """grid_sampler(..., use_custom=True) (voxurf_coarse.py:522-543) for C==1 -> [M]."""
    out = trilinear_custom(grid, _norm_coords(scene, xyz))
    return out.reshape(grid.shape[1], -1).T.reshape(*xyz.shape[:-1], grid.shape[1]).squeeze()


def lookup_dense(scene, grid, xyz, padding_mode='zeros'):
    """DenseGrid.forward (grid.py:47-58): F.grid_sample bilinear, align_corners, zeros padding."""
    C = grid.shape[1]
    out = F.grid_sample(grid, _norm_coords(scene, xyz), mode='bilinear', align_corners=True,
                        padding_mode=padding_mode)
    out = out.reshape(C, -1).T.reshape(*xyz.shape[:-1], C)
    return out.squeeze(-1) if C == 1 else out


def query_first_crossing(scene: Scene, sdf_d, t_min, rays_o, rays_d):
    """Tail shared by query_sdf_point_wocuda / _wodeform (voxurf_coarse.py:766-786, :811-831): first sign change of the
    dense SDF row, linear zero crossing, surface point -> pts [N,3], hit mask [N].  Differentiable like the reference."""
    prev_sdf, next_sdf = sdf_d[:, :-1], sdf_d[:, 1:]
    sign = prev_sdf * next_sdf
    sign = torch.where(sign <= 0, torch.ones_like(sign), torch.zeros_like(sign))
    prev_idx = torch.argmax(sign, 1, keepdim=True)
    next_idx = prev_idx + 1
    sdf1 = torch.gather(sdf_d, 1, prev_idx).squeeze(-1)
    sdf2 = torch.gather(sdf_d, 1, next_idx).squeeze(-1)
    dist = scene.stepsize * scene.voxel_size
    z1 = prev_idx.squeeze(-1) * dist + dist * 0.5
    z2 = next_idx.squeeze(-1) * dist + dist * 0.5
    z0 = (sdf1 * z2 - sdf2 * z1) / (sdf1 - sdf2 + 1e-10)
    z0 = torch.where(z0 < z1, torch.zeros_like(z0), z0)
    z0 = torch.where(z0 > z2, torch.zeros_like(z0), z0)
    hit = ((sdf1 * sdf2) < 0) * (z0 > 1e-10)
    interpx = t_min + z0 / rays_d.norm(dim=-1, keepdim=False)
    return rays_o + rays_d * interpx[..., None], hit


def query_wodeform(scene: Scene, sdf_grid, rays_o, rays_d, jitter=None):
    """query_sdf_point_wocuda_wodeform(keep_dim=True) (voxurf_coarse.py:797-833): RAW template looked up with border padding
    (grid_sampler's F.grid_sample path, :539-540) at EVERY dense slot -> pts [N,3], hit [N], sdf_d [N,S]."""
    pts, _, _, t_min, _ = sample_dense(scene, rays_o, rays_d, jitter)
    sdf_d = lookup_dense(scene, sdf_grid, pts, padding_mode='border').reshape(pts.shape[0], pts.shape[1])
    out, hit = query_first_crossing(scene, sdf_d, t_min, rays_o, rays_d)
    return out, hit, sdf_d


def mapped_sdf_grid(P):
    """sp10(alpha) * (sigmoid(sp10(beta) * sdf) - 0.5)   (voxurf_coarse.py:946-949)."""
    sp = lambda t: F.softplus(t, beta=10)
    return sp(P['sdf_alpha']) * (torch.sigmoid(sp(P['sdf_beta']) * P['sdf']) - 0.5)


# ----------------------------------------------------------------------------------------------
# MLPs
# ----------------------------------------------------------------------------------------------
def warp_mlp(P, scene, pts, chunk=8192 * 2):
    """DeformedImplicitField.forward (deform_net.py:23-29): ReLU MLP 3->128x4->4, x output_range."""
    outs = []
    for x in pts.split(chunk, 0):
        h = x
        for li, (Wt, b) in enumerate(P['warp']):
            h = F.linear(h, Wt, b)
            if li < len(P['warp']) - 1:
                h = F.relu(h)
        outs.append(h)
    out = torch.cat(outs, 0) * scene.output_range
    return out[:, :3], out[:, 3:]


def rgbnet_mlp(P, feat):
    h = feat
    for li, (Wt, b) in enumerate(P['rgbnet']):
        h = F.linear(h, Wt, b)
        if li < len(P['rgbnet']) - 1:
            h = F.relu(h)
    return h


def barf_weights(scene, progress, L):
    """positional_encoding_barf weights (voxurf_coarse.py:721-732). progress: python float."""
    if scene.barf_c2f is None:
        return None
    start, end = scene.barf_c2f
    alpha = (torch.tensor(progress, dtype=torch.float32) - start) / (end - start) * L
    k = torch.arange(L, dtype=torch.float32)
    return (1 - (alpha - k).clamp_(min=0, max=1).mul_(np.pi).cos_()) / 2


def posenc(scene, x, L, progress):
    """[x, w*sin(2^k x), w*cos(2^k x)] in the reference's memory order (voxurf_coarse.py:1011-1016)."""
    freq = torch.tensor([2. ** i for i in range(L)])
    emb = (x.unsqueeze(-1) * freq).flatten(-2)
    enc = torch.cat([emb.sin(), emb.cos()], -1)
    w = barf_weights(scene, progress, L)
    if w is not None:
        shape = enc.shape
        enc = (enc.view(-1, L) * w).view(*shape)
    return torch.cat([x, enc], dim=-1)


# ----------------------------------------------------------------------------------------------
# NeuS alpha, transmittance
# ----------------------------------------------------------------------------------------------
def s_val_at(scene, global_step):
    return 1. / (global_step + scene.s_ratio / scene.s_start - scene.step_start) * scene.s_ratio


def neus_alpha(scene, viewdirs, ray_id, dist, sdf, gradients, s_val_tensor):
    """neus_alpha_from_sdf_scatter with use_mid=True, cos_anneal_ratio=1 (voxurf_coarse.py:483-519)."""
    dirs = viewdirs[ray_id]
    inv_s = torch.ones(1) / s_val_tensor
    true_cos = (dirs * gradients).sum(-1, keepdim=True)
    iter_cos = -(F.relu(-true_cos * 0.5 + 0.5) * 0.0 + F.relu(-true_cos) * 1.0)
    sdf = sdf.unsqueeze(-1)
    nxt = sdf + iter_cos * dist.reshape(-1, 1) * 0.5
    prv = sdf - iter_cos * dist.reshape(-1, 1) * 0.5
    prev_cdf = torch.sigmoid(prv * inv_s.reshape(-1, 1))
    next_cdf = torch.sigmoid(nxt * inv_s.reshape(-1, 1))
    p = prev_cdf - next_cdf
    return ((p + 1e-5) / (prev_cdf + 1e-5)).clip(0.0, 1.0).squeeze(-1)


class Alphas2Weights(torch.autograd.Function):
    """voxurf_coarse.py:1316-1332 over the C restatement of the CUDA kernels."""

    @staticmethod
    def forward(ctx, alpha, ray_id, N):
        w, T, last, i_s, i_e = native_ops.alpha2weight(alpha, ray_id, N)
        ctx.save_for_backward(alpha, w, T, last, i_s, i_e)
        ctx.n_rays = N
        return w, last

    @staticmethod
    def backward(ctx, gw, gl):
        alpha, w, T, last, i_s, i_e = ctx.saved_tensors
        g = native_ops.alpha2weight_backward(alpha, w, T, last, i_s, i_e, ctx.n_rays, gw, gl)
        return g, None, None


def segment_sum(src, index, N):
    """torch_scatter.segment_coo(reduce='sum') restated from its documentation (parity unpinned)."""
    out = torch.zeros([N, *src.shape[1:]])
    return out.index_add(0, index, src)


def total_variation(v):
    """voxurf_coarse.py:1298-1313 without mask."""
    tv2 = (v[:, :, 1:, :, :] - v[:, :, :-1, :, :]).abs()
    tv3 = (v[:, :, :, 1:, :] - v[:, :, :, :-1, :]).abs()
    tv4 = (v[:, :, :, :, 1:] - v[:, :, :, :, :-1]).abs()
    return (tv2.sum() + tv3.sum() + tv4.sum()) / 3 / torch.ones_like(v).sum()


# ----------------------------------------------------------------------------------------------
# Voxurf.forward (train) and .inference
# ----------------------------------------------------------------------------------------------
def _geometry(P, scene, ray_pts):
    """warp -> custom lookup -> Jacobian / normal (voxurf_coarse.py:946-984). ray_pts must require grad."""
    sdf_grid = mapped_sdf_grid(P)
    deform, correction = warp_mlp(P, scene, ray_pts)
    new_coords = deform + ray_pts
    sdf = lookup_custom(scene, sdf_grid, new_coords)
    ones = torch.ones_like(new_coords[:, 0])
    cols = [torch.autograd.grad(new_coords[:, c], [ray_pts], grad_outputs=ones, create_graph=True)[0]
            for c in range(3)]
    grad_deform = torch.stack(cols, dim=2)
    sdf_final = sdf + correction.squeeze(-1)
    sdf_deform = sdf_final - lookup_custom(scene, sdf_grid, ray_pts)
    if sdf_final.shape[0] == 0:
        gradient = torch.zeros((0, 3))
    else:
        gradient = torch.autograd.grad(sdf_final, [ray_pts], grad_outputs=torch.ones_like(sdf_final),
                                       create_graph=True)[0]
    return sdf_final, sdf_deform, grad_deform, correction, gradient


def _color(P, scene, ray_pts, ray_id, viewdirs, gradient, progress):
    """k0 lookup, PE, normal, rgbnet, sigmoid (voxurf_coarse.py:1005-1033)."""
    k0 = lookup_dense(scene, P['k0'], ray_pts)
    rays_xyz = (ray_pts - scene.xyz_min) / (scene.xyz_max - scene.xyz_min)
    xyz_emb = posenc(scene, rays_xyz, scene.posbase_pe, progress)
    view_emb = posenc(scene, viewdirs, scene.viewbase_pe, progress)
    feat = torch.cat([k0, xyz_emb, view_emb.flatten(0, -2)[ray_id]], -1)
    normal = gradient / (gradient.norm(dim=-1, keepdim=True) + 1e-5)
    feat = torch.cat([feat, normal], -1)
    return torch.sigmoid(rgbnet_mlp(P, feat)), feat


def voxurf_forward(P, scene: Scene, rays_o, rays_d, viewdirs, jitter=None, global_step=None, render_grad=False):
    """Voxurf.forward with use_deform=True (voxurf_coarse.py:922-1092). Rays must carry grad (pose)."""
    is_train = global_step is not None
    progress = (global_step / scene.N_iters) if is_train else 1.
    N = len(rays_o)
    pts, mask_out, step, t_min, t_max = sample_dense(scene, rays_o, rays_d, jitter if is_train else None)
    ray_pts, ray_id, step_c, keep = compact_samples(pts, mask_out, step)
    if not ray_pts.requires_grad:
        ray_pts = ray_pts.requires_grad_(True)
    sdf_final, sdf_deform, grad_deform, correction, gradient = _geometry(P, scene, ray_pts)
    dist = scene.stepsize * scene.voxel_size
    if is_train:
        s_val = s_val_at(scene, global_step)
        s_t = torch.ones(1) * s_val
    else:
        s_val = 0
        s_t = torch.ones(1) * scene.s_start   # value of self.s_val left from construction
    alpha = neus_alpha(scene, viewdirs, ray_id, dist, sdf_final, gradient, s_t)
    weights, alphainv_last = Alphas2Weights.apply(alpha, ray_id, N)
    rgb, feat = _color(P, scene, ray_pts, ray_id, viewdirs, gradient, progress)
    rgb_marched = segment_sum(weights.unsqueeze(-1) * rgb, ray_id, N)
    cum_weights = segment_sum(weights.unsqueeze(-1), ray_id, N)
    rgb_marched = (rgb_marched + (1 - cum_weights) * scene.bg).clamp(0, 1)
    normal_marched = None
    if render_grad:
        nrm = gradient / (gradient.norm(2, -1, keepdim=True) + 1e-6)
        normal_marched = segment_sum(weights.unsqueeze(-1) * nrm, ray_id, N)
    n_step = segment_sum(weights * step_c, ray_id, N)
    depth = t_min / rays_d.norm(dim=-1, keepdim=False) + n_step
    k0_tv = total_variation(P['k0'])
    return {
        'alphainv_cum': alphainv_last, 'weights': weights, 'cum_weights': cum_weights,
        'rgb_marched': rgb_marched, 'normal_marched': normal_marched, 'raw_alpha': alpha, 'raw_rgb': rgb,
        'depth': depth, 'disp': 1 / depth, 'mask': keep, 'mask_outbbox': mask_out.flatten()[keep],
        'gradient': gradient, 's_val': s_val, 'k0_tv': k0_tv, 'sdf_deform': sdf_deform,
        'grad_deform': grad_deform, 'sdf_correct': correction,
        # extras for kernel-level parity (not part of the reference dict)
        '_ray_pts': ray_pts, '_ray_id': ray_id, '_step': step_c, '_t_min': t_min, '_t_max': t_max,
        '_sdf_final': sdf_final, '_rgb_feat': feat,
    }


def voxurf_inference(P, scene: Scene, rays_o, rays_d, viewdirs, global_step=None):
    """Voxurf.inference (voxurf_coarse.py:1094-1222)."""
    is_train = global_step is not None
    progress = (global_step / scene.N_iters) if is_train else 1.
    N = len(rays_o)
    ray_pts, ray_id, step_id, mask_out, t_min = sample_variable(scene, rays_o, rays_d)
    if ray_pts.shape[0] == 1:
        ray_pts, ray_id, step_id = ray_pts.repeat(2, 1), ray_id.repeat(2), step_id.repeat(2)
    with torch.enable_grad():
        ray_pts = ray_pts.detach().requires_grad_(True)
        sdf_grid = mapped_sdf_grid(P)
        deform, correction = warp_mlp(P, scene, ray_pts)
        new_coords = deform + ray_pts
        sdf_final = lookup_custom(scene, sdf_grid, new_coords) + correction.squeeze(-1)
        if sdf_final.shape[0] == 0:
            gradient = torch.zeros((0, 3))
        else:
            gradient = torch.autograd.grad(sdf_final, [ray_pts], grad_outputs=torch.ones_like(sdf_final),
                                           create_graph=True)[0]
    gradient_error = ((torch.linalg.norm(gradient, ord=2, dim=-1) - 1.0) ** 2).mean()
    dist = scene.stepsize * scene.voxel_size
    if is_train:
        s_val = s_val_at(scene, global_step)
        s_t = torch.ones(1) * s_val
    else:
        s_val = 0
        s_t = torch.ones(1) * scene.s_start
    alpha = neus_alpha(scene, viewdirs, ray_id, dist, sdf_final, gradient, s_t)
    weights, alphainv_last = Alphas2Weights.apply(alpha, ray_id, N)
    rgb, _ = _color(P, scene, ray_pts, ray_id, viewdirs, gradient, progress)
    rgb_marched = segment_sum(weights.unsqueeze(-1) * rgb, ray_id, N)
    cum_weights = segment_sum(weights.unsqueeze(-1), ray_id, N)
    rgb_marched = (rgb_marched + (1 - cum_weights) * scene.bg).clamp(0, 1)
    nrm = gradient / (gradient.norm(2, -1, keepdim=True) + 1e-6)
    normal_marched = segment_sum(weights.unsqueeze(-1) * nrm, ray_id, N)
    depth = segment_sum(weights * step_id * dist, ray_id, N)
    return {
        'alphainv_cum': alphainv_last, 'weights': weights, 'cum_weights': cum_weights,
        'rgb_marched': rgb_marched, 'normal_marched': normal_marched, 'raw_alpha': alpha, 'raw_rgb': rgb,
        'depth': depth, 'disp': 1 / depth, 'mask': torch.ones_like(step_id), 'mask_outbbox': mask_out,
        'gradient': gradient, 'gradient_error': gradient_error, 's_val': s_val,
        '_ray_pts': ray_pts, '_ray_id': ray_id, '_step_id': step_id,
    }


# ----------------------------------------------------------------------------------------------
# losses (lib/losses.py)
# ----------------------------------------------------------------------------------------------
def dynamic_weight(w0, w1, it, total):
    return w0 * math.exp(math.log(w1 / w0) / total * it)


def object_losses(out, target, mask, iteration, total_iterations, weight_main=1.0, weight_tv_k0=0.01,
                  weight_mask=0.1, use_deform=True):
    """losses.py:34-74 -> (dict of scalars, dict of weights, total loss)."""
    S, Wt = {}, {}
    S['img_render'] = F.mse_loss(out['rgb_marched'] * mask, target * mask, reduction='sum') / (mask.sum() * 3)
    Wt['img_render'] = weight_main
    pout = out['alphainv_cum'].clamp(1e-6, 1 - 1e-6)
    S['weight_entropy_last'] = -(pout * torch.log(pout) + (1 - pout) * torch.log(1 - pout)).mean()
    Wt['weight_entropy_last'] = 0.01
    if weight_tv_k0 > 0:
        S['tv_k0'] = out['k0_tv']
        Wt['tv_k0'] = weight_tv_k0
    S['grad_constraint'] = torch.abs(out['gradient'].norm(dim=-1) - 1).mean()
    Wt['grad_constraint'] = 1.0
    if use_deform:
        w = dynamic_weight(1e-1, 1e-3, iteration, total_iterations)
        S['grad_deform_constraint'] = out['grad_deform'].norm(dim=-1).mean()
        S['sdf_correct_constraint'] = torch.abs(out['sdf_correct']).mean()
        S['sdf_deform_constraint'] = torch.abs(out['sdf_deform']).mean()
        Wt['grad_deform_constraint'] = Wt['sdf_correct_constraint'] = Wt['sdf_deform_constraint'] = w
    S['mask_render'] = F.binary_cross_entropy(out['cum_weights'].clip(1e-3, 1.0 - 1e-3), mask)
    Wt['mask_render'] = weight_mask
    loss = 0
    for k, v in S.items():
        loss = loss + v * Wt[k]
    return S, Wt, loss


# ----------------------------------------------------------------------------------------------
# Adam (lib/utils.py:82-198) - functional, one tensor
# ----------------------------------------------------------------------------------------------
def adam_update(p, g, m, v, step, lr, beta1=0.9, beta2=0.99, eps=1e-8):
    """In-place on p, m, v (no weight decay, no amsgrad, no per-voxel lr: the live configuration)."""
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))


class TrainState:
    """Minimal counterpart of recon_scene.optimize_increamental's object-branch step (recon_scene.py:572-606,
    :648-649, :742-747, :765-771) for trajectory tests and the CPU baseline: explicit ray indices + jitter."""

    LR = {'k0': 1e-1, 'rgbnet': 1e-3, 'warp': 1e-3, 'sdf_alpha': 1e-2, 'sdf_beta': 1e-2}

    def __init__(self, P, scene, pose_init, Ks, images, masks, se3_refine=None, lr_pose=1e-3, lr_pose_end=1e-4,
                 pose_iters=1, lrate_decay=10, loss_scale=0.1, fix_first=True):
        self.P, self.scene = params_require_grad(P), scene
        self.pose_init, self.Ks, self.images, self.masks = pose_init, Ks, images, masks
        V = pose_init.shape[0]
        self.se3 = (torch.zeros(V, 6) if se3_refine is None else se3_refine.clone()).requires_grad_(True)
        self.groups = []
        for name, t in flat_param_list(P):
            key = name.split('.')[0]
            self.groups.append(dict(name=name, p=t, lr=self.LR[key], m=torch.zeros_like(t), v=torch.zeros_like(t)))
        self.pose_m, self.pose_v = torch.zeros(V, 6), torch.zeros(V, 6)
        self.lr_pose = lr_pose
        self.pose_gamma = (lr_pose_end / (1e-10 + lr_pose)) ** (1. / pose_iters)
        self.decay = 0.1 ** (1 / (lrate_decay * 1000))
        self.loss_scale, self.fix_first = loss_scale, fix_first
        self.n_step = 0

    def loss_and_grads(self, ray_idx, jitter, global_step):
        for g in self.groups:
            g['p'].grad = None
        self.se3.grad = None
        w2c = current_pose_pnp(self.se3, self.pose_init, self.fix_first)
        c2w = pose_invert(w2c)
        ro, rd, vd, target, mask = select_training_rays(ray_idx, self.images, self.masks, self.Ks, c2w)
        out = voxurf_forward(self.P, self.scene, ro, rd, vd, jitter=jitter, global_step=global_step)
        S, Wt, loss = object_losses(out, target, mask, global_step, self.scene.N_iters)
        (loss * self.loss_scale).backward()
        return out, S, loss

    def step(self, ray_idx, jitter, global_step, optimize_pose=True):
        out, S, loss = self.loss_and_grads(ray_idx, jitter, global_step)
        self.n_step += 1
        with torch.no_grad():
            for g in self.groups:
                g['lr'] = g['lr'] * self.decay          # recon_scene.py:742-746 (decay precedes step :768)
            for g in self.groups:
                if g['p'].grad is not None:
                    adam_update(g['p'], g['p'].grad, g['m'], g['v'], self.n_step, g['lr'])
            if optimize_pose and self.se3.grad is not None:
                adam_update(self.se3, self.se3.grad, self.pose_m, self.pose_v, self.n_step, self.lr_pose,
                            beta2=0.999)
                self.lr_pose *= self.pose_gamma
        return out, S, loss


# ----------------------------------------------------------------------------------------------
# DirectVoxGO twin (lib/dvgo_ori.py)
# ----------------------------------------------------------------------------------------------
def cumprod_exclusive(p):
    return torch.cat([torch.ones_like(p[..., [0]]), p.clamp_min(1e-10).cumprod(-1)], -1)


def dvgo_forward(density, k0, rgbnet, scene: Scene, rays_o, rays_d, viewdirs, alpha_init=1e-2, jitter=None,
                 global_step=None, fast_color_thres=0., rgbnet_direct=True, voxel_size_ratio=1.0):
    """DirectVoxGO.forward, post-activation branch, no mask cache (dvgo_ori.py:289-379)."""
    act_shift = np.log(1 / (1 - alpha_init) - 1)
    pts, mask_out, _, _, _ = sample_dense(scene, rays_o, rays_d, jitter if global_step is not None else None)
    interval = scene.stepsize * voxel_size_ratio
    alpha = torch.zeros_like(pts[..., 0])
    dens = lookup_dense(scene, density, pts[~mask_out])
    alpha[~mask_out] = 1 - torch.exp(-F.softplus(dens + act_shift) * interval)
    alphainv_cum = cumprod_exclusive(1 - alpha)
    weights = alpha * alphainv_cum[..., :-1]
    mask = weights > fast_color_thres
    kk = torch.zeros(*weights.shape, k0.shape[1])
    kk[mask] = lookup_dense(scene, k0, pts[mask])
    if rgbnet is None:
        rgb = torch.sigmoid(kk)
    else:
        k0_view = kk if rgbnet_direct else kk[..., 3:]
        vfreq = torch.tensor([2. ** i for i in range(scene.viewbase_pe)])
        pfreq = torch.tensor([2. ** i for i in range(scene.posbase_pe)])
        vemb = (viewdirs.unsqueeze(-1) * vfreq).flatten(-2)
        vemb = torch.cat([viewdirs, vemb.sin(), vemb.cos()], -1)
        rxyz = (pts[mask] - scene.xyz_min) / (scene.xyz_max - scene.xyz_min)
        xemb = (rxyz.unsqueeze(-1) * pfreq).flatten(-2)
        xemb = torch.cat([rxyz, xemb.sin(), xemb.cos()], -1)
        feat = torch.cat([k0_view[mask], xemb,
                          vemb.flatten(0, -2).unsqueeze(-2).repeat(1, weights.shape[-1], 1)[mask.flatten(0, -2)]], -1)
        logit = torch.zeros(*weights.shape, 3)
        h = feat
        for li, (Wt, b) in enumerate(rgbnet):
            h = F.linear(h, Wt, b)
            if li < len(rgbnet) - 1:
                h = F.relu(h)
        logit[mask] = h
        if not rgbnet_direct:
            logit[mask] = logit[mask] + kk[..., :3][mask]
        rgb = torch.sigmoid(logit)
    rgb_marched = ((weights[..., None] * rgb).sum(-2) + alphainv_cum[..., [-1]] * scene.bg).clamp(0, 1)
    depth = (rays_o[..., None, :] - pts).norm(dim=-1)
    depth = (weights * depth).sum(-1) + alphainv_cum[..., -1] * scene.far
    return {'alphainv_cum': alphainv_cum, 'weights': weights, 'rgb_marched': rgb_marched, 'raw_alpha': alpha,
            'raw_rgb': rgb, 'depth': depth, 'disp': 1 / depth, 'mask': mask, 'mask_outbbox': mask_out}
