"""Drop-in for the reference's `lib/dvgo_ori.py` (DirectVoxGO, the API twin of Voxurf): same constructor kwargs,
forward signature, return-dict keys and `state_dict` names, rendered by the HIP kernels through one autograd node.

Implemented branch: post-activated density (the default, lib/dvgo_ori.py:317-318), no mask cache, rgbnet present
(`rgbnet_dim > 0`, direct or diffuse+view split).  Gradients flow to `density`, `k0` and `rgbnet.*` through
`rgb_marched`, `alphainv_cum[..., -1]` and `depth` (what the DVGO losses consume); the per-sample fields are returned
detached.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops, render_utils
from .engine import SceneConfig
from .grid import channels_last_view
from .voxurf_coarse import get_rays, ndc_rays  # noqa: F401  (API surface)


def cumprod_exclusive(p):
    """lib/dvgo_ori.py:478-480"""
    return torch.cat([torch.ones_like(p[..., [0]]), p.clamp_min(1e-10).cumprod(-1)], -1)


def get_ray_marching_ray(alpha):
    """lib/dvgo_ori.py:482-485 for [N,S] alpha, via pp_march_dvgo_fwd (rows = rays)."""
    N, S = alpha.shape
    a = alpha.contiguous().float().reshape(-1)
    rs = (torch.arange(N + 1, device=a.device) * S).int()
    w, T, last = torch.empty_like(a), torch.empty_like(a), torch.empty(N, device=a.device)
    i_end = torch.empty(N, dtype=torch.int32, device=a.device)
    cw = torch.empty(N, device=a.device)
    ops.march_dvgo_fwd(a, None, None, rs, N, w, T, last, i_end, None, cw, None)
    return w.reshape(N, S), torch.cat([T.reshape(N, S), last[:, None]], -1)


def total_variation(v, mask=None):
    """lib/dvgo_ori.py:487-496: mean (not sum/numel) of |diff| per axis."""
    if mask is not None:
        raise NotImplementedError
    from .voxurf_coarse import total_variation as tv_sum
    C, (X, Y, Z) = v.shape[1], v.shape[2:]
    # sum|d_x|/n_x + sum|d_y|/n_y + sum|d_z|/n_z differs from the Voxurf normalisation; evaluate per axis on device
    t2 = v.diff(dim=2).abs().mean()
    t3 = v.diff(dim=3).abs().mean()
    t4 = v.diff(dim=4).abs().mean()
    return (t2 + t3 + t4) / 3


def get_rays_of_a_view(H, W, K, c2w, ndc, inverse_y, flip_x, flip_y, mode='center'):
    """lib/dvgo_ori.py:560-566: un-normalised rays_d, unit viewdirs."""
    rays_o, rays_d, viewdirs = get_rays(H, W, K, c2w, inverse_y, flip_x, flip_y, mode, normalize=False)
    if ndc:
        rays_o, rays_d = ndc_rays(H, W, K[0][0], 1., rays_o, rays_d)
    return rays_o, rays_d, viewdirs


def batch_indices_generator(N, BS):
    """lib/dvgo_ori.py:670-677"""
    idx, top = torch.LongTensor(np.random.permutation(N)), 0
    while True:
        if top + BS > N:
            idx, top = torch.LongTensor(np.random.permutation(N)), 0
        yield idx[top:top + BS]
        top += BS


class _DVGORender(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, st, density, k0, *mlp):
        cfg, sc = st['cfg'], st['cfg'].pp
        dev = density.device
        M, N, cap = st['M'], st['N'], st['cap']
        f = dict(device=dev, dtype=torch.float32)
        count, pts, ray_id, ray_start = st['count'], st['pts'], st['ray_id'], st['ray_start']
        dens = torch.zeros(cap, 1, **f)
        ops.grid_sample_fwd(sc, density[0, 0].contiguous(), 1, pts[:cap], 0, dens) if M > 0 else None
        exp_d, alpha = render_utils.raw2alpha(dens.reshape(-1), model.act_shift, st['interval'])
        w, T, last = torch.zeros(cap, **f), torch.ones(cap, **f), torch.empty(N, **f)
        i_end, cw = torch.empty(N, dtype=torch.int32, device=dev), torch.empty(N, **f)
        ops.march_dvgo_fwd(alpha, None, None, ray_start, N, w, T, last, i_end, None, cw, None)
        sel = (w > model.fast_color_thres).to(torch.uint8)
        n_gemm = model.rgbnet_kwargs['rgbnet_depth'] - 1
        ld = st['ld']
        skip = 0 if model.rgbnet_direct else 3
        feat = torch.empty(cap, ld, **f)
        k0_raw = None if model.rgbnet_direct else torch.empty(cap, model.k0_dim, **f)
        ops.feat_generic_fwd(sc, channels_last_view(k0), pts, st['viewdirs'], ray_id, None, None, sel, skip, ld, count, cap,
                             feat, k0_raw)
        params = torch.cat([p.detach().reshape(-1) for p in st['packed'](mlp)])
        acts = torch.empty(n_gemm, cap, 128, **f)
        rgb = torch.empty(cap, 3, **f)
        ops.mlp_fwd(params, feat, ld, n_gemm, count, cap, k0_raw, model.k0_dim, acts, rgb)
        rgb_eff = torch.where(sel.bool()[:, None], rgb, torch.full_like(rgb, 0.5))   # rgb_logit stays 0 where unmasked
        rgb_acc, depth_acc = torch.empty(N, 3, **f), torch.empty(N, **f)
        ops.march_dvgo_fwd(alpha, rgb_eff, st['dist_o'], ray_start, N, w, T, last, i_end, rgb_acc, cw, depth_acc)
        pre = rgb_acc + last[:, None] * cfg.bg
        depth = depth_acc + last * cfg.far
        ctx.model, ctx.st = model, st
        ctx.save_for_backward(density, k0, dens, exp_d, alpha, w, T, last, i_end, sel, feat, params, acts, rgb, rgb_eff,
                              pre, *([k0_raw] if k0_raw is not None else []))
        ctx.mark_non_differentiable(sel)
        return pre.clamp(0, 1), last.clone(), depth, w.clone(), alpha.clone(), rgb_eff.clone(), sel

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_rgbm, g_last, g_depth, g_w, g_alpha_up, g_rgb_up, _g_sel):
        model, st = ctx.model, ctx.st
        cfg, sc = st['cfg'], st['cfg'].pp
        (density, k0, dens, exp_d, alpha, w, T, last, i_end, sel, feat, params, acts, rgb, rgb_eff, pre, *rest) = ctx.saved_tensors
        k0_raw = rest[0] if rest else None
        N, cap, ld = st['N'], st['cap'], st['ld']
        dev = density.device
        f = dict(device=dev, dtype=torch.float32)
        g_acc = (g_rgbm * ((pre >= 0) & (pre <= 1))).contiguous()
        g_last_t = (g_last + cfg.bg * g_acc.sum(-1) + cfg.far * g_depth).contiguous()
        g_alpha, g_rgb = torch.empty(cap, **f), torch.empty(cap, 3, **f)
        # upstream gradients of the per-sample outputs (weights, raw_alpha, raw_rgb) enter next to the compositing terms
        ops.march_bwd(alpha, rgb_eff, st['dist_o'], w, T, last, st['ray_start'], i_end, N, 0.0, None, g_acc, None, g_last_t,
                      g_depth.contiguous(), g_w.contiguous().float(), g_alpha, g_rgb)
        g_alpha = g_alpha + g_alpha_up
        g_rgb = (g_rgb + g_rgb_up) * sel.bool()[:, None]
        n_gemm = model.rgbnet_kwargs['rgbnet_depth'] - 1
        scratch = torch.empty(3 * cap * 128 + 16384, **f)
        pgrad = torch.zeros_like(params)
        g_feat = torch.empty(cap, ld, **f)
        g_k0raw = None if k0_raw is None else torch.zeros(cap, model.k0_dim, **f)
        ops.mlp_bwd(params, feat, ld, n_gemm, acts, rgb, g_rgb.contiguous(), st['count'], cap, scratch, pgrad, g_feat,
                    g_k0raw, model.k0_dim)
        if g_k0raw is not None:
            g_k0raw = g_k0raw * sel.bool()[:, None]
        k0_grad = torch.zeros_like(k0, memory_format=torch.channels_last_3d)
        ops.feat_generic_bwd_k0(sc, st['pts'], sel, 0 if model.rgbnet_direct else 3, ld, st['count'], cap, g_feat, g_k0raw,
                                channels_last_view(k0_grad))
        g_dens = render_utils.raw2alpha_backward(exp_d, g_alpha, st['interval'])
        g_dens[st['M']:] = 0
        dgrad = torch.zeros_like(density)
        if st['M'] > 0:
            ops.grid_sample_bwd(sc, density[0, 0].contiguous(), 1, st['pts'][:cap], 0, g_dens.reshape(-1, 1).contiguous(),
                                dgrad[0, 0], None)
        return (None, None, dgrad, k0_grad, *st['unpack'](pgrad))


class DirectVoxGO(torch.nn.Module):
    """lib/dvgo_ori.py:14-435"""

    def __init__(self, xyz_min, xyz_max, num_voxels=0, num_voxels_base=0, alpha_init=None, nearest=False,
                 pre_act_density=False, in_act_density=False, mask_cache_path=None, mask_cache_thres=1e-3,
                 fast_color_thres=0, rgbnet_dim=0, rgbnet_direct=False, rgbnet_full_implicit=False, rgbnet_depth=3,
                 rgbnet_width=128, posbase_pe=5, viewbase_pe=4, **kwargs):
        super().__init__()
        if (rgbnet_dim <= 0 or rgbnet_dim % 4 or rgbnet_dim > 16 or rgbnet_full_implicit or rgbnet_width != 128 or nearest
                or pre_act_density or in_act_density or mask_cache_path):
            raise NotImplementedError('HIP DirectVoxGO: rgbnet_dim in {4,8,12,16}, width 128, post-activation, no mask cache')
        self.register_buffer('xyz_min', torch.Tensor(xyz_min))
        self.register_buffer('xyz_max', torch.Tensor(xyz_max))
        self.fast_color_thres, self.nearest = fast_color_thres, nearest
        self.pre_act_density, self.in_act_density = pre_act_density, in_act_density
        self.num_voxels_base = num_voxels_base
        self.voxel_size_base = ((self.xyz_max - self.xyz_min).prod() / self.num_voxels_base).pow(1 / 3)
        self.alpha_init = alpha_init
        self.act_shift = np.log(1 / (1 - alpha_init) - 1)
        self._set_grid_resolution(num_voxels)
        ws = [int(v) for v in self.world_size.tolist()]
        self.density = torch.nn.Parameter(torch.zeros([1, 1, *ws]))
        self.rgbnet_kwargs = {'rgbnet_dim': rgbnet_dim, 'rgbnet_direct': rgbnet_direct,
                              'rgbnet_full_implicit': rgbnet_full_implicit, 'rgbnet_depth': rgbnet_depth,
                              'rgbnet_width': rgbnet_width, 'posbase_pe': posbase_pe, 'viewbase_pe': viewbase_pe}
        self.rgbnet_full_implicit, self.rgbnet_direct = rgbnet_full_implicit, rgbnet_direct
        self.k0_dim = rgbnet_dim
        self.k0 = torch.nn.Parameter(torch.zeros([1, self.k0_dim, *ws]).contiguous(memory_format=torch.channels_last_3d))
        self.register_buffer('posfreq', torch.FloatTensor([(2 ** i) for i in range(posbase_pe)]))
        self.register_buffer('viewfreq', torch.FloatTensor([(2 ** i) for i in range(viewbase_pe)]))
        dim0 = (3 + 3 * posbase_pe * 2) + (3 + 3 * viewbase_pe * 2) + (self.k0_dim if rgbnet_direct else self.k0_dim - 3)
        self.dim0 = dim0
        self.rgbnet = nn.Sequential(
            nn.Linear(dim0, rgbnet_width), nn.ReLU(inplace=True),
            *[nn.Sequential(nn.Linear(rgbnet_width, rgbnet_width), nn.ReLU(inplace=True)) for _ in range(rgbnet_depth - 2)],
            nn.Linear(rgbnet_width, 3))
        nn.init.constant_(self.rgbnet[-1].bias, 0)
        self.mask_cache_path, self.mask_cache_thres = mask_cache_path, mask_cache_thres
        self.mask_cache = self.nonempty_mask = None
        self._cfg = None

    def _set_grid_resolution(self, num_voxels):
        self.num_voxels = num_voxels
        self.voxel_size = ((self.xyz_max - self.xyz_min).prod() / num_voxels).pow(1 / 3)
        self.world_size = ((self.xyz_max - self.xyz_min) / self.voxel_size).long()
        self.voxel_size_ratio = self.voxel_size / self.voxel_size_base

    def get_kwargs(self):
        return {'xyz_min': self.xyz_min.cpu().numpy(), 'xyz_max': self.xyz_max.cpu().numpy(),
                'num_voxels': self.num_voxels, 'num_voxels_base': self.num_voxels_base, 'alpha_init': self.alpha_init,
                'nearest': self.nearest, 'pre_act_density': self.pre_act_density, 'in_act_density': self.in_act_density,
                'mask_cache_path': self.mask_cache_path, 'mask_cache_thres': self.mask_cache_thres,
                'fast_color_thres': self.fast_color_thres, **self.rgbnet_kwargs}

    def get_MaskCache_kwargs(self):
        return {'xyz_min': self.xyz_min.cpu().numpy(), 'xyz_max': self.xyz_max.cpu().numpy(), 'act_shift': self.act_shift,
                'voxel_size_ratio': self.voxel_size_ratio, 'nearest': self.nearest,
                'pre_act_density': self.pre_act_density, 'in_act_density': self.in_act_density}

    def activate_density(self, density, interval=None):
        interval = interval if interval is not None else self.voxel_size_ratio
        shape = density.shape
        _, a = render_utils.raw2alpha(density.reshape(-1), self.act_shift, float(interval))
        return a.reshape(shape)

    def k0_total_variation(self):
        return total_variation(self.k0, self.nonempty_mask)

    def density_total_variation(self):
        return total_variation(self.activate_density(self.density, 1), self.nonempty_mask)

    def _linears(self):
        return [m for m in self.rgbnet.modules() if isinstance(m, nn.Linear)]

    def _scene(self, rk):
        key = (float(rk['stepsize']), float(rk['near']), float(rk['far']), float(rk['bg']))
        if self._cfg is None or self._cfg_key != key:
            self._cfg = SceneConfig(self.xyz_min.cpu().numpy(), self.xyz_max.cpu().numpy(), int(self.num_voxels),
                                    stepsize=key[0], near=key[1], far=key[2], bg=key[3], barf_c2f=None,
                                    posbase_pe=self.rgbnet_kwargs['posbase_pe'], viewbase_pe=self.rgbnet_kwargs['viewbase_pe'],
                                    k0_dim=self.k0_dim)
            self._cfg_key = key
        return self._cfg

    def forward(self, rays_o, rays_d, viewdirs, global_step=None, **render_kwargs):
        """lib/dvgo_ori.py:289-379.  Extension: render_kwargs['jitter'] overrides the internally drawn per-ray jitter."""
        if not rays_o.is_cuda:
            raise RuntimeError('poseprobe_amd.DirectVoxGO runs on the HIP path only: inputs must be CUDA tensors')
        cfg = self._scene(render_kwargs)
        dev = rays_o.device
        ro, rd, vd = (t.detach().contiguous().float() for t in (rays_o, rays_d, viewdirs))
        N, S = ro.shape[0], cfg.n_samples
        is_train = global_step is not None
        jitter = None
        if is_train:
            jitter = render_kwargs.get('jitter')
            jitter = torch.rand(N, device=dev) if jitter is None else jitter.to(dev).float().contiguous()
        f, i = dict(device=dev, dtype=torch.float32), dict(device=dev, dtype=torch.int32)
        sc_cap = N * S
        t_min, t_max = torch.empty(N, **f), torch.empty(N, **f)
        ray_start, count = torch.zeros(N + 1, **i), torch.zeros(1, **i)
        pts, ray_id, step_k, step = torch.empty(sc_cap, 3, **f), torch.empty(sc_cap, **i), torch.empty(sc_cap, **i), torch.empty(sc_cap, **f)
        keep = torch.empty(N * S, device=dev, dtype=torch.uint8)
        ops.sample_dense(cfg.pp, ro, rd, jitter, sc_cap, t_min, t_max, ray_start, count, pts, ray_id, step_k, step, keep)
        M = int(count.item())
        cap = max(M, 1)
        width = self.dim0
        ld = (width + 31) // 32 * 32
        lins = self._linears()

        def packed(mlp):
            W0 = torch.zeros(128, ld, device=dev)
            W0[:, :width] = mlp[0]
            out = [W0, mlp[1]]
            for k in range(1, len(lins)):
                out += [mlp[2 * k], mlp[2 * k + 1]]
            return out

        def unpack(flat):
            o, res = 0, []
            W0 = flat[o:o + 128 * ld].reshape(128, ld)[:, :width].contiguous(); o += 128 * ld
            res += [W0, flat[o:o + 128]]; o += 128
            for k in range(1, len(lins) - 1):
                res += [flat[o:o + 16384].reshape(128, 128), flat[o + 16384:o + 16512]]; o += 16512
            res += [flat[o:o + 384].reshape(3, 128), flat[o + 384:o + 387]]
            return res

        dist_o = torch.zeros(cap, **f)
        if M > 0:
            dist_o[:M] = (ro[ray_id[:M].long()] - pts[:M]).norm(dim=-1)
        st = dict(cfg=cfg, M=M, N=N, cap=cap, ld=ld, count=count, pts=pts, ray_id=ray_id, ray_start=ray_start, viewdirs=vd,
                  interval=float(render_kwargs['stepsize'] * self.voxel_size_ratio), dist_o=dist_o, packed=packed, unpack=unpack)
        mlp = []
        for l in lins:
            mlp += [l.weight, l.bias]
        if not self.k0[0].permute(1, 2, 3, 0).is_contiguous():
            self.k0.data = self.k0.data.contiguous(memory_format=torch.channels_last_3d)
        rgb_marched, last, depth, w, alpha, rgb, sel = _DVGORender.apply(self, st, self.density, self.k0, *mlp)
        # dense [N,S] views of the per-sample fields (dvgo_ori.py:367-377)
        flat = (ray_id[:M].long() * S + step_k[:M].long())
        dense = lambda v, fill: torch.full((N * S, *v.shape[1:]), fill, **f).index_put_((flat,), v[:M]).reshape(N, S, *v.shape[1:])
        alpha_d, w_d = dense(alpha, 0.), dense(w, 0.)
        rgb_d = dense(rgb, 0.5)
        mask_d = torch.zeros(N * S, dtype=torch.bool, device=dev).index_put_((flat,), sel[:M].bool()).reshape(N, S)
        aic = cumprod_exclusive(1 - alpha_d)
        aic = torch.cat([aic[:, :-1], last[:, None]], -1)        # differentiable last column
        return {'alphainv_cum': aic, 'weights': w_d, 'rgb_marched': rgb_marched, 'raw_alpha': alpha_d, 'raw_rgb': rgb_d,
                'depth': depth, 'disp': 1 / depth, 'mask': mask_d, 'mask_outbbox': ~keep.bool().reshape(N, S)}

    def extract_geometry(self, *a, **k):
        raise NotImplementedError('mesh extraction needs `mcubes`, which is out of scope of the hot path (DESIGN.md 8)')


# ---- remaining names of lib/dvgo_ori.py's module surface ---------------------------------------------------------
class MaskCache(torch.nn.Module):
    """lib/dvgo_ori.py:440-473: occupancy query alpha(xyz) >= threshold on the max-pooled density grid of a coarse-stage
    checkpoint.  The checkpoint is read weights-only (utils.load_checkpoint_file); the lookup is the HIP trilinear operator
    of DenseGrid (align_corners, zeros padding), the activation is the reference's post-activation form (`nearest`,
    `pre_act_density`, `in_act_density` select the reference's other three forms)."""

    def __init__(self, path, mask_cache_thres, ks=3):
        super().__init__()
        from .grid import DenseGrid
        from .utils import load_checkpoint_file
        st = load_checkpoint_file(path)
        kw = st['MaskCache_kwargs']
        self.mask_cache_thres = mask_cache_thres
        density = F.max_pool3d(torch.as_tensor(st['model_state_dict']['density'], dtype=torch.float32), kernel_size=ks,
                               padding=ks // 2, stride=1)
        self.act_shift, self.voxel_size_ratio = float(kw['act_shift']), float(kw['voxel_size_ratio'])
        self.nearest = bool(kw.get('nearest', False))
        self.pre_act_density, self.in_act_density = bool(kw.get('pre_act_density', False)), bool(kw.get('in_act_density', False))
        if self.nearest:
            raise NotImplementedError('MaskCache(nearest=True): no shipped configuration uses nearest-neighbour lookups')
        if self.pre_act_density:                             # (the reference then samples the RAW density: lib/dvgo_ori.py:463-465)
            pass
        elif self.in_act_density:
            density = F.softplus(density + self.act_shift)
        self.grid = DenseGrid(channels=1, world_size=list(density.shape[2:]), xyz_min=np.asarray(kw['xyz_min'], np.float32),
                              xyz_max=np.asarray(kw['xyz_max'], np.float32))
        self.grid.grid.data = density.contiguous(memory_format=torch.channels_last_3d)
        self.grid.grid.requires_grad = False
        self.register_buffer('xyz_min', self.grid.xyz_min.clone())
        self.register_buffer('xyz_max', self.grid.xyz_max.clone())

    @property
    def density(self):
        return self.grid.grid

    @torch.no_grad()
    def forward(self, xyz):
        val = self.grid(xyz.reshape(-1, 3)).reshape(xyz.shape[:-1])
        if self.pre_act_density:
            alpha = val
        elif self.in_act_density:
            alpha = 1 - torch.exp(-val * self.voxel_size_ratio)
        else:
            alpha = 1 - torch.exp(-F.softplus(val + self.act_shift) * self.voxel_size_ratio)
        return alpha >= self.mask_cache_thres


def extract_fields(bound_min, bound_max, resolution, query_func, N=64):
    """lib/dvgo_ori.py:679-693: query_func sampled on a resolution^3 lattice of the box, in blocks of N^3 points."""
    axes = [torch.linspace(float(bound_min[a]), float(bound_max[a]), resolution) for a in range(3)]
    u = np.zeros([resolution] * 3, dtype=np.float32)
    with torch.no_grad():
        for x0 in range(0, resolution, N):
            for y0 in range(0, resolution, N):
                for z0 in range(0, resolution, N):
                    xs, ys, zs = axes[0][x0:x0 + N], axes[1][y0:y0 + N], axes[2][z0:z0 + N]
                    pts = torch.stack(torch.meshgrid(xs, ys, zs, indexing='ij'), dim=-1).reshape(-1, 3)
                    val = query_func(pts).reshape(len(xs), len(ys), len(zs))
                    u[x0:x0 + len(xs), y0:y0 + len(ys), z0:z0 + len(zs)] = val.detach().cpu().numpy()
    return u


def extract_geometry(bound_min, bound_max, resolution, threshold, query_func, N=64):
    """lib/dvgo_ori.py:695-703 needs `mcubes.marching_cubes`, which is not available offline: the sampled field is what
    this package can deliver (extract_fields); triangulate it with any marching-cubes implementation."""
    raise NotImplementedError('extract_geometry (lib/dvgo_ori.py:695-703): marching cubes (`mcubes`) is not available '
                              'offline; use extract_fields(...) and triangulate the returned lattice elsewhere')
