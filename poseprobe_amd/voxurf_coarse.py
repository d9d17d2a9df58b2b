"""Drop-in for the reference's `lib/voxurf_coarse.py` module surface (imported there as `Model`):
`Voxurf`, `pose_model`, `Alphas2Weights`, `total_variation`, `get_rays*`, `get_training_rays*`.

Same constructor kwargs, forward/inference signatures, return-dict keys, `get_kwargs()` and `state_dict` names
(SURVEY.md 8b); everything per ray / per sample runs in libposeprobe_hip.so through ONE autograd node whose backward
is the hand-derived kernel chain (engine.RenderCore.backward).  There is no eager / CPU fallback: the module refuses
to run on non-CUDA tensors.
"""
import numpy as np
import torch
import torch.nn as nn

from . import camera, grid, ops
from .engine import (FlatParams, RenderCore, SceneConfig, Workspace, pack_rgbnet, pack_warp, unpack_rgbnet,
                     unpack_warp)
from .grid import channels_last_view


class pose_model(torch.nn.Module):
    """lib/voxurf_coarse.py:27-39"""

    def __init__(self, i_train=[], camera_noise=0.05):
        super().__init__()
        self.i_train = i_train
        self.camera_noise = camera_noise
        self.se3_refine = torch.nn.Parameter(torch.zeros((len(self.i_train), 6), dtype=torch.float32))
        self.se3_align_refine = torch.nn.Parameter(torch.zeros((1, 6), dtype=torch.float32))
        se3_noise = torch.randn(len(i_train), 6) * self.camera_noise
        self.pose_noise = _se3_to_SE3_host(se3_noise)


def _se3_to_SE3_host(wu):
    """Host (CPU) evaluation used once at construction for the constant pose noise (lib/camera.py:127-142)."""
    w, u = wu.split([3, 3], dim=-1)
    wx = camera.lie.skew_symmetric(w)
    theta = w.norm(dim=-1)[..., None, None]
    I = torch.eye(3)

    def taylor(kind):
        ans, denom = torch.zeros_like(theta), 1.
        for i in range(11):
            if kind == 0:
                if i > 0:
                    denom *= (2 * i) * (2 * i + 1)
            elif kind == 1:
                denom *= (2 * i + 1) * (2 * i + 2)
            else:
                denom *= (2 * i + 2) * (2 * i + 3)
            ans = ans + (-1) ** i * theta ** (2 * i) / denom
        return ans

    A, B, C = taylor(0), taylor(1), taylor(2)
    R = I + A * wx + B * wx @ wx
    V = I + B * wx + C * wx @ wx
    return torch.cat([R, V @ u[..., None]], dim=-1)


# ---------------------------------------------------------------------------------------------------------------
class _TotalVariation(torch.autograd.Function):
    @staticmethod
    def forward(ctx, v):
        C, (X, Y, Z) = v.shape[1], v.shape[2:]
        out = torch.zeros(1, device=v.device)
        ops.grid_tv_value(channels_last_view(v), (X, Y, Z), C, out)
        ctx.save_for_backward(v)
        return (out / 3 / v.numel())[0]

    @staticmethod
    def backward(ctx, g):
        (v,) = ctx.saved_tensors
        C, (X, Y, Z) = v.shape[1], v.shape[2:]
        gv = torch.zeros_like(v, memory_format=torch.channels_last_3d)
        ops.grid_tv_grad(channels_last_view(v), (X, Y, Z), C, 1.0 / (3 * v.numel()), g.reshape(1).contiguous().float(),
                         channels_last_view(gv))
        return gv


def total_variation(v, mask=None):
    """lib/voxurf_coarse.py:1298-1313 for the live case (no nonempty mask)."""
    if mask is not None:
        raise NotImplementedError('masked total_variation: MaskCache is never constructed on the live path')
    if not v[0].permute(1, 2, 3, 0).is_contiguous():
        v = v.contiguous(memory_format=torch.channels_last_3d)
    return _TotalVariation.apply(v)


def ray_start_from_ids(ray_id, N):
    """sorted ray_id[M] (int64/int32) -> exclusive prefix ray_start[N+1] int32 (replaces kernel.cu:607-636)."""
    counts = torch.bincount(ray_id.long(), minlength=N)
    rs = torch.zeros(N + 1, dtype=torch.int32, device=ray_id.device)
    rs[1:] = counts.cumsum(0).int()
    return rs


class Alphas2Weights(torch.autograd.Function):
    """lib/voxurf_coarse.py:1316-1332 over pp_alpha2weight_{fwd,bwd}."""

    @staticmethod
    def forward(ctx, alpha, ray_id, N):
        alpha = alpha.contiguous().float()
        M = alpha.shape[0]
        rs = ray_start_from_ids(ray_id, N)
        f = dict(device=alpha.device, dtype=torch.float32)
        w, T, last = torch.empty(M, **f), torch.empty(M, **f), torch.empty(N, **f)
        i_end = torch.empty(N, device=alpha.device, dtype=torch.int32)
        ops.alpha2weight_fwd(alpha, rs, N, w, T, last, i_end)
        ctx.save_for_backward(alpha, w, T, last, rs, i_end)
        ctx.n_rays = N
        return w, last

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_weights, grad_last):
        alpha, w, T, last, rs, i_end = ctx.saved_tensors
        g = torch.empty_like(alpha)
        ops.alpha2weight_bwd(alpha, w, T, last, rs, i_end, ctx.n_rays, grad_weights.contiguous().float(),
                             grad_last.contiguous().float(), g)
        return g, None, None


# ---------------------------------------------------------------------------------------------------------------
class _VoxurfRender(torch.autograd.Function):
    """One autograd node for Voxurf.forward: inputs rays + every trainable tensor, outputs the render dict."""

    @staticmethod
    def forward(ctx, model, ws, inv_s, pe_w, rays_o, rays_d, viewdirs, k0, sdf_alpha, sdf_beta, *mlp):
        core = model._core
        flat = FlatParams(rays_o.device, moments=False)
        rg = [(mlp[2 * i], mlp[2 * i + 1]) for i in range(4)]
        wp = [(mlp[8 + 2 * i], mlp[8 + 2 * i + 1]) for i in range(5)]
        flat.load_reference(sdf_alpha, sdf_beta, rg, wp)
        k0_cl = channels_last_view(k0)
        sdf_g = model.sdf.grid.detach()[0, 0].contiguous()
        core.forward(ws, k0_cl, sdf_g, flat.view('sdf_ab'), flat.view('rgbnet'), flat.view('warp'), inv_s, pe_w)
        ctx.model, ctx.ws, ctx.flat, ctx.inv_s, ctx.pe_w = model, ws, flat, inv_s, pe_w
        ctx.k0, ctx.sdf_g = k0, sdf_g
        ctx.set_materialize_grads(False)        # outputs the loss does not use arrive as None in backward, not as zero tensors
        M = ws.M
        depth = ws.t_min / rays_d.detach().norm(dim=-1) + ws.depth_acc
        outs = (ws.rgb_marched, ws.alphainv_last, ws.cum_weights.unsqueeze(-1), ws.weights[:M], ws.alpha[:M], ws.rgb[:M],
                depth, ws.gradient[:M], ws.sdf_deform[:M], ws.grad_deform[:M].reshape(M, 3, 3),
                ws.warp_out[:M, 3:4], ws.depth_acc)
        # k0_tv (lib/voxurf_coarse.py:1067: evaluated on every forward) rides on this node: its gradient is then ADDED IN PLACE to
        # the colour-grid gradient the render backward produces - a node of its own hands autograd a second dense 196 MB gradient
        # to allocate, zero-fill and sum (0.3 ms per step at 160^3)
        C, (X, Y, Z) = k0.shape[1], k0.shape[2:]
        tv = torch.zeros(1, device=k0.device)
        ops.grid_tv_value(k0_cl, (X, Y, Z), C, tv)
        return tuple(o.clone() for o in outs) + ((tv / 3 / k0.numel())[0],)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_rgbm, g_last, g_cw, g_w, g_alpha, g_rgb, g_depth, g_grad, g_sdfd, g_gdef, g_corr, g_nstep, g_tv):
        model, ws, flat = ctx.model, ctx.ws, ctx.flat
        core, M, cap = model._core, ws.M, ws.cap
        ws.alloc_backward()
        dev = ws.rays_o.device

        def padded(t, width=None):
            """upstream [M,...] gradient as the kernels want it: contiguous fp32, NO copy into a capacity-sized buffer - every
            kernel stops at the device-side sample count (M rows), so M rows are all that is ever read; None stays None (NULL =
            'no gradient')."""
            if t is None:
                return None
            return t.reshape(M, width).contiguous().float() if width else t.reshape(M).contiguous().float()

        # ray-level upstream gradients are read where they are (no staging copies)
        zero = lambda *shape: torch.zeros(*shape, device=dev)
        ws.g_rgbm = zero(ws.N, 3) if g_rgbm is None else g_rgbm.contiguous().float()
        ws.g_last = zero(ws.N) if g_last is None else g_last.contiguous().float()
        ws.g_cw = zero(ws.N) if g_cw is None else g_cw.reshape(-1).contiguous().float()
        # `depth` = t_min / |d| + sum w step and `_n_step` = sum w step share the per-sample path; only `depth` has the ray-level term
        g_depth = None if g_depth is None else g_depth.contiguous().float()
        if g_depth is None and g_nstep is None:
            g_depth_all = None
        elif g_nstep is None:
            g_depth_all = g_depth
        else:
            g_depth_all = g_nstep.contiguous().float() if g_depth is None else g_depth + g_nstep.contiguous().float()
        gg = padded(g_grad, 3)

        def capacity_sized(t, width=None):
            """the two upstream gradients that are ADDED to capacity-sized buffers by a torch op (rare: raw_alpha / raw_rgb in a loss)"""
            if t is None:
                return None
            b = torch.zeros((cap,) if width is None else (cap, width), device=dev)
            b[:M] = t.reshape(M, -1) if width else t.reshape(M)
            return b

        def add_gradient(w):
            if gg is not None:
                w.g_gradient[:M].add_(gg)

        k0_grad = torch.zeros_like(ctx.k0, memory_format=torch.channels_last_3d)
        core.backward(ws, channels_last_view(ctx.k0), ctx.sdf_g, flat.view('sdf_ab'), flat.view('rgbnet'), flat.view('warp'),
                      ctx.inv_s, ctx.pe_w, channels_last_view(k0_grad), flat.view('sdf_ab', 'grad'),
                      flat.view('rgbnet', 'grad'), flat.view('warp', 'grad'),
                      g_depth=g_depth_all, g_weights=padded(g_w),
                      g_gradient_ext=add_gradient, g_sdf_deform=padded(g_sdfd), g_grad_deform=padded(g_gdef, 9),
                      g_correction=padded(g_corr), g_alpha_ext=capacity_sized(g_alpha), g_rgb_ext=capacity_sized(g_rgb, 3))
        if g_tv is not None:                 # d k0_tv / d k0 added in place (read-modify-write of the gradient the scatter just filled)
            C, (X, Y, Z) = ctx.k0.shape[1], ctx.k0.shape[2:]
            ops.grid_tv_grad(channels_last_view(ctx.k0), (X, Y, Z), C, 1.0 / (3 * ctx.k0.numel()), g_tv.reshape(1).contiguous().float(),
                             channels_last_view(k0_grad))
        go, gd, gv = (torch.empty_like(ws.rays_o) for _ in range(3))
        ops.raygen_select_bwd(core.cfg.pp, None, None, None, 0, 0, True, ws.rays_o, ws.rays_d, ws.t_min, ws.ray_start,
                              ws.g_pts, ws.step, ws.g_view_s, None, None, None, g_depth, go, gd, gv, None)
        g = flat.export_grads()
        mlp_grads = []
        for W, b in g['rgbnet'] + g['warp']:
            mlp_grads += [W.contiguous(), b.contiguous()]
        return (None, None, None, None, go, gd, gv, k0_grad, g['sdf_alpha'], g['sdf_beta'], *mlp_grads)


def _ws_alloc_backward(self):
    if hasattr(self, 'g_alpha'):
        return
    f = dict(dtype=torch.float32, device=self.rays_o.device)
    e, N, cap = torch.empty, self.N, self.cap
    self.g_rgbm, self.g_last, self.g_cw = e(N, 3, **f), e(N, **f), e(N, **f)
    self.g_alpha, self.g_rgb = e(cap, **f), e(cap, 3, **f)
    self.g_feat = e(cap, ops.FEAT_LD, **f)
    self.g_gradient, self.g_pts, self.g_view_s = e(cap, 3, **f), e(cap, 3, **f), e(cap, 3, **f)
    self.g_warp_out = e(cap, 16, **f)
    self.scratch = e(3 * cap * 4 * 128 + 49152, **f)
    self.scratch_rgb = e(3 * cap * 128 + 49152, **f)


Workspace.alloc_backward = _ws_alloc_backward


class _WarpNet(nn.Module):
    """Parameter container with the reference's names: warp_network.progress and
    warp_network.deform_net.net.net.{0..4}.0.{weight,bias} (lib/deformation/deform_net.py:12-31, modules.py:43-124)."""

    def __init__(self, range_shape, hidden=128):
        super().__init__()
        self.progress = nn.Parameter(torch.tensor(0.))
        self.output_range = float(np.asarray(range_shape).max())
        dims = [3, hidden, hidden, hidden, hidden, 4]
        layers = []
        for i in range(5):
            lin = nn.Linear(dims[i], dims[i + 1])
            nn.init.kaiming_normal_(lin.weight, a=0.0, nonlinearity='relu', mode='fan_in')   # modules.py:136-139
            layers.append(nn.Sequential(lin))
        nn.init.zeros_(layers[-1][0].weight)                                                   # modules.py:166-171
        nn.init.zeros_(layers[-1][0].bias)
        fc = nn.Module()
        fc.net = nn.Sequential(*layers)
        bvp = nn.Module()
        bvp.net = fc
        self.deform_net = bvp

    def linears(self):
        return [seq[0] for seq in self.deform_net.net.net]


class _CrossingDense(torch.autograd.Function):
    """query_sdf_point_wocuda_wodeform (lib/voxurf_coarse.py:797-837) as one differentiable node: RAW template looked up with
    border padding at every dense slot, first sign change, linear zero crossing.  Inputs that carry gradients: rays_o, rays_d
    and t_min (the torch restatement of the slab test, so that d t_min / d ray is chained by autograd); outputs pts [N,3]
    and the dense SDF row [N,S] are differentiable, the hit mask is not."""

    @staticmethod
    def forward(ctx, rays_o, rays_d, t_min_graph, model, core, grid, jitter, dist, render_kwargs):
        cfg = core.cfg
        ro, rd = rays_o.detach().contiguous().float(), rays_d.detach().contiguous().float()
        N, S, dev = ro.shape[0], cfg.n_samples, ro.device
        pts_all, _, _, t_min, _ = model.sample_ray_ori(ro, rd, render_kwargs['near'], render_kwargs['far'],
                                                       render_kwargs['stepsize'], is_train=jitter is not None, jitter=jitter)
        dense = torch.empty(N * S, 1, device=dev)
        ops.grid_sample_fwd(cfg.pp, grid, 1, pts_all.reshape(-1, 3).contiguous(), 1, dense)
        dense = dense.reshape(N, S)
        pts_out = torch.empty(N, 3, device=dev)
        mask = torch.empty(N, device=dev, dtype=torch.uint8)
        sdf_d = torch.empty(N, S, device=dev)
        ops.sdf_first_crossing(dense, None, None, N, S, dist, t_min, ro, rd, sdf_d, pts_out, mask, None)
        mask = mask.bool()
        if ctx is not None:
            ctx.save_for_backward(ro, rd, t_min, sdf_d, grid, jitter)
            ctx.pp, ctx.dist = cfg.pp, dist
            ctx.mark_non_differentiable(mask)
        return pts_out, mask, sdf_d

    @staticmethod
    def backward(ctx, g_pts, _g_mask, g_sdf):
        ro, rd, t_min, sdf_d, grid, jitter = ctx.saved_tensors
        N, S = sdf_d.shape
        g_o, g_d, g_t = torch.empty_like(ro), torch.empty_like(rd), torch.empty_like(t_min)
        if g_pts is None and g_sdf is None:
            return (None,) * 9
        ops.sdf_crossing_dense_bwd(ctx.pp, grid, ro, rd, t_min, jitter, N, S, ctx.dist, sdf_d,
                                   None if g_pts is None else g_pts.contiguous().float(),
                                   None if g_sdf is None else g_sdf.contiguous().float(), g_o, g_d, g_t)
        return g_o, g_d, g_t, None, None, None, None, None, None


class Voxurf(torch.nn.Module):
    """lib/voxurf_coarse.py:45-1263 (constructor :49-229)."""

    def __init__(self, xyz_min, xyz_max, num_voxels=0, num_voxels_base=0, alpha_init=None, nearest=False,
                 mask_cache_path=None, fast_color_thres=0, rgbnet_dim=0, rgbnet_direct=False,
                 rgbnet_full_implicit=False, rgbnet_depth=3, rgbnet_width=128, posbase_pe=5, viewbase_pe=4,
                 geo_rgb_dim=3, grad_mode='interpolate', s_ratio=2000, s_start=0.2, s_learn=False, step_start=0,
                 smooth_ksize=0, smooth_sigma=1, camera_noise=0., barf_c2f=None, i_train=[], N_iters=20000,
                 flow_ckpt_path='', flow_backbone='', HW=[[512, 512]], sg_config=None, optimize_sdf=False,
                 range_shape=None, rect_size=None, **kwargs):
        super().__init__()
        if not (rgbnet_dim == 12 and rgbnet_direct and rgbnet_depth == 4 and rgbnet_width == 128 and geo_rgb_dim == 3
                and not rgbnet_full_implicit and smooth_ksize == 0 and not s_learn):
            raise NotImplementedError('the HIP path implements the shipped e2e configuration: rgbnet_dim=12, '
                                      'rgbnet_direct, depth 4, width 128, geo_rgb_dim=3, no smoothing, s_learn=False')
        self.i_train, self.N_iters, self.camera_noise, self.barf_c2f = i_train, N_iters, camera_noise, barf_c2f
        self.register_buffer('xyz_min', torch.Tensor(xyz_min))
        self.register_buffer('xyz_max', torch.Tensor(xyz_max))
        self.flow_ckpt_path, self.flow_backbone, self.HW = flow_ckpt_path, flow_backbone, HW
        self.range_shape, self.rect_size, self.sg_config = range_shape, rect_size, sg_config
        self.fast_color_thres, self.nearest = fast_color_thres, nearest
        self.s_ratio, self.s_start, self.s_learn, self.step_start = s_ratio, s_start, s_learn, step_start
        self.s_val = torch.ones(1) * s_start            # not a registered parameter on GPU (voxurf_coarse.py:94)
        self.optimize_sdf = optimize_sdf
        self.num_voxels_base = num_voxels_base
        self.voxel_size_base = ((self.xyz_max - self.xyz_min).prod() / self.num_voxels_base).pow(1 / 3)
        self.diagonal_length = torch.sqrt(torch.sum(self.xyz_max - self.xyz_min ** 2))   # sic (voxurf_coarse.py:102)
        self.alpha_init = alpha_init
        self.act_shift = np.log(1 / (1 - alpha_init) - 1)
        self._set_grid_resolution(num_voxels)
        self.progress = torch.nn.Parameter(torch.tensor(0.))
        self.sdf_alpha = torch.nn.Parameter(torch.Tensor([10.0]))
        self.sdf_beta = torch.nn.Parameter(torch.Tensor([2.0]))
        self.warp_network = _WarpNet(range_shape)
        self.sdf = grid.create_grid('DenseGrid', channels=1, world_size=self.world_size, xyz_min=self.xyz_min,
                                    xyz_max=self.xyz_max)
        for p in self.sdf.parameters():
            p.requires_grad = False                      # frozen template (voxurf_coarse.py:136-138)
        self.sdf.grid.data = self._cube_init(rect_size)
        self.rgbnet_kwargs = {'rgbnet_dim': rgbnet_dim, 'rgbnet_direct': rgbnet_direct,
                              'rgbnet_full_implicit': rgbnet_full_implicit, 'rgbnet_depth': rgbnet_depth,
                              'rgbnet_width': rgbnet_width, 'posbase_pe': posbase_pe, 'viewbase_pe': viewbase_pe}
        self.rgbnet_full_implicit, self.rgbnet_direct, self.geo_rgb_dim = rgbnet_full_implicit, rgbnet_direct, geo_rgb_dim
        self.k0_dim = rgbnet_dim
        self.k0 = grid.create_grid('DenseGrid', channels=self.k0_dim, world_size=self.world_size, xyz_min=self.xyz_min,
                                   xyz_max=self.xyz_max)
        self.register_buffer('posfreq', torch.FloatTensor([(2 ** i) for i in range(posbase_pe)]))
        self.register_buffer('viewfreq', torch.FloatTensor([(2 ** i) for i in range(viewbase_pe)]))
        dim0 = (3 + 3 * posbase_pe * 2) + (3 + 3 * viewbase_pe * 2) + self.k0_dim + geo_rgb_dim
        self.rgbnet = nn.Sequential(
            nn.Linear(dim0, rgbnet_width), nn.ReLU(inplace=True),
            *[nn.Sequential(nn.Linear(rgbnet_width, rgbnet_width), nn.ReLU(inplace=True))
              for _ in range(rgbnet_depth - 2)],
            nn.Linear(rgbnet_width, 3))
        nn.init.constant_(self.rgbnet[-1].bias, 0)
        self.mask_cache_path, self.mask_cache_thres = mask_cache_path, 1e9
        self.mask_cache = self.nonempty_mask = None
        # frozen convs kept only for state_dict compatibility (voxurf_coarse.py:231-265)
        self.grad_conv = nn.Conv3d(1, 3, (3, 3, 3), stride=1, padding=1, padding_mode='replicate')
        self.tv_smooth_conv = nn.Conv3d(1, 1, (3, 3, 3), stride=1, padding=1, padding_mode='replicate')
        for p in list(self.grad_conv.parameters()) + list(self.tv_smooth_conv.parameters()):
            p.requires_grad = False
        self.grad_mode = grad_mode
        self._core = None

    # ---- construction helpers -----------------------------------------------------------------------------
    def _set_grid_resolution(self, num_voxels):
        self.num_voxels = num_voxels
        self.voxel_size = ((self.xyz_max - self.xyz_min).prod() / num_voxels).pow(1 / 3)
        self.world_size = ((self.xyz_max - self.xyz_min) / self.voxel_size).long()
        self.voxel_size_ratio = self.voxel_size / self.voxel_size_base

    def _cube_init(self, rect_size):
        from .params_init import cube_sdf
        cfg = type('C', (), {})()
        cfg.xyz_min, cfg.xyz_max = self.xyz_min.cpu().numpy(), self.xyz_max.cpu().numpy()
        cfg.world_size = [int(v) for v in self.world_size.tolist()]
        return cube_sdf(cfg, list(rect_size))

    def _scene(self, render_kwargs):
        key = (float(render_kwargs['stepsize']), float(render_kwargs['near']), float(render_kwargs['far']),
               float(render_kwargs['bg']), int(self.num_voxels))
        if self._core is None or self._core_key != key:
            cfg = SceneConfig(self.xyz_min.cpu().numpy(), self.xyz_max.cpu().numpy(), int(self.num_voxels),
                              stepsize=key[0], near=key[1], far=key[2], bg=key[3], N_iters=self.N_iters,
                              s_ratio=self.s_ratio, s_start=self.s_start, step_start=self.step_start,
                              barf_c2f=None if self.barf_c2f is None else tuple(self.barf_c2f),
                              posbase_pe=self.rgbnet_kwargs['posbase_pe'], viewbase_pe=self.rgbnet_kwargs['viewbase_pe'],
                              k0_dim=self.k0_dim, out_range=self.warp_network.output_range)
            self._core, self._core_key = RenderCore(cfg, ctx=getattr(self, 'pp_ctx', None)), key
        return self._core

    def get_kwargs(self):
        return {'xyz_min': self.xyz_min.cpu().numpy(), 'xyz_max': self.xyz_max.cpu().numpy(),
                'num_voxels': self.num_voxels, 'num_voxels_base': self.num_voxels_base, 'alpha_init': self.alpha_init,
                'nearest': self.nearest, 'mask_cache_path': self.mask_cache_path,
                'mask_cache_thres': self.mask_cache_thres, 'fast_color_thres': self.fast_color_thres,
                'geo_rgb_dim': self.geo_rgb_dim, 'flow_ckpt_path': self.flow_ckpt_path,
                'flow_backbone': self.flow_backbone, 'sg_config': self.sg_config, 'HW': self.HW,
                'i_train': self.i_train, 'N_iters': self.N_iters, 'camera_noise': self.camera_noise,
                'range_shape': self.range_shape, 'rect_size': self.rect_size, **self.rgbnet_kwargs}

    def get_MaskCache_kwargs(self):
        return {'xyz_min': self.xyz_min.cpu().numpy(), 'xyz_max': self.xyz_max.cpu().numpy(),
                'act_shift': self.act_shift, 'voxel_size_ratio': self.voxel_size_ratio, 'nearest': self.nearest}

    @torch.no_grad()
    def maskout_near_cam_vox(self, cam_o, near):
        """voxurf_coarse.py:379-391 (writes sdf = 1 near the given points)."""
        g = self.sdf.grid
        xs = [torch.linspace(self.xyz_min[i], self.xyz_max[i], g.shape[2 + i], device=g.device) for i in range(3)]
        xyz = torch.stack(torch.meshgrid(*xs, indexing='ij'), -1)
        nearest = torch.stack([(xyz.unsqueeze(-2) - co).pow(2).sum(-1).sqrt().amin(-1) for co in cam_o.split(100)]).amin(0)
        g[nearest[None, None] <= near] = 1

    @torch.no_grad()
    def scale_volume_grid(self, num_voxels):
        self._set_grid_resolution(num_voxels)
        self.sdf.scale_volume_grid(self.world_size)
        if self.k0_dim > 0:
            self.k0.scale_volume_grid(self.world_size)
        self._core = None

    def k0_total_variation(self, k0_tv=1., k0_grad_tv=0.):
        if k0_grad_tv > 0:
            raise NotImplementedError
        return total_variation(self.k0.grid) if k0_tv > 0 else 0

    # ---- parameter gathering ----------------------------------------------------------------------------------
    def _mlp_tensors(self):
        rg = [self.rgbnet[0], self.rgbnet[2][0], self.rgbnet[3][0], self.rgbnet[4]]
        out = []
        for lin in rg + self.warp_network.linears():
            out += [lin.weight, lin.bias]
        return out

    def _set_progress(self, global_step):
        v = 1. if global_step is None else global_step / self.N_iters
        self.progress.data.fill_(v)
        self.warp_network.progress.data.fill_(v)
        return v

    def _inv_s(self, global_step, is_train):
        """neus_alpha_from_sdf_scatter's s_val bookkeeping (voxurf_coarse.py:485-495)."""
        if is_train:
            s_val = 1. / (global_step + self.s_ratio / self.s_start - self.step_start) * self.s_ratio
            self.s_val = torch.ones(1) * s_val
        else:
            s_val = 0
        inv_s = float((torch.ones(1) / self.s_val)[0])
        return s_val, inv_s

    def _check_inputs(self, *ts):
        for t in ts:
            if not t.is_cuda:
                raise RuntimeError('poseprobe_amd.Voxurf runs on the HIP path only: inputs must be CUDA tensors')

    # ---- sampling (API parity helpers) ------------------------------------------------------------------------
    def _sample_dense(self, core, rays_o, rays_d, jitter):
        cfg = core.cfg
        N, S = rays_o.shape[0], cfg.n_samples
        dev = rays_o.device
        f, i = dict(device=dev, dtype=torch.float32), dict(device=dev, dtype=torch.int32)
        sc = max(N * S, N)
        buf = dict(t_min=torch.empty(N, **f), t_max=torch.empty(N, **f), ray_start=torch.zeros(N + 1, **i),
                   count=torch.zeros(1, **i), pts=torch.empty(sc, 3, **f), ray_id=torch.empty(sc, **i),
                   step_k=torch.empty(sc, **i), step=torch.empty(sc, **f),
                   keep=torch.empty(N * S, device=dev, dtype=torch.uint8))
        ops.sample_dense(cfg.pp, rays_o, rays_d, jitter, sc, buf['t_min'], buf['t_max'], buf['ray_start'], buf['count'],
                         buf['pts'], buf['ray_id'], buf['step_k'], buf['step'], buf['keep'])
        buf['M'] = int(buf['count'].item())     # the reference synchronises here too (boolean-mask compaction)
        return buf

    def sample_ray_ori(self, rays_o, rays_d, near, far, stepsize, is_train=False, jitter=None, **render_kwargs):
        """Dense-layout outputs of the reference's sample_ray_ori (voxurf_coarse.py:697-719):
        rays_pts[N,S,3], mask_outbbox[N,S], step[N,S], t_min[N], t_max[N] - reconstructed from the compacted sampler."""
        core = self._scene(dict(near=near, far=far, stepsize=stepsize, bg=render_kwargs.get('bg', 0)))
        self._check_inputs(rays_o, rays_d)
        N, S = rays_o.shape[0], core.cfg.n_samples
        if is_train and jitter is None:
            jitter = torch.rand(N, device=rays_o.device)
        b = self._sample_dense(core, rays_o.contiguous().float(), rays_d.contiguous().float(), jitter if is_train else None)
        keep = b['keep'].bool().reshape(N, S)
        rng = torch.arange(S, device=rays_o.device)[None].float().repeat(N, 1)
        if is_train:
            rng = rng + jitter.reshape(-1, 1)
        step = stepsize * self.voxel_size.to(rays_o.device) * rng
        interpx = b['t_min'][..., None] + step / rays_d.norm(dim=-1, keepdim=True)
        pts = rays_o[..., None, :] + rays_d[..., None, :] * interpx[..., None]
        return pts, ~keep, step, b['t_min'], b['t_max']


    # ---- surface-point queries (voxurf_coarse.py:734-920) -------------------------------------------------------
    def _query_crossing(self, rays_o, rays_d, global_step, use_deform, mapped, render_kwargs):
        """mapped=True: query_sdf_point_wocuda (mapped SDF at the in-bbox samples, optional warp), forward only - its one caller,
        the PnP hand-off (recon_scene.py:290-296), detaches the points.  mapped=False: query_sdf_point_wocuda_wodeform, which
        the live loop differentiates w.r.t. the rays (recon_scene.py:336-340 while <= 2 views are active): an autograd node
        whose backward is pp_sdf_crossing_dense_bwd."""
        self._check_inputs(rays_o, rays_d)
        core = self._scene(dict(bg=0, **{k: render_kwargs[k] for k in ('near', 'far', 'stepsize')}))
        cfg = core.cfg
        N, S, dev = rays_o.shape[0], cfg.n_samples, rays_o.device
        is_train = global_step is not None
        jitter = None
        if is_train:
            jitter = render_kwargs.get('jitter')
            jitter = torch.rand(N, device=dev) if jitter is None else jitter.to(dev).float().contiguous()
        dist = float(np.float32(cfg.stepsize) * np.float32(cfg.voxel_size))
        if not mapped:
            needs_grad = torch.is_grad_enabled() and (rays_o.requires_grad or rays_d.requires_grad)
            # t_min enters the surface point explicitly; its dependence on the ray (slab test) is plain torch algebra
            t_min = self._entry_distance(rays_o, rays_d, render_kwargs['near'], render_kwargs['far']) if needs_grad else None
            grid = self.sdf.grid[0, 0].contiguous()
            if needs_grad:
                pts, mask, sdf_d = _CrossingDense.apply(rays_o, rays_d, t_min, self, core, grid, jitter, dist,
                                                        render_kwargs)
            else:
                with torch.no_grad():
                    pts, mask, sdf_d = _CrossingDense.forward(None, rays_o, rays_d, None, self, core, grid, jitter, dist,
                                                              render_kwargs)
            return pts, mask, sdf_d
        with torch.no_grad():
            ro, rd = rays_o.detach().contiguous().float(), rays_d.detach().contiguous().float()
            pts_out = torch.empty(N, 3, device=dev)
            mask = torch.empty(N, device=dev, dtype=torch.uint8)
            sdf_d = torch.empty(N, S, device=dev)
            sb = self._sample_dense(core, ro, rd, jitter)
            M = sb['M']
            cap = max(M, 1)
            vd = rd / rd.norm(dim=-1, keepdim=True)
            warp_out = torch.zeros(cap, 16, device=dev)
            flat = FlatParams(dev)
            mlp = self._mlp_tensors()
            flat.load_reference(self.sdf_alpha, self.sdf_beta, [(mlp[2 * k], mlp[2 * k + 1]) for k in range(4)],
                                [(mlp[8 + 2 * k], mlp[8 + 2 * k + 1]) for k in range(5)])
            if use_deform and M > 0:
                acts = torch.empty(4, cap * 4, 128, device=dev)
                ops.warp_fwd(flat.view('warp'), sb['pts'], sb['count'], cap, cfg.out_range, acts, warp_out, core.ctx)
            alpha, grad, sdf_final = torch.empty(cap, device=dev), torch.empty(cap, 3, device=dev), torch.zeros(cap, device=dev)
            if M > 0:
                ops.geometry_fwd(cfg.pp, self.sdf.grid[0, 0].contiguous(), flat.view('sdf_ab'), sb['pts'], warp_out,
                                 vd.contiguous(), sb['ray_id'], sb['count'], cap, 1.0, alpha, grad, sdf_final, None, None)
            ops.sdf_first_crossing(sdf_final, sb['ray_start'], sb['step_k'], N, S, dist, sb['t_min'], ro, rd, sdf_d,
                                   pts_out, mask, None)
        return pts_out, mask.bool(), sdf_d

    def _query_finish(self, rays_o, rays_d, pts, mask, sdf_d, keep_dim, return_depth, t_min_fn=None):
        if keep_dim:
            return pts, mask, sdf_d
        if return_depth:
            depth = ((pts - rays_o) * rays_d).sum(-1) / (rays_d * rays_d).sum(-1) / 1.0
            # interpx = z0 + t_min/|d| in the reference (:792); with unit rays_d this equals the distance along the ray
            return pts[mask], (depth * rays_d.norm(dim=-1))[mask]
        return pts[mask]

    def query_sdf_point_wocuda(self, rays_o, rays_d, global_step=None, keep_dim=False, return_depth=False,
                               use_deform=False, **render_kwargs):
        """voxurf_coarse.py:734-795 (forward only here: the reference's one caller detaches the points, recon_scene.py:290-296)."""
        pts, mask, sdf_d = self._query_crossing(rays_o, rays_d, global_step, use_deform, True, render_kwargs)
        return self._query_finish(rays_o, rays_d, pts, mask, sdf_d, keep_dim, return_depth)

    def query_sdf_point_wocuda_wodeform(self, rays_o, rays_d, global_step=None, keep_dim=False, return_depth=False,
                                        **render_kwargs):
        """voxurf_coarse.py:797-837; differentiable w.r.t. rays_o / rays_d (pose gradient of the reprojection loss)."""
        pts, mask, sdf_d = self._query_crossing(rays_o, rays_d, global_step, False, False, render_kwargs)
        return self._query_finish(rays_o, rays_d, pts, mask, sdf_d, keep_dim, return_depth)

    def _entry_distance(self, rays_o, rays_d, near, far):
        """t_min of the slab test (voxurf_coarse.py:701-705) restated in torch: the kernels produce the same values in the
        sampler, this copy exists so that d t_min / d ray reaches the pose wherever t_min enters an output explicitly."""
        vec = torch.where(rays_d == 0, torch.full_like(rays_d, 1e-6), rays_d)
        lo, hi = self.xyz_min.to(rays_o.device), self.xyz_max.to(rays_o.device)
        return torch.minimum((hi - rays_o) / vec, (lo - rays_o) / vec).amax(-1).clamp(min=near, max=far)

    def query_sdf_point_wocuda_render(self, rays_o, rays_d, global_step=None, keep_dim=False, return_depth=False,
                                      use_deform=True, **render_kwargs):
        """voxurf_coarse.py:839-920: expected depth from the rendering weights; differentiable (pose, warp, grid) because
        it runs through the same autograd node as forward()."""
        viewdirs = rays_d / rays_d.norm(dim=-1, keepdim=True)
        out = self.forward(rays_o, rays_d, viewdirs, use_deform=use_deform, global_step=global_step, **render_kwargs)
        nrm = rays_d.norm(dim=-1)
        n_step = out['_n_step']                                            # = sum_i w_i step_i (differentiable)
        t_min = self._entry_distance(rays_o, rays_d, render_kwargs['near'], render_kwargs['far'])
        depth = t_min[..., None] + n_step[..., None] / nrm[..., None]
        mask = n_step > 0.
        if keep_dim:
            return rays_o + rays_d * depth, mask, depth
        pts = rays_o[mask] + rays_d[mask] * depth[mask]
        return (pts, depth[mask]) if return_depth else pts

    def _pe_weights(self, cfg, progress, dev):
        """Coarse-to-fine weights of the step on the device.  They are a function of `progress` that is CONSTANT outside the
        c2f window (all zeros before barf_c2f[0], all ones after barf_c2f[1]): the upload (a pageable host-to-device copy, i.e. a
        synchronisation point) only happens when the values change."""
        w = cfg.pe_weights(progress)
        key = (w.tobytes(), str(dev))
        cached = getattr(self, '_pe_cache', None)
        if cached is None or cached[0] != key:
            self._pe_cache = cached = (key, torch.from_numpy(w).to(dev))
        return cached[1]

    # ---- forward ----------------------------------------------------------------------------------------------
    def forward(self, rays_o, rays_d, viewdirs, use_deform=True, global_step=None, **render_kwargs):
        """voxurf_coarse.py:922-1092.  Extension: render_kwargs['jitter'] ([N] in [0,1)) overrides the internally drawn
        per-ray jitter (the reference draws it from the global CUDA generator, :713)."""
        if not use_deform:
            raise NotImplementedError('use_deform=False is never taken on the live path (recon_scene.py:603)')
        if self.fast_color_thres > 0:
            raise NotImplementedError('fast_color_thres > 0 (all shipped configs use 0)')
        self._check_inputs(rays_o, rays_d, viewdirs)
        core = self._scene(render_kwargs)
        cfg = core.cfg
        is_train = global_step is not None
        progress = self._set_progress(global_step)
        N = len(rays_o)
        ro, rd, vd = rays_o.contiguous().float(), rays_d.contiguous().float(), viewdirs.contiguous().float()
        jitter = None
        if is_train:
            jitter = render_kwargs.get('jitter')
            jitter = torch.rand(N, device=ro.device) if jitter is None else jitter.to(ro.device).float().contiguous()
        sb = self._sample_dense(core, ro.detach(), rd.detach(), jitter)
        M = sb['M']
        cap = max((M + 4095) // 4096 * 4096, 4096)
        ws = Workspace(N, cap, ro.device, sample_capacity=sb['pts'].shape[0], backward=False, ctx=core.ctx)
        ws.M = M
        ws.rays_o, ws.rays_d, ws.viewdirs = ro.detach(), rd.detach(), vd.detach()
        for k in ('t_min', 't_max', 'ray_start', 'count', 'pts', 'ray_id', 'step_k', 'step'):
            setattr(ws, k, sb[k])
        s_val, inv_s = self._inv_s(global_step, is_train)
        pe_w = self._pe_weights(cfg, progress, ro.device)
        self.k0.ensure_layout()
        outs = _VoxurfRender.apply(self, ws, inv_s, pe_w, ro, rd, vd, self.k0.grid, self.sdf_alpha, self.sdf_beta,
                                   *self._mlp_tensors())
        (rgb_marched, alphainv_last, cum_weights, weights, alpha, rgb, depth, gradient, sdf_deform, grad_deform,
         correction, n_step, k0_tv) = outs
        if torch.is_grad_enabled() and (rays_o.requires_grad or rays_d.requires_grad):
            # same value, but with the explicit t_min / |d| term differentiable w.r.t. the ray (voxurf_coarse.py:1057)
            depth = self._entry_distance(rays_o, rays_d, render_kwargs['near'], render_kwargs['far']) / rays_d.norm(dim=-1) + n_step
        normal_marched = None
        if render_kwargs.get('render_grad', False):
            normal = gradient.detach() / (gradient.detach().norm(2, -1, keepdim=True) + 1e-6)
            normal_marched = torch.zeros(N, 3, device=ro.device).index_add_(0, ws.ray_id[:M].long(),
                                                                             weights.detach().unsqueeze(-1) * normal)
        return {
            'alphainv_cum': alphainv_last, 'weights': weights, 'cum_weights': cum_weights, 'rgb_marched': rgb_marched,
            'normal_marched': normal_marched, 'raw_alpha': alpha, 'raw_rgb': rgb, 'depth': depth, 'disp': 1 / depth,
            'mask': sb['keep'].bool(), 'mask_outbbox': torch.zeros(M, dtype=torch.bool, device=ro.device),
            'gradient': gradient, 's_val': s_val, 'k0_tv': k0_tv, 'sdf_deform': sdf_deform,
            'grad_deform': grad_deform, 'sdf_correct': correction,
            '_t_min': ws.t_min, '_n_step': n_step,   # extras (not in the reference dict) for query_sdf_point_wocuda_render
        }

    @torch.no_grad()
    def inference(self, rays_o, rays_d, viewdirs, global_step=None, **render_kwargs):
        """voxurf_coarse.py:1094-1222: variable-length sampler (sample_pts_on_rays semantics), forward chain only."""
        self._check_inputs(rays_o, rays_d, viewdirs)
        core = self._scene(render_kwargs)
        cfg = core.cfg
        is_train = global_step is not None
        progress = self._set_progress(global_step)
        N = len(rays_o)
        dev = rays_o.device
        ro, rd, vd = rays_o.contiguous().float(), rays_d.contiguous().float(), viewdirs.contiguous().float()
        f, i = dict(device=dev, dtype=torch.float32), dict(device=dev, dtype=torch.int32)
        sc = N * (cfg.n_samples + 2)
        t_min, t_max, n_steps = torch.empty(N, **f), torch.empty(N, **f), torch.empty(N, **i)
        ray_start, count = torch.zeros(N + 1, **i), torch.zeros(1, **i)
        pts, ray_id, step_id = torch.empty(sc, 3, **f), torch.empty(sc, **i), torch.empty(sc, **i)
        ops.sample_var(cfg.pp, ro, rd, sc, t_min, t_max, n_steps, ray_start, count, pts, ray_id, step_id)
        M = int(count.item())
        cap = max((M + 4095) // 4096 * 4096, 4096)
        ws = Workspace(N, cap, dev, sample_capacity=sc, backward=False, keep_activations=False, ctx=core.ctx)
        ws.M = M
        ws.rays_o, ws.rays_d, ws.viewdirs = ro, rd, vd
        ws.t_min, ws.t_max, ws.ray_start, ws.count, ws.pts, ws.ray_id, ws.step_k = t_min, t_max, ray_start, count, pts, ray_id, step_id
        dist = float(np.float32(cfg.stepsize) * np.float32(cfg.voxel_size))
        ws.step = step_id[:].float() * dist
        s_val, inv_s = self._inv_s(global_step, is_train)
        pe_w = self._pe_weights(cfg, progress, dev)
        self.k0.ensure_layout()
        flat = FlatParams(dev)
        mlp = self._mlp_tensors()
        flat.load_reference(self.sdf_alpha, self.sdf_beta, [(mlp[2 * k], mlp[2 * k + 1]) for k in range(4)],
                            [(mlp[8 + 2 * k], mlp[8 + 2 * k + 1]) for k in range(5)])
        core.forward(ws, channels_last_view(self.k0.grid), self.sdf.grid[0, 0].contiguous(), flat.view('sdf_ab'),
                     flat.view('rgbnet'), flat.view('warp'), inv_s, pe_w)
        gradient = ws.gradient[:M]
        normal = gradient / (gradient.norm(2, -1, keepdim=True) + 1e-6)
        weights = ws.weights[:M]
        normal_marched = torch.zeros(N, 3, device=dev).index_add_(0, ray_id[:M].long(), weights.unsqueeze(-1) * normal)
        depth = ws.depth_acc.clone()
        return {
            'alphainv_cum': ws.alphainv_last, 'weights': weights, 'cum_weights': ws.cum_weights.unsqueeze(-1),
            'rgb_marched': ws.rgb_marched, 'normal_marched': normal_marched, 'raw_alpha': ws.alpha[:M],
            'raw_rgb': ws.rgb[:M], 'depth': depth, 'disp': 1 / depth, 'mask': torch.ones_like(step_id[:M]).long(),
            'mask_outbbox': None, 'gradient': gradient,
            'gradient_error': ((torch.linalg.norm(gradient, ord=2, dim=-1) - 1.0) ** 2).mean(), 's_val': s_val,
            'ray_id': ray_id[:M].long(), 'step_id': step_id[:M].long(),
        }


    @torch.no_grad()
    def inference_rays(self, rays_o, rays_d, viewdirs, global_step=None, **render_kwargs):
        """The per-RAY outputs of `inference` (rgb_marched, alphainv_cum, cum_weights, depth, disp, normal_marched) WITHOUT the
        host round trip for the sample count: what a whole-view driver keeps of a chunk (lib/nvs_fun.py:74-85 discards every
        per-sample entry).  Same kernels on the same samples as `inference` - the chunk's buffers are sized for the sampler's
        worst case (N (S + 2) samples) and kept on the module, every kernel stops at the device-side count, the normals are
        composited by the marching kernel itself - so a view is 40 chunks enqueued back to back instead of 40 syncs."""
        self._check_inputs(rays_o, rays_d, viewdirs)
        core = self._scene(render_kwargs)
        cfg = core.cfg
        is_train = global_step is not None
        progress = self._set_progress(global_step)
        N, dev = len(rays_o), rays_o.device
        ro, rd, vd = rays_o.contiguous().float(), rays_d.contiguous().float(), viewdirs.contiguous().float()
        sc = N * (cfg.n_samples + 2)
        cap = (sc + 4095) // 4096 * 4096
        key = (N, cap, str(dev), id(core))
        cache = getattr(self, '_ray_cache', None)
        if cache is None or cache['key'] != key:
            f, i = dict(device=dev, dtype=torch.float32), dict(device=dev, dtype=torch.int32)
            ws = Workspace(N, cap, dev, sample_capacity=cap, backward=False, keep_activations=False, ctx=core.ctx)
            ws.n_steps = torch.empty(N, **i)
            ws.nrm = torch.empty(cap, 3, **f)
            ws.normal_marched = torch.empty(N, 3, **f)
            self._ray_cache = cache = dict(key=key, ws=ws, flat=FlatParams(dev, moments=False))
        ws, flat = cache['ws'], cache['flat']
        ws.rays_o, ws.rays_d, ws.viewdirs = ro, rd, vd
        ops.sample_var(cfg.pp, ro, rd, cap, ws.t_min, ws.t_max, ws.n_steps, ws.ray_start, ws.count, ws.pts, ws.ray_id, ws.step_k)
        dist = float(np.float32(cfg.stepsize) * np.float32(cfg.voxel_size))
        torch.mul(ws.step_k, dist, out=ws.step)                     # step_id * dist (fp32 product of an exact integer)
        s_val, inv_s = self._inv_s(global_step, is_train)
        pe_w = self._pe_weights(cfg, progress, dev)
        self.k0.ensure_layout()
        mlp = self._mlp_tensors()
        flat.load_reference(self.sdf_alpha, self.sdf_beta, [(mlp[2 * k], mlp[2 * k + 1]) for k in range(4)],
                            [(mlp[8 + 2 * k], mlp[8 + 2 * k + 1]) for k in range(5)])
        sdf_g = self.sdf.grid[0, 0].contiguous()
        ops.warp_fwd(flat.view('warp'), ws.pts, ws.count, ws.cap, cfg.out_range, ws.warp_acts, ws.warp_out, core.ctx)
        ops.geometry_fwd(cfg.pp, sdf_g, flat.view('sdf_ab'), ws.pts, ws.warp_out, ws.viewdirs, ws.ray_id, ws.count, ws.cap, inv_s,
                         ws.alpha, ws.gradient, ws.sdf_final, ws.sdf_deform, ws.grad_deform)
        ops.color_feat_fwd(cfg.pp, channels_last_view(self.k0.grid), ws.pts, ws.viewdirs, ws.ray_id, ws.gradient, pe_w, ws.count,
                           ws.cap, ws.feat)
        ops.rgbnet_fwd(flat.view('rgbnet'), ws.feat, ws.count, ws.cap, ws.rgb_acts, ws.rgb, core.ctx)
        # rows past the count hold stale values: harmless, the marching kernel walks ray_start ranges only
        torch.div(ws.gradient, ws.gradient.norm(2, -1, keepdim=True) + 1e-6, out=ws.nrm)
        ops.march_fwd(ws.alpha, ws.rgb, ws.step, ws.nrm, ws.ray_start, N, cfg.bg, ws.weights, ws.T, ws.alphainv_last, ws.i_end,
                      ws.rgb_marched, ws.rgb_pre, ws.cum_weights, ws.depth_acc, ws.normal_marched)
        depth = ws.depth_acc.clone()
        return {'alphainv_cum': ws.alphainv_last.clone(), 'cum_weights': ws.cum_weights.clone().unsqueeze(-1),
                'rgb_marched': ws.rgb_marched.clone(), 'normal_marched': ws.normal_marched.clone(), 'depth': depth, 'disp': 1 / depth,
                's_val': s_val}


# ---------------------------------------------------------------------------------------------------------------
# rays (lib/voxurf_coarse.py:1339-1631)
# ---------------------------------------------------------------------------------------------------------------
class _RaysAtPixels(torch.autograd.Function):
    """c2w[V,3,4] + flat pixel indices -> rays, differentiable w.r.t. c2w (pp_raygen_select_{fwd,bwd})."""

    @staticmethod
    def forward(ctx, c2w, ray_idx, intr, H, W, inverse_y, normalize):
        N = ray_idx.shape[0]
        dev = c2w.device
        o, d, v = (torch.empty(N, 3, device=dev) for _ in range(3))
        sc = ops.make_scene([0, 0, 0], [1, 1, 1], [2, 2, 2], 1.0, 1.0, 0., 1., 0.)
        c2w_c = c2w.contiguous().float()
        ops.raygen_select_fwd(sc, ray_idx, c2w_c, intr, H, W, inverse_y, normalize, None, None, o, d, v, None, None)
        ctx.save_for_backward(c2w_c, ray_idx, intr, o, d)
        ctx.meta = (H, W, inverse_y, normalize, sc)
        return o, d, v

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, go, gd, gv):
        c2w, ray_idx, intr, o, d = ctx.saved_tensors
        H, W, inverse_y, normalize, sc = ctx.meta
        if not normalize:
            raise NotImplementedError('pose gradients are implemented for the Voxurf ray variant (normalised rays_d)')
        N, V = ray_idx.shape[0], c2w.shape[0]
        dev = c2w.device
        g = torch.zeros(V, 3, 4, device=dev)
        zero_start = torch.zeros(N + 1, dtype=torch.int32, device=dev)
        dummy = torch.zeros(1, 3, device=dev)
        # no samples: the kernel only projects the per-ray grads (rays_d and viewdirs are one tensor) onto c2w
        ops.raygen_select_bwd(sc, ray_idx, c2w, intr, H, W, inverse_y, o, d, torch.zeros(N, device=dev), zero_start, dummy,
                              torch.zeros(1, device=dev), None, go.contiguous().float(), gd.contiguous().float(),
                              gv.contiguous().float(), None, None, None, None, g)
        return g, None, None, None, None, None, None


def _intr_of(K, device):
    K = torch.as_tensor(np.asarray(K.detach().cpu() if isinstance(K, torch.Tensor) else K), dtype=torch.float32)
    if K.dim() == 2:
        K = K[None]
    return torch.stack([K[:, 0, 0], K[:, 1, 1], K[:, 0, 2], K[:, 1, 2]], -1).to(device).contiguous()


def get_rays(H, W, K, c2w, inverse_y, flip_x, flip_y, mode='center', normalize=False):
    if mode != 'center':
        raise NotImplementedError("only mode='center' is used on the live path")
    dev = c2w.device
    ii = torch.arange(W, device=dev)
    jj = torch.arange(H, device=dev)
    if flip_x:
        ii = ii.flip(0)
    if flip_y:
        jj = jj.flip(0)
    idx = (jj[:, None] * W + ii[None, :]).reshape(-1).int().contiguous()
    o, d, v = _RaysAtPixels.apply(c2w[None, :3, :4], idx, _intr_of(K, dev), H, W, bool(inverse_y), normalize)
    return o.reshape(H, W, 3), d.reshape(H, W, 3), v.reshape(H, W, 3)


def get_rays_of_a_view(H, W, K, c2w, ndc, inverse_y, flip_x, flip_y, mode='center'):
    """voxurf_coarse.py:1402-1407: rays_d = viewdirs = d/|d|."""
    rays_o, rays_d, viewdirs = get_rays(H, W, K, c2w, inverse_y, flip_x, flip_y, mode, normalize=True)
    if ndc:
        rays_o, rays_d = ndc_rays(H, W, K[0][0], 1., rays_o, rays_d)
    return rays_o, rays_d, viewdirs


def ndc_rays(H, W, focal, near, rays_o, rays_d):
    """lib/voxurf_coarse.py:1371-1399.  No PoseProbe configuration sets ndc=True (forward-facing LLFF scenes only) and the
    HIP sampler has no NDC parametrisation, so the name exists for API parity and refuses instead of rendering wrongly."""
    raise NotImplementedError('ndc_rays (lib/voxurf_coarse.py:1371-1399): NDC rays are not used by any PoseProbe '
                              'configuration and are not part of the HIP path')


def select_training_rays(ray_idx, rgb_tr_ori, mask_tr_ori, train_poses, HW, Ks, inverse_y=True):
    """Index-first equivalent of get_training_rays_flatten(...)[indices] (voxurf_coarse.py:1518-1549 +
    recon_scene.py:598-600): only the N selected pixels are generated instead of all V*H*W rays."""
    H, W = int(HW[0][0]), int(HW[0][1])
    dev = train_poses.device
    idx = ray_idx.to(dev).int().contiguous()
    o, d, v = _RaysAtPixels.apply(train_poses[:, :3, :4], idx, _intr_of(Ks, dev), H, W, bool(inverse_y), True)
    rgb = rgb_tr_ori.reshape(-1, 3)[idx.long()]
    mask = mask_tr_ori.reshape(-1, 1)[idx.long()]
    return rgb, mask, o, d, v


def get_training_rays_flatten(rgb_tr_ori, mask_tr_ori, train_poses, HW, Ks, ndc, inverse_y, flip_x, flip_y):
    """voxurf_coarse.py:1518-1549 (all V*H*W rays; kept for API parity - prefer select_training_rays)."""
    assert len(rgb_tr_ori) == len(train_poses) and len(rgb_tr_ori) == len(Ks) and len(rgb_tr_ori) == len(HW)
    ro, rd, vd, imsz = [], [], [], []
    for c2w, img, (H, W), K in zip(train_poses, rgb_tr_ori, HW, Ks):
        o, d, v = get_rays_of_a_view(int(H), int(W), K, c2w, ndc, inverse_y, flip_x, flip_y)
        ro.append(o.flatten(0, 1)), rd.append(d.flatten(0, 1)), vd.append(v.flatten(0, 1))
        imsz.append(int(H) * int(W))
    rgb_tr = torch.cat([im.flatten(0, 1) for im in rgb_tr_ori])
    mask_tr = torch.cat([m.flatten(0, 1) for m in mask_tr_ori])
    return rgb_tr, mask_tr, torch.cat(ro), torch.cat(rd), torch.cat(vd), imsz


def get_training_rays(rgb_tr, mask_tr, train_poses, HW, Ks, ndc, inverse_y, flip_x, flip_y):
    """voxurf_coarse.py:1494-1515"""
    H, W = HW[0]
    K = Ks[0]
    outs = [get_rays_of_a_view(int(H), int(W), K, c2w, ndc, inverse_y, flip_x, flip_y) for c2w in train_poses]
    return (rgb_tr, mask_tr, torch.stack([o[0] for o in outs]), torch.stack([o[1] for o in outs]),
            torch.stack([o[2] for o in outs]), [1] * len(rgb_tr))


# ----------------------------------------------------------------------------------------------------------------------
# Remaining names of the reference's module surface.  None of the shipped end-to-end configurations reaches
# them (`ray_sampler='flatten'`, no mask cache), but a caller that imports `lib.voxurf_coarse as Model` finds them here, on
# the same HIP operators as the live path.
# ----------------------------------------------------------------------------------------------------------------------
class MaskCache(nn.Module):
    """Known-free-space query of a coarse SDF (lib/voxurf_coarse.py:1271-1296): forward(xyz) = sdf(xyz) < threshold with the
    trilinear lookup of DenseGrid (align_corners, zeros padding).  `path`: an .npz with `sdf_grid_xyz [X,Y,Z]`, `xyz_min`,
    `xyz_max` (weights-only load).  The reference reads a pickled dict from an .npy; such a file is refused rather than
    unpickled - convert it once with numpy on the side that trusts it."""

    def __init__(self, path, mask_cache_thres, ks=3):
        super().__init__()
        try:
            d = np.load(path, allow_pickle=False)
            sdf0, lo, hi = d['sdf_grid_xyz'], d['xyz_min'], d['xyz_max']
        except (ValueError, KeyError, IndexError, TypeError) as e:
            raise ValueError(f'MaskCache: {path} is not an .npz with sdf_grid_xyz / xyz_min / xyz_max (a pickled dict, as '
                             'lib/voxurf_coarse.py:1274 reads, is not loaded: it could execute code)') from e
        self.mask_cache_thres = mask_cache_thres
        self.nearest = False
        self.grid = grid.DenseGrid(channels=1, world_size=list(sdf0.shape), xyz_min=np.asarray(lo, np.float32),
                                   xyz_max=np.asarray(hi, np.float32))
        self.grid.grid.data = torch.tensor(np.asarray(sdf0, np.float32))[None, None].contiguous(memory_format=torch.channels_last_3d)
        self.grid.grid.requires_grad = False
        self.register_buffer('xyz_min', self.grid.xyz_min.clone())
        self.register_buffer('xyz_max', self.grid.xyz_max.clone())

    @property
    def sdf(self):
        return self.grid.grid

    @torch.no_grad()
    def forward(self, xyz):
        return self.grid(xyz.reshape(-1, 3)).reshape(xyz.shape[:-1]) < self.mask_cache_thres


def get_rays_of_a_view_semantic(H, W, K, c2w, ndc, inverse_y, flip_x, flip_y, background_sampler, object_sampler,
                                boundary_sampler, mode='center'):
    """lib/voxurf_coarse.py:1410-1453: rays at pixels drawn 20 % / 30 % / 50 % from the background / boundary / object
    pixel lists (each sampler = (ys, xs)); returns rays_o, rays_d, viewdirs and [x, y] of the drawn pixels.  Like the
    reference's variant (and unlike get_rays_of_a_view) rays_d stays un-normalised."""
    if ndc or flip_x or flip_y or mode != 'center':
        raise NotImplementedError('semantic sampler: only mode="center" without flips / NDC (the reference\'s flips index a '
                                  '1-D tensor along dimension 1 and would fail as well)')
    n = min(boundary_sampler[0].shape[0], object_sampler[0].shape[0])
    parts = []
    for sampler, share in ((background_sampler, 0.2), (boundary_sampler, 0.3), (object_sampler, 0.5)):
        pick = torch.randint(low=0, high=sampler[0].shape[0], size=[int(n * share)])
        parts.append((sampler[0][pick], sampler[1][pick]))
    y = torch.cat([p[0] for p in parts]).long()
    x = torch.cat([p[1] for p in parts]).long()
    dev = c2w.device
    idx = (y * W + x).to(dev).int().contiguous()
    o, d, v = _RaysAtPixels.apply(c2w[None, :3, :4], idx, _intr_of(K, dev), H, W, bool(inverse_y), False)
    return o, d, v, [x, y]


def get_training_rays_semantic(rgb_tr_ori, mask_tr_ori, train_poses, HW, Ks, ndc, inverse_y, flip_x, flip_y, samplers):
    """lib/voxurf_coarse.py:1456-1491."""
    rgb, msk, ro, rd, vd, imsz = [], [], [], [], [], []
    for i, (c2w, img, mask, (H, W), K) in enumerate(zip(train_poses, rgb_tr_ori, mask_tr_ori, HW, Ks)):
        o, d, v, (x, y) = get_rays_of_a_view_semantic(int(H), int(W), K, c2w, ndc, inverse_y, flip_x, flip_y,
                                                     samplers['background'][i], samplers['object'][i], samplers['boundary'][i])
        x, y = x.to(img.device), y.to(img.device)
        rgb.append(img[y, x]), msk.append(mask[y, x]), ro.append(o), rd.append(d), vd.append(v), imsz.append(o.shape[0])
    return torch.cat(rgb), torch.cat(msk), torch.cat(ro), torch.cat(rd), torch.cat(vd), imsz


def get_training_rays_in_maskcache_sampling_grad(rgb_tr_ori, train_poses, HW, Ks, ndc, inverse_y, flip_x, flip_y, model,
                                                 render_kwargs):
    """lib/voxurf_coarse.py:1552-1588: the rays of every view that have at least one sample inside the bounding box AND in
    the mask cache's known space; differentiable w.r.t. the poses (the selection itself is not)."""
    assert len(rgb_tr_ori) == len(train_poses) and len(rgb_tr_ori) == len(Ks) and len(rgb_tr_ori) == len(HW)
    if getattr(model, 'mask_cache', None) is None:
        raise ValueError('in_maskcache sampling needs model.mask_cache (a MaskCache)')
    rgb, ro, rd, vd, imsz = [], [], [], [], []
    CHUNK = 4096
    for c2w, img, (H, W), K in zip(train_poses, rgb_tr_ori, HW, Ks):
        H, W = int(H), int(W)
        assert img.shape[:2] == (H, W)
        o, d, v = get_rays_of_a_view(H, W, K, c2w, ndc, inverse_y, flip_x, flip_y)
        o, d, v = o.reshape(-1, 3), d.reshape(-1, 3), v.reshape(-1, 3)
        hit = torch.zeros(H * W, dtype=torch.bool, device=o.device)
        with torch.no_grad():
            for b in range(0, H * W, CHUNK):
                pts, out, _, _, _ = model.sample_ray_ori(rays_o=o[b:b + CHUNK].detach(), rays_d=d[b:b + CHUNK].detach(),
                                                         **render_kwargs)
                inside = ~out
                inside[inside.clone()] &= model.mask_cache(pts[inside])
                hit[b:b + CHUNK] = inside.any(-1)
        rgb.append(img.reshape(-1, img.shape[-1])[hit.to(img.device)]), ro.append(o[hit]), rd.append(d[hit]), vd.append(v[hit])
        imsz.append(int(hit.sum()))
    return torch.cat(rgb), torch.cat(ro), torch.cat(rd), torch.cat(vd), imsz


@torch.no_grad()
def get_training_rays_in_maskcache_sampling(rgb_tr_ori, train_poses, HW, Ks, ndc, inverse_y, flip_x, flip_y, model,
                                            render_kwargs):
    """lib/voxurf_coarse.py:1591-1631 (the no-grad twin of the function above)."""
    return get_training_rays_in_maskcache_sampling_grad(rgb_tr_ori, train_poses, HW, Ks, ndc, inverse_y, flip_x, flip_y, model,
                                                        render_kwargs)


def _no_mesh(name, where):
    def method(self, *a, **k):
        raise NotImplementedError(f'{name} ({where}): marching cubes (`mcubes`) is not available offline and mesh '
                                  'extraction is outside the hot path (DESIGN.md 8); query the SDF with '
                                  'Voxurf.query_sdf_point_wocuda* or sample `sdf.grid` directly')
    method.__name__ = name
    return method


Voxurf.extract_deform_geometry = _no_mesh('extract_deform_geometry', 'lib/voxurf_coarse.py:1224-1248')
Voxurf.extract_geometry = _no_mesh('extract_geometry', 'lib/voxurf_coarse.py:1250-1263')
