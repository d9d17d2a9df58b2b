"""Scene branch on MI355X: drop-in for the reference's `NeRF` module (lib/bg_nerf/source/models/frequency_nerf.py:72-343)
on the render path used by `Graph.render` (lib/bg_nerf/source/models/renderer.py:532-627).

Same constructor arguments (`opt` tree), same parameter names (`mlp_feat.{0..7}.{weight,bias}`, `mlp_rgb.{0,1}.{weight,bias}`,
`progress`) and shapes - a reference `state_dict` loads as is - and the same methods:

    forward_samples(opt, center, ray, depth_samples, embedder_pts, embedder_view, mode)  -> rgb_samples, density_samples
    forward(opt, points_3D_samples, ray, embedder_pts, embedder_view, mode)
    composite(opt, ray, pred_dict, depth_samples)  -> rgb, rgb_var, depth, depth_var, opacity, weights, all_cumulated

Everything numerical runs in the HIP kernels of csrc/pp_nerf.hip through the C ABI (pp_nerf_fwd / pp_nerf_bwd /
pp_nerf_composite_fwd / pp_nerf_composite_bwd); there is no eager fallback.  The parameters are views into ONE packed
device buffer in the layout the kernels read (`pp_nerf_layout`), so nothing is repacked per step and the whole network is a
single flat tensor for the fused Adam kernel (SceneEngine below).

Supported architecture = the reference default (lib/bg_nerf/train_settings/default_config.py:90-105) that every shipped
PoseProbe configuration uses; anything else raises NotImplementedError rather than silently taking another path.
"""
import math

import torch

from . import ops


class Options(dict):
    """Attribute-style settings tree (stands in for the reference's easydict)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def default_options(barf_c2f=(0.4, 0.7), sample_intvs=128, white_bg=False):
    """The settings the render path reads, at the values PoseProbe trains with (lib/bg_nerf/train_settings/default_config.py
    :90-120, joint_pose_nerf_training/dtu/sparf.py)."""
    posenc = Options(include_pi_in_posenc=True, add_raw_3D_points=True, add_raw_rays=True, log_sampling=True, L_3D=10, L_view=4)
    arch = Options(layers_feat=[None] + [256] * 8, layers_feat_fine=None, layers_rgb=[None, 128, 3], skip=[4], posenc=posenc,
                   density_activ='softplus', tf_init=True)
    nerf = Options(view_dep=True, density_noise_reg=False, setbg_opaque=white_bg, sample_intvs=sample_intvs,
                   sample_stratified=True, fine_sampling=False, sample_intvs_fine=128, rand_rays=1024,
                   depth=Options(param='metric', range=[1, 0]))
    return Options(arch=arch, nerf=nerf, barf_c2f=list(barf_c2f) if barf_c2f else None, mask_img=False,
                   camera=Options(ndc=False), huber_loss_for_photometric=True)


def sample_depth(opt, batch_size, num_rays, n_samples, depth_range, mode=None, device='cuda', generator=None):
    """Stratified depth samples [B, num_rays, n_samples, 1] (renderer.py:665-700)."""
    depth_min, depth_max = depth_range
    if opt.nerf.sample_stratified and mode not in ('val', 'eval', 'test'):
        rand = torch.rand(batch_size, num_rays, n_samples, 1, device=device, generator=generator)
    else:
        rand = 0.5 * torch.ones(batch_size, num_rays, n_samples, 1, device=device)
    rand = rand + torch.arange(n_samples, device=device)[None, None, :, None].float()
    depth = rand / n_samples * (depth_max - depth_min) + depth_min
    return {'metric': depth, 'inverse': 1 / (depth + 1e-8)}[opt.nerf.depth.param]


def sample_depth_from_pdf(weights, n_samples_coarse, n_samples_fine, depth_range, det, generator=None, grid=None):
    """Inverse-transform sampling from the coarse weights (renderer.py:702-738); weights [B, R, N] -> [B, R, Nf, 1].
    `grid` [Nf + 1] replays a recorded draw of the (shared across rays) uniform grid."""
    depth_min, depth_max = depth_range
    dev = weights.device
    pdf = weights / (weights.sum(dim=-1, keepdim=True) + 1e-6)
    cdf = torch.cat([torch.zeros_like(pdf[..., :1]), pdf.cumsum(dim=-1)], dim=-1)
    if grid is not None:
        grid = grid.to(dev)
    elif det:
        grid = torch.linspace(0, 1, n_samples_fine + 1, device=dev)
    else:
        grid = torch.rand(n_samples_fine + 1, generator=generator).to(dev)
    unif = (0.5 * (grid[:-1] + grid[1:])).repeat(*cdf.shape[:-1], 1)
    idx = torch.searchsorted(cdf, unif, right=True)
    bins = torch.linspace(depth_min, depth_max, n_samples_coarse + 1, device=dev).repeat(*cdf.shape[:-1], 1)
    lo, hi = (idx - 1).clamp(min=0), idx.clamp(max=n_samples_coarse)
    d_lo, d_hi = bins.gather(2, lo), bins.gather(2, hi)
    c_lo, c_hi = cdf.gather(2, lo), cdf.gather(2, hi)
    t = (unif - c_lo) / (c_hi - c_lo + 1e-8)
    return (d_lo + t * (d_hi - d_lo))[..., None]


class _LinearParams(torch.nn.Module):
    """Holds `weight` / `bias` (views into the packed block) under the reference's `torch.nn.Linear` names."""

    def __init__(self, weight, bias):
        super().__init__()
        self.weight = torch.nn.Parameter(weight)
        self.bias = torch.nn.Parameter(bias)


class _Workspace:
    """Per-shape buffers that no pending backward depends on: the backward pass's scratch (used only inside one pp_nerf_bwd
    call), the device-side sample count, and an activation block for passes that nobody will differentiate through autograd
    (no_grad forwards, SceneEngine's own forward / backward pairs).  A forward pass that autograd may differentiate owns its
    activation block (see _NerfSamples): any number of them may be pending at the same shape, as in the reference."""

    def __init__(self, R, S, dev):
        M = R * S
        self.acts_floats, s = ops.nerf_workspace(M, R)
        f = dict(dtype=torch.float32, device=dev)
        self.R, self.S, self.dev = R, S, dev
        self._acts = None
        self.scratch = torch.zeros(s, **f)
        self.count = torch.tensor([M], dtype=torch.int32, device=dev)

    @property
    def acts(self):
        if self._acts is None:
            self._acts = torch.empty(self.acts_floats, dtype=torch.float32, device=self.dev)
        return self._acts

    def new_acts(self):
        return torch.empty(self.acts_floats, dtype=torch.float32, device=self.dev)


class _NerfSamples(torch.autograd.Function):
    """center[R,3], ray[R,3], depth[R,S] -> rgb_samples[R*S,3], density_samples[R*S]."""

    @staticmethod
    def forward(ctx, net, center, ray, depth, *params):
        R, S = depth.shape
        ws = net._workspace(R, S)
        f = dict(dtype=torch.float32, device=depth.device)
        center, ray, depth = center.contiguous().float(), ray.contiguous().float(), depth.contiguous().float()
        rgb_s, dens = torch.empty(R * S, 3, **f), torch.empty(R * S, **f)
        # a pass that may be differentiated keeps its own activations until its backward has run (the caching allocator
        # recycles the block afterwards); anything else shares the per-shape block
        acts = ws.new_acts() if any(ctx.needs_input_grad) else ws.acts
        ops.nerf_fwd(net.flat, center, ray, depth, net.band_weights(), ws.count, R, S, acts, rgb_s, dens, net.ctx)
        ctx.net, ctx.ws, ctx.acts = net, ws, acts
        ctx.save_for_backward(ray, depth, rgb_s)
        return rgb_s, dens

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_rgb_s, g_dens):
        net, ws, acts = ctx.net, ctx.ws, ctx.acts
        ray, depth, rgb_s = ctx.saved_tensors
        ctx.acts = None                                            # once_differentiable: the block is free after this call
        R, S = depth.shape
        f = dict(dtype=torch.float32, device=depth.device)
        pgrad = torch.zeros_like(net.flat)
        g_center, g_ray = torch.empty(R, 3, **f), torch.empty(R, 3, **f)
        ops.nerf_bwd(net.flat, ray, depth, ws.count, R, S, acts, rgb_s, g_rgb_s.contiguous().float(),
                     g_dens.contiguous().float(), ws.scratch, pgrad, g_center, g_ray, net.ctx)
        return (None, g_center, g_ray, None, *net._views(pgrad))


class _Composite(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rgb_s, dens, depth, ray, white_bg):
        R, S = depth.shape
        f = dict(dtype=torch.float32, device=depth.device)
        rgb_s, dens, depth, ray = (t.contiguous().float() for t in (rgb_s, dens, depth, ray))
        rgb, d, op, w = torch.empty(R, 3, **f), torch.empty(R, **f), torch.empty(R, **f), torch.empty(R, S, **f)
        cum, rv, dv = torch.empty(R, **f), torch.empty(R, **f), torch.empty(R, **f)
        ops.nerf_composite_fwd(rgb_s, dens, depth, ray, R, S, white_bg, rgb, d, op, w, cum, rv, dv)
        ctx.save_for_backward(rgb_s, dens, depth, ray, w)
        ctx.white_bg = white_bg
        ctx.mark_non_differentiable(cum, rv, dv)
        return rgb, d, op, w.clone(), cum, rv, dv

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_rgb, g_d, g_op, g_w, *_):
        rgb_s, dens, depth, ray, w = ctx.saved_tensors
        R, S = depth.shape
        f = dict(dtype=torch.float32, device=depth.device)
        g_rgb_s, g_dens, g_ray = torch.empty(R * S, 3, **f), torch.empty(R * S, **f), torch.empty(R, 3, **f)
        ops.nerf_composite_bwd(rgb_s, dens, depth, ray, w, R, S, ctx.white_bg, g_rgb.contiguous(), g_d.contiguous(),
                               g_op.contiguous(), g_w.contiguous(), g_rgb_s, g_dens, g_ray)
        return g_rgb_s, g_dens, None, g_ray, None


class NeRF(torch.nn.Module):
    """MLP network corresponding to NeRF (frequency_nerf.py:72)."""

    L_3D, L_VIEW = 10, 4

    def __init__(self, opt, is_fine_network=False, device='cuda', options=None):
        """options: {name: value} of a private pp_context for this network's kernels (e.g. {'nerf_split': 0}: every product on
        the fp32 MFMA instructions); None = the host's default context."""
        super().__init__()
        self.opt = opt
        self.ctx = ops.Context(**options) if options else None
        self._check_supported(opt, is_fine_network)
        off = ops.nerf_layout()
        self._off = off
        dev = torch.device(device)
        self.flat = torch.zeros(off[22], dtype=torch.float32, device=dev)
        w, b = self._views(self.flat, split=True)
        self.mlp_feat = torch.nn.ModuleList([_LinearParams(w[i], b[i]) for i in range(8)])
        self.mlp_rgb = torch.nn.ModuleList([_LinearParams(w[8 + i], b[8 + i]) for i in range(2)])
        if opt.arch.tf_init:
            self.initialize()
        # a Parameter so that it is checkpointed, as in the reference (frequency_nerf.py:79-85)
        self.progress = torch.nn.Parameter(torch.tensor(1. if opt.barf_c2f is None else 0., device=dev))
        self._ws = {}

    @staticmethod
    def _check_supported(opt, is_fine_network):
        a, p = opt.arch, opt.arch.posenc
        layers = a.layers_feat_fine if is_fine_network and a.layers_feat_fine is not None else a.layers_feat
        ok = (list(layers[1:]) == [256] * 8 and list(a.layers_rgb[1:]) == [128, 3] and list(a.skip) == [4] and p.L_3D == 10
              and p.L_view == 4 and p.add_raw_3D_points and p.add_raw_rays and p.log_sampling and p.include_pi_in_posenc
              and a.density_activ == 'softplus' and opt.nerf.view_dep and not opt.nerf.density_noise_reg)
        if not ok:
            raise NotImplementedError('bg_nerf: only the reference default architecture (8 x 256, skip [4], L_3D = 10, '
                                      'L_view = 4 with raw coordinates, softplus density, view dependent colour, no density '
                                      'noise) has HIP kernels')

    def _views(self, flat, split=False):
        """Per-parameter views of a packed block in registration order (mlp_feat.i.weight, .bias, ..., mlp_rgb.i.*)."""
        o = self._off
        w, b = [], []
        for l, k in enumerate((64, 256, 256, 256, 320, 256, 256)):
            used = {0: 63, 4: 319}.get(l, k)
            w.append(flat[o[2 * l]:o[2 * l] + 256 * k].view(256, k)[:, :used])
            b.append(flat[o[2 * l + 1]:o[2 * l + 1] + 256])
        w.append(flat[o[16]:o[16] + 257 * 256].view(257, 256))          # density row + 256 feature rows
        b.append(flat[o[17]:o[17] + 257])
        w.append(flat[o[18]:o[18] + 128 * 288].view(128, 288)[:, :283])
        b.append(flat[o[19]:o[19] + 128])
        w.append(flat[o[20]:o[20] + 3 * 128].view(3, 128))
        b.append(flat[o[21]:o[21] + 3])
        if split:
            return w, b
        return [t for pair in zip(w, b) for t in pair]

    def _apply(self, fn, *a, **k):
        probe = fn(torch.empty(0, device=self.flat.device))
        if probe.device != self.flat.device or probe.dtype != torch.float32:
            raise RuntimeError('bg_nerf.NeRF lives in one packed fp32 device buffer: construct it with device=... instead of '
                               'moving / casting it')
        return self

    def state_dict(self, *args, **kwargs):
        """Compact copies under the reference's names: the parameters are views into the packed block, and serialising a view
        would write the whole 2 MB block once per tensor."""
        sd = super().state_dict(*args, **kwargs)
        for k in list(sd.keys()):
            sd[k] = sd[k].clone()
        return sd

    def initialize(self):
        """TensorFlow-style Xavier initialisation (frequency_nerf.py:128-148)."""
        gain = torch.nn.init.calculate_gain('relu')
        with torch.no_grad():
            for li, lin in enumerate(self.mlp_feat):
                if li == len(self.mlp_feat) - 1:
                    torch.nn.init.xavier_uniform_(lin.weight[:1])
                    torch.nn.init.xavier_uniform_(lin.weight[1:], gain=gain)
                else:
                    torch.nn.init.xavier_uniform_(lin.weight, gain=gain)
                lin.bias.zero_()
            torch.nn.init.xavier_uniform_(self.mlp_rgb[0].weight, gain=gain)
            torch.nn.init.xavier_uniform_(self.mlp_rgb[1].weight)
            self.mlp_rgb[0].bias.zero_(), self.mlp_rgb[1].bias.zero_()

    # ------------------------------------------------------------------------------------------------------------
    def band_weights(self):
        """Device tensor [14]: coarse-to-fine weights of the 10 point bands and the 4 view bands (frequency_nerf.py:255-262),
        computed on the device from `progress` (the trainer updates it with in-place fills on `.data`, which no version
        counter sees) so the schedule never forces a host synchronisation; same elementwise op order as the reference."""
        dev = self.flat.device
        if self.opt.barf_c2f is None:
            if getattr(self, '_band_ones', None) is None:
                self._band_ones = torch.ones(14, dtype=torch.float32, device=dev)
            return self._band_ones
        start, end = self.opt.barf_c2f
        if dev.type == 'cuda':
            # one launch (pp_nerf_band_weights); a ring of outputs, because a pending autograd backward may still hold the
            # weights of an earlier forward while `progress` moves on
            if getattr(self, '_band_ring', None) is None:
                self._band_ring, self._band_i = torch.empty(16, 14, dtype=torch.float32, device=dev), 0
            out = self._band_ring[self._band_i]
            self._band_i = (self._band_i + 1) % 16
            ops.nerf_band_weights(self.progress.data, float(start), float(end), self.L_3D, self.L_VIEW, out)
            return out
        if getattr(self, '_band_consts', None) is None:
            L = torch.tensor([float(self.L_3D)] * self.L_3D + [float(self.L_VIEW)] * self.L_VIEW, device=dev)
            k = torch.tensor(list(range(self.L_3D)) + list(range(self.L_VIEW)), dtype=torch.float32, device=dev)
            self._band_consts = (L, k)
        L, k = self._band_consts
        alpha = (self.progress.data - start) / (end - start) * L
        return (1 - (alpha - k).clamp_(min=0, max=1).mul_(math.pi).cos_()) / 2

    def _workspace(self, R, S):
        ws = self._ws.get((R, S))
        if ws is None:
            if len(self._ws) >= 4:
                self._ws.clear()
            ws = self._ws[(R, S)] = _Workspace(R, S, self.flat.device)
        return ws

    def _params(self):
        return [p for lin in list(self.mlp_feat) + list(self.mlp_rgb) for p in (lin.weight, lin.bias)]

    def forward_samples(self, opt, center, ray, depth_samples, embedder_pts=None, embedder_view=None, mode=None):
        """center, ray [B, N, 3]; depth_samples [B, N, S, 1] (frequency_nerf.py:268-288).  The embedders are accepted for
        signature compatibility; the encoding is part of the kernel."""
        B, N, S = depth_samples.shape[:3]
        rgb_s, dens = _NerfSamples.apply(self, center.reshape(B * N, 3), ray.reshape(B * N, 3),
                                         depth_samples.reshape(B * N, S), *self._params())
        return dict(rgb_samples=rgb_s.view(B, N, S, 3), density_samples=dens.view(B, N, S))

    def forward(self, opt, points_3D_samples, ray, embedder_pts=None, embedder_view=None, mode=None):
        """points_3D_samples [B, N, S, 3], ray [B, N, 3] (frequency_nerf.py:172-227): every point is a one-sample ray at
        depth 0 whose origin is the point itself."""
        B, N, S = points_3D_samples.shape[:3]
        pts = points_3D_samples.reshape(B * N * S, 3)
        rays = ray[:, :, None, :].expand(B, N, S, 3).reshape(B * N * S, 3)
        rgb_s, dens = _NerfSamples.apply(self, pts, rays, torch.zeros(B * N * S, 1, device=pts.device), *self._params())
        return dict(rgb_samples=rgb_s.view(B, N, S, 3), density_samples=dens.view(B, N, S))

    def composite(self, opt, ray, pred_dict, depth_samples):
        """Quadrature compositing (frequency_nerf.py:290-343); adds rgb, rgb_var, depth, depth_var, opacity, weights,
        all_cumulated to pred_dict.  depth_samples are treated as constants, as produced by the stratified sampler."""
        B, N, S = depth_samples.shape[:3]
        white = bool(opt.nerf.setbg_opaque or opt.mask_img)
        rgb, d, op, w, cum, rv, dv = _Composite.apply(pred_dict['rgb_samples'].reshape(B * N * S, 3),
                                                     pred_dict['density_samples'].reshape(B * N * S),
                                                     depth_samples.reshape(B * N, S), ray.reshape(B * N, 3), white)
        pred_dict.update(rgb=rgb.view(B, N, 3), rgb_var=rv.view(B, N, 1), depth=d.view(B, N, 1), depth_var=dv.view(B, N, 1),
                         opacity=op.view(B, N, 1), weights=w.view(B, N, S, 1), all_cumulated=cum.view(B, N))
        return pred_dict


def _cam2world(X, pose_w2c):
    """camera -> world for points X [B, N, 3] under world-to-camera poses [B, 3, 4] (utils/camera.py:321-338)."""
    Rinv = pose_w2c[..., :3].transpose(-1, -2)
    c2w = torch.cat([Rinv, -Rinv @ pose_w2c[..., 3:]], dim=-1)
    return torch.cat([X, torch.ones_like(X[..., :1])], dim=-1) @ c2w.transpose(-1, -2)


def get_center_and_ray_at_pixels(pose_w2c, pixels, intr):
    """Camera centres and un-normalised ray directions of given pixels (utils/camera.py:384-416): pixels [N, 2] (shared by
    all images) or [B, N, 2]; differentiable w.r.t. the poses through a handful of per-ray torch ops."""
    B = len(pose_w2c)
    xy = pixels.unsqueeze(0).repeat(B, 1, 1) if pixels.dim() == 2 else pixels
    grid = torch.cat([xy, torch.ones_like(xy[..., :1])], dim=-1) @ intr.inverse().transpose(-1, -2)
    center = _cam2world(torch.zeros_like(grid), pose_w2c)
    return center, _cam2world(grid, pose_w2c) - center


def get_center_and_ray(pose_w2c, H, W, intr):
    """All H * W pixel centres (x + 0.5, y + 0.5), row-major (utils/camera.py:347-381)."""
    dev = pose_w2c.device
    Y, X = torch.meshgrid(torch.arange(H, dtype=torch.float32, device=dev) + 0.5,
                          torch.arange(W, dtype=torch.float32, device=dev) + 0.5, indexing='ij')
    return get_center_and_ray_at_pixels(pose_w2c, torch.stack([X, Y], dim=-1).view(-1, 2), intr)


class SceneRenderer:
    """The render path of the reference's `Graph` (renderer.py:532-627): rays from poses, stratified coarse samples, coarse
    NeRF + compositing and - once `iter` has passed `ratio_start_fine_sampling_at_x * max_iter` - the fine NeRF on the union of
    the coarse samples and inverse-transform samples of the coarse weights.  Returns the same keys (`*_fine` for the second
    pass).  `rand` (optional) replays recorded uniform draws: [coarse [B, N, S, 1], fine grid [Nf + 1]]."""

    def __init__(self, opt, device='cuda'):
        self.opt, self.device = opt, torch.device(device)
        self.nerf = NeRF(opt, device=device)
        self.nerf_fine = NeRF(opt, is_fine_network=True, device=device) if opt.nerf.fine_sampling else None

    def render(self, opt, pose, H, W, intr, pixels=None, ray_idx=None, depth_range=None, iter=None, mode=None, rand=None):
        B = len(pose)
        if pixels is not None:
            center, ray = get_center_and_ray_at_pixels(pose, pixels, intr)
        else:
            center, ray = get_center_and_ray(pose, H, W, intr)
            if ray_idx is not None:
                if ray_idx.dim() == 2 and ray_idx.shape[0] == B:
                    bi = torch.arange(B, device=ray_idx.device)[:, None]
                    center, ray = center[bi, ray_idx.long()], ray[bi, ray_idx.long()]
                else:
                    center, ray = center[:, ray_idx], ray[:, ray_idx]
        if opt.camera.ndc:
            raise NotImplementedError('bg_nerf: NDC rays are not used by any PoseProbe configuration')
        N, S = ray.shape[1], opt.nerf.sample_intvs
        pred = Options(origins=center, viewdirs=ray)
        if rand is not None:
            jitter = rand[0].to(self.device) + torch.arange(S, device=self.device)[None, None, :, None].float()
            depth = jitter / S * (depth_range[1] - depth_range[0]) + depth_range[0]
            depth = {'metric': depth, 'inverse': 1 / (depth + 1e-8)}[opt.nerf.depth.param]
        else:
            depth = sample_depth(opt, B, N, S, depth_range, mode=mode, device=self.device)
        coarse = self.nerf.forward_samples(opt, center, ray, depth, mode=mode)
        coarse['t'] = depth
        pred.update(self.nerf.composite(opt, ray, coarse, depth))
        start = getattr(opt.nerf, 'ratio_start_fine_sampling_at_x', None)
        skip_fine = start is not None and iter is not None and iter < opt.max_iter * start
        if opt.nerf.fine_sampling and not skip_fine:
            with torch.no_grad():
                det = mode not in ('train', 'test-optim') or not opt.nerf.sample_stratified
                grid = rand[1].to(self.device) if (rand is not None and not det) else None
                fine_t = sample_depth_from_pdf(pred['weights'][..., 0].detach(), S, opt.nerf.sample_intvs_fine, depth_range,
                                               det=det, grid=grid)
            depth = torch.cat([depth, fine_t], dim=2).sort(dim=2).values
            fine = self.nerf_fine.forward_samples(opt, center, ray, depth, mode=mode)
            fine['t'] = depth
            fine = self.nerf_fine.composite(opt, ray, fine, depth)
            pred.update({k + '_fine': v for k, v in fine.items()})
        return pred


    def render_up_to_maxdepth(self, opt, pose, H, W, intr, depth_max, depth_min, pixels, iter=None, mode=None):
        """Rendering with a different far bound per ray (renderer.py:742-878, sample_depth_diff_max_range_per_ray :880-909):
        depth_max [B, N]; deterministic samples depth_min + (i + 1) / S * (depth_max - depth_min); the fine network, when
        active, is evaluated on the SAME samples (only the accumulated transmittance is wanted from this pass)."""
        center, ray = get_center_and_ray_at_pixels(pose, pixels, intr)
        S = opt.nerf.sample_intvs
        steps = (1.0 + torch.arange(S, device=self.device).float())[None, None, :, None] / S
        depth = steps * (depth_max[..., None, None] - depth_min) + depth_min
        pred = Options(origins=center, viewdirs=ray)
        coarse = self.nerf.forward_samples(opt, center, ray, depth, mode=mode)
        coarse['t'] = depth
        pred.update(self.nerf.composite(opt, ray, coarse, depth))
        start = getattr(opt.nerf, 'ratio_start_fine_sampling_at_x', None)
        skip_fine = start is not None and iter is not None and iter < opt.max_iter * start
        if opt.nerf.fine_sampling and not skip_fine:
            fine = self.nerf_fine.forward_samples(opt, center, ray, depth, mode=mode)
            fine['t'] = depth
            fine = self.nerf_fine.composite(opt, ray, fine, depth)
            pred.update({k + '_fine': v for k, v in fine.items()})
        return pred

    @torch.no_grad()
    def render_by_slices(self, opt, pose, H, W, intr, depth_range, iter=None, mode=None):
        """Full H x W images in slices of `opt.nerf.rand_rays` rays (renderer.py:629-663): per-ray outputs concatenated along
        the ray axis, `*_fine` keys when the fine pass is active."""
        keys = ['rgb', 'rgb_var', 'depth', 'depth_var', 'opacity', 'all_cumulated']
        parts = {}
        for c in range(0, H * W, opt.nerf.rand_rays):
            ray_idx = torch.arange(c, min(c + opt.nerf.rand_rays, H * W), device=self.device)
            ret = self.render(opt, pose, H, W, intr, ray_idx=ray_idx, depth_range=depth_range, iter=iter, mode=mode)
            for k in keys + [k + '_fine' for k in keys]:
                if k in ret:
                    parts.setdefault(k, []).append(ret[k])
        return Options({k: torch.cat(v, dim=1) for k, v in parts.items()})


def photometric_loss(rgb, image, huber=True):
    """Render loss of the branch (training/core/base_losses.py:151-156, :304-305)."""
    if huber:
        return torch.nn.functional.huber_loss(rgb, image, reduction='mean', delta=0.5) * 2.
    return torch.nn.functional.mse_loss(rgb, image)


class _NetState:
    """Gradient block and Adam moments of one packed network."""

    def __init__(self, net):
        self.net = net
        self.grad = torch.zeros_like(net.flat)
        self.m, self.v = torch.zeros_like(net.flat), torch.zeros_like(net.flat)
        self.seg_end = torch.tensor([net.flat.numel()], dtype=torch.int32, device=net.flat.device)
        self.steps = 0               # optimiser steps this network has taken (torch.optim.Adam keeps `step` per parameter)
        self.has_grad = False


class SceneEngine:
    """One optimisation step of the scene branch without autograd bookkeeping: forward, photometric loss, backward and one
    fused Adam update per packed parameter block (renderer.py:420-423 train_iteration + the optimiser step of
    lib/recon_scene.py:765).  With `net_fine` and `fine=True` the step is the reference's hierarchical one
    (renderer.py:586-611): the fine network runs on the union of the coarse samples and inverse-transform samples of the
    coarse weights, `loss = huber(rgb) + huber(rgb_fine)` (base_losses.py:304-307), both networks receive gradients.
    Ray gradients (sum over the passes) are returned for the caller's pose chain."""

    def __init__(self, net, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, net_fine=None):
        self.net, self.net_fine = net, net_fine
        self.lr, self.betas, self.eps = lr, betas, eps
        self.states = [_NetState(net)] + ([_NetState(net_fine)] if net_fine is not None else [])
        self.grad, self.m, self.v = self.states[0].grad, self.states[0].m, self.states[0].v
        self.step_count = 0
        self.seg_lr = torch.tensor([lr], dtype=torch.float32, device=net.flat.device)

    @staticmethod
    def _pass(state, center, ray, depth, image):
        """One network: forward, compositing, 2 * huber(delta = 0.5, mean), backward.  -> loss, g_center, g_ray, weights."""
        net = state.net
        R, S = depth.shape
        ws = net._workspace(R, S)
        f = dict(dtype=torch.float32, device=depth.device)
        if getattr(ws, 'step_bufs', None) is None:
            ws.step_bufs = dict(rgb_s=torch.empty(R * S, 3, **f), dens=torch.empty(R * S, **f), rgb=torch.empty(R, 3, **f),
                                d=torch.empty(R, **f), op=torch.empty(R, **f), w=torch.empty(R, S, **f), cum=torch.empty(R, **f),
                                rv=torch.empty(R, **f), dv=torch.empty(R, **f), g_rgb_s=torch.empty(R * S, 3, **f),
                                g_dens=torch.empty(R * S, **f), g_ray_c=torch.empty(R, 3, **f), g_center=torch.empty(R, 3, **f),
                                g_ray=torch.empty(R, 3, **f), zero_r=torch.zeros(R, **f), g_rgb=torch.empty(R, 3, **f),
                                loss=torch.zeros(256, **f), loss_i=-1)
        b = ws.step_bufs
        white = bool(net.opt.nerf.setbg_opaque or net.opt.mask_img)
        ops.nerf_fwd(net.flat, center, ray, depth, net.band_weights(), ws.count, R, S, ws.acts, b['rgb_s'], b['dens'], net.ctx)
        ops.nerf_composite_fwd(b['rgb_s'], b['dens'], depth, ray, R, S, white, b['rgb'], b['d'], b['op'], b['w'], b['cum'],
                               b['rv'], b['dv'])
        # 2 * huber(delta = 0.5, mean) and its gradient on the [R,3] colours in one launch; the loss lands in a ring slot (the
        # returned 0-dim tensor is rewritten 256 passes later: read it - .item() / copy - before that)
        b['loss_i'] = (b['loss_i'] + 1) % 256
        loss = b['loss'][b['loss_i']]
        ops.nerf_huber_loss(b['rgb'], image.contiguous(), 0.5, 2.0, loss, b['g_rgb'])
        ops.nerf_composite_bwd(b['rgb_s'], b['dens'], depth, ray, b['w'], R, S, white, b['g_rgb'], b['zero_r'],
                               b['zero_r'], None, b['g_rgb_s'], b['g_dens'], b['g_ray_c'])
        ops.nerf_bwd(net.flat, ray, depth, ws.count, R, S, ws.acts, b['rgb_s'], b['g_rgb_s'], b['g_dens'], ws.scratch,
                     state.grad, b['g_center'], b['g_ray'], net.ctx)
        state.has_grad = True
        return loss, b['g_center'], b['g_ray'] + b['g_ray_c'], b['w']

    def forward_backward(self, center, ray, depth, image, fine=False, depth_range=None, fine_grid=None):
        """center, ray [R,3]; depth [R,S]; image [R,3] -> (loss, g_center, g_ray); parameter gradients are accumulated into the
        networks' gradient blocks, which optimizer_step() consumes and re-zeroes.  fine=True adds the second pass
        (`depth_range` required; `fine_grid` [Nf + 1] replays the sampler's uniform draw)."""
        loss, g_center, g_ray, w = self._pass(self.states[0], center, ray, depth, image)
        if fine:
            if self.net_fine is None or depth_range is None:
                raise ValueError('SceneEngine: fine=True needs net_fine and depth_range')
            opt = self.net.opt
            S = depth.shape[1]
            det = not opt.nerf.sample_stratified
            fine_t = sample_depth_from_pdf(w[None], S, opt.nerf.sample_intvs_fine, depth_range, det=det, grid=fine_grid)
            depth_f = torch.cat([depth, fine_t[0, :, :, 0]], dim=1).sort(dim=1).values.contiguous()
            loss_f, gc_f, gr_f, _ = self._pass(self.states[1], center, ray, depth_f, image)
            loss, g_center, g_ray = loss + loss_f, g_center + gc_f, g_ray + gr_f
        return loss, g_center, g_ray

    def optimizer_step(self, grad_scale=1.0):
        """torch.optim.Adam semantics (lib/utils.py:294-299); also re-zeroes the gradient blocks for the next step.  Like
        torch's optimiser, a network that received no gradient this iteration (the fine one before its start) is skipped and
        its step counter - hence its bias correction - does not advance."""
        self.step_count += 1
        for st in self.states:
            if not st.has_grad:
                continue
            st.steps += 1
            st.has_grad = False
            ops.adam_flat(st.net.flat, st.grad, st.m, st.v, st.seg_end, self.seg_lr, grad_scale, self.betas[0], self.betas[1],
                          self.eps, st.steps, True)

    def set_lr(self, lr):
        self.lr = lr
        self.seg_lr.fill_(lr)

    def step(self, center, ray, depth, image, **kw):
        loss, g_center, g_ray = self.forward_backward(center, ray, depth, image, **kw)
        self.optimizer_step()
        return loss, g_center, g_ray
