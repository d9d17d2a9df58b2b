"""Host-side orchestration of the HIP hot path: workspace, forward chain, hand-derived backward chain and
the fused train step (ray select -> render -> loss -> backward -> optimiser), with zero host syncs.

Mirrors one iteration of the object branch in recon_scene.optimize_increamental
(lib/recon_scene.py:572-606, :648-649, :742-747, :765-771) and Voxurf.forward (lib/voxurf_coarse.py:922-1092).
torch is used for device memory and streams only; every arithmetic step is a kernel of libposeprobe_hip.so.
"""
import math
from dataclasses import dataclass

import numpy as np
import torch

from . import ops

PAD = 64


def _pad(n):
    return (n + PAD - 1) // PAD * PAD


@dataclass
class SceneConfig:
    xyz_min: np.ndarray
    xyz_max: np.ndarray
    num_voxels: int
    stepsize: float = 1.5
    near: float = 0.24
    far: float = 4.8
    bg: float = 0.0
    N_iters: int = 10000
    s_ratio: float = 50.0
    s_start: float = 0.2
    step_start: float = 0.0
    barf_c2f: tuple = (0.6, 1.0)
    posbase_pe: int = 5
    viewbase_pe: int = 1
    k0_dim: int = 12
    out_range: float = 1.0
    inverse_y: bool = True
    sdf_index_exact: bool = False    # True: exact voxel index in the custom SDF sampler above 2^24 voxels (default: the reference's fp32 index)

    def __post_init__(self):
        # identical fp32 arithmetic to Voxurf._set_grid_resolution (voxurf_coarse.py:319-323)
        lo = torch.tensor(np.asarray(self.xyz_min, dtype=np.float32))
        hi = torch.tensor(np.asarray(self.xyz_max, dtype=np.float32))
        vs = ((hi - lo).prod() / self.num_voxels).pow(1 / 3)
        self.voxel_size = float(vs)
        self.world_size = [int(v) for v in ((hi - lo) / vs).long().tolist()]
        self.pp = ops.make_scene(lo.tolist(), hi.tolist(), self.world_size, self.voxel_size, self.stepsize, self.near,
                                 self.far, self.bg, self.out_range, self.k0_dim, self.posbase_pe, self.viewbase_pe,
                                 sdf_index_exact=True if self.sdf_index_exact else None)
        self.n_samples = self.pp.n_samples

    def s_val(self, global_step):
        return 1. / (global_step + self.s_ratio / self.s_start - self.step_start) * self.s_ratio

    def pe_weights(self, progress):
        """BARF c2f weights for xyz (L=posbase_pe) then view (L=viewbase_pe); voxurf_coarse.py:721-732 (fp32 ops)."""
        out = []
        for L in (self.posbase_pe, self.viewbase_pe):
            if self.barf_c2f is None:
                out.append(np.ones(L, dtype=np.float32))
                continue
            start, end = self.barf_c2f
            alpha = (torch.tensor(progress, dtype=torch.float32) - start) / (end - start) * L
            k = torch.arange(L, dtype=torch.float32)
            w = (1 - (alpha - k).clamp_(min=0, max=1).mul_(np.pi).cos_()) / 2
            out.append(w.numpy())
        return np.concatenate(out).astype(np.float32)


def dynamic_weight(w0, w1, it, total):
    """lib/losses.py:30-32"""
    return w0 * math.exp(math.log(w1 / w0) / total * it)


class Workspace:
    """All per-ray / per-sample device buffers for N rays and `capacity` samples (default N*S)."""

    def __init__(self, N, capacity, device, sample_capacity=None, backward=True, keep_activations=True, ctx=None):
        f = dict(dtype=torch.float32, device=device)
        i = dict(dtype=torch.int32, device=device)
        self.N, self.cap = N, capacity
        self.sample_cap = sc_ = sample_capacity or capacity
        e = torch.empty
        # rays
        self.rays_o, self.rays_d, self.viewdirs = e(N, 3, **f), e(N, 3, **f), e(N, 3, **f)
        self.target, self.mask_px = e(N, 3, **f), e(N, **f)
        self.t_min, self.t_max = e(N, **f), e(N, **f)
        self.ray_start = torch.zeros(N + 1, **i)
        self.count = torch.zeros(1, **i)
        # samples
        self.pts, self.ray_id, self.step_k, self.step = e(sc_, 3, **f), e(sc_, **i), e(sc_, **i), e(sc_, **f)
        wsz = ops.mlp_workspaces(capacity)                 # sizes come from the library (pp_rgbnet_workspace / pp_warp_workspace)
        assert wsz['warp'][0] == 4 * capacity * 4 * 128 and wsz['rgbnet'][0] == 3 * capacity * 128
        # keep_activations=False (inference: nobody will differentiate this pass): the MLP activations are not kept when the forward kernels can run without them
        # (pp_warp_fwd / pp_rgbnet_fwd with acts = NULL: the split-precision kernels, the default): 2.5 KB per sample less to write
        from . import _lib
        octx = _lib.default_context() if ctx is None else ctx          # the context the forward kernels will be called with
        split = octx.get('mlp_split') if octx.get('mlp_fused') else 0
        self.warp_acts = e(4, capacity * 4, 128, **f) if (backward or keep_activations or not split & 1) else None
        self.warp_out = e(capacity, 16, **f)
        self.alpha, self.gradient = e(capacity, **f), e(capacity, 3, **f)
        self.sdf_final, self.sdf_deform, self.grad_deform = e(capacity, **f), e(capacity, **f), e(capacity, 9, **f)
        self.feat = e(capacity, ops.FEAT_LD, **f)
        self.rgb_acts = e(3, capacity, 128, **f) if (backward or keep_activations or not split & 4) else None
        self.rgb = e(capacity, 3, **f)
        self.weights, self.T = e(capacity, **f), e(capacity, **f)
        # ray outputs
        self.alphainv_last, self.i_end = e(N, **f), e(N, **i)
        self.rgb_marched, self.rgb_pre = e(N, 3, **f), e(N, 3, **f)
        self.cum_weights, self.depth_acc = e(N, **f), e(N, **f)
        self.zero_block = torch.zeros(16, **f)          # loss_out[8] | tv_out[1] | mask_sum[1] : one memset per step
        self.loss_out, self.tv_out, self.mask_sum = self.zero_block[:8], self.zero_block[8:9], self.zero_block[9:10]
        if not backward:
            return
        # backward buffers
        self.g_rgbm, self.g_last, self.g_cw = e(N, 3, **f), e(N, **f), e(N, **f)
        self.g_alpha, self.g_rgb = e(capacity, **f), e(capacity, 3, **f)
        self.g_feat = e(capacity, ops.FEAT_LD, **f)
        self.g_gradient, self.g_pts, self.g_view_s = e(capacity, 3, **f), e(capacity, 3, **f), e(capacity, 3, **f)
        self.g_grad_deform, self.g_corr, self.g_sdf_deform = e(capacity, 9, **f), e(capacity, **f), e(capacity, **f)
        self.g_warp_out = e(capacity, 16, **f)
        self.scratch = e(wsz['warp'][1], **f)              # Ybar of the warp chain (+ transposed weights, layered path)
        self.scratch_rgb = e(wsz['rgbnet'][1], **f)   # Ybar of rgbnet: its weight-gradient kernel may still be
        #                                                         reading it on the side stream while the warp chain runs
        self.g_rays_o, self.g_rays_d, self.g_viewdirs = e(N, 3, **f), e(N, 3, **f), e(N, 3, **f)


class FlatParams:
    """Packed small parameters: [sdf_ab | rgbnet | warp], each segment padded to 64 floats (16-B aligned sub-blocks)."""
    SEG = [('sdf_ab', 2), ('rgbnet', ops.RGBNET_PARAMS), ('warp', ops.WARP_PARAMS)]

    def __init__(self, device, moments=True):
        """moments=False: parameters + gradients only (the drop-in autograd node, whose optimiser lives outside)."""
        self.off, o = {}, 0
        for name, n in self.SEG:
            self.off[name] = (o, n)
            o += _pad(n)
        self.n = o
        z = lambda: torch.zeros(o, dtype=torch.float32, device=device)
        self.data, self.grad = z(), z()
        self.m, self.v = (z(), z()) if moments else (None, None)

    def view(self, name, which='data'):
        o, n = self.off[name]
        return getattr(self, which)[o:o + n]

    # ---- conversion from / to the reference's state_dict tensors -------------------------------------------
    def load_reference(self, sdf_alpha, sdf_beta, rgbnet, warp):
        """rgbnet: 4 x (W[out,in], b) ; warp: 5 x (W, b) in the reference layouts."""
        with torch.no_grad():
            self.view('sdf_ab').copy_(torch.cat([sdf_alpha.reshape(1), sdf_beta.reshape(1)]))
            self.view('rgbnet').copy_(pack_rgbnet(rgbnet).to(self.data.device))
            self.view('warp').copy_(pack_warp(warp).to(self.data.device))

    def export_grads(self):
        g = {'sdf_alpha': self.view('sdf_ab', 'grad')[0:1].clone(), 'sdf_beta': self.view('sdf_ab', 'grad')[1:2].clone()}
        g['rgbnet'] = unpack_rgbnet(self.view('rgbnet', 'grad'))
        g['warp'] = unpack_warp(self.view('warp', 'grad'))
        return g


def pack_rgbnet(layers, in_dim=57):
    W0, b0 = layers[0]
    W0p = torch.zeros(128, 64, dtype=torch.float32, device=W0.device)
    W0p[:, :W0.shape[1]] = W0
    parts = [W0p.reshape(-1), b0]
    for W, b in layers[1:]:
        parts += [W.reshape(-1), b]
    return torch.cat([p.detach().float() for p in parts])


def unpack_rgbnet(flat, in_dim=57):
    o, out = 0, []
    W0 = flat[o:o + 128 * 64].reshape(128, 64)[:, :in_dim]; o += 128 * 64
    b0 = flat[o:o + 128]; o += 128
    out.append((W0, b0))
    for shp in ((128, 128), (128, 128), (3, 128)):
        n = shp[0] * shp[1]
        W = flat[o:o + n].reshape(shp); o += n
        b = flat[o:o + shp[0]]; o += shp[0]
        out.append((W, b))
    return out


def pack_warp(layers):
    parts = []
    for W, b in layers:
        parts += [W.reshape(-1), b]
    return torch.cat([p.detach().float() for p in parts])


def unpack_warp(flat):
    o, out = 0, []
    for shp in ((128, 3), (128, 128), (128, 128), (128, 128), (4, 128)):
        n = shp[0] * shp[1]
        W = flat[o:o + n].reshape(shp); o += n
        b = flat[o:o + shp[0]]; o += shp[0]
        out.append((W, b))
    return out


class RenderCore:
    """Forward and backward kernel chains over a Workspace.  Parameters are passed as raw device tensors:
    k0_cl [X,Y,Z,C] (channels-last), sdf [X,Y,Z], flat (FlatParams-like with .view)."""

    def __init__(self, cfg: SceneConfig, ctx=None):
        """ctx: the pp_context (ops.Context) every option-dependent kernel of this core is called with; None = the host's
        default context.  Its option `side_stream` (1 / 2) runs the weight-gradient kernel of each MLP chain on the context's
        auxiliary stream beside the small, latency-bound kernels that follow the chain's data-gradient kernel and joins it before
        the next register-hungry MLP kernel.  Measured on MI355X (kernel trace): the overlap happens, but the small kernels only
        get the leftover wave slots (15 -> 45 us each) and the fork / join edges cost ~10 us per chain, so the step is 1 % SLOWER:
        off by default (PP_SIDE_STREAM=1 in the host's environment turns it on in the default context)."""
        self.cfg = cfg
        self.ctx = ctx

    # -- forward -------------------------------------------------------------------------------------------
    def sample(self, ws, jitter):
        ops.sample_dense(self.cfg.pp, ws.rays_o, ws.rays_d, jitter, ws.sample_cap, ws.t_min, ws.t_max, ws.ray_start, ws.count,
                         ws.pts, ws.ray_id, ws.step_k, ws.step)

    def forward(self, ws, k0_cl, sdf, sdf_ab, rgbnet_p, warp_p, inv_s, pe_w, step_w=None, before_k0_use=None):
        cfg, sc = self.cfg, self.cfg.pp
        ops.warp_fwd(warp_p, ws.pts, ws.count, ws.cap, cfg.out_range, ws.warp_acts, ws.warp_out, self.ctx)
        ops.geometry_fwd(sc, sdf, sdf_ab, ws.pts, ws.warp_out, ws.viewdirs, ws.ray_id, ws.count, ws.cap, inv_s,
                         ws.alpha, ws.gradient, ws.sdf_final, ws.sdf_deform, ws.grad_deform)
        if before_k0_use is not None:
            before_k0_use()             # multi-GPU: the all-gather of the updated grid overlaps everything above
        ops.color_feat_fwd(sc, k0_cl, ws.pts, ws.viewdirs, ws.ray_id, ws.gradient, pe_w, ws.count, ws.cap, ws.feat)
        ops.rgbnet_fwd(rgbnet_p, ws.feat, ws.count, ws.cap, ws.rgb_acts, ws.rgb, self.ctx)
        ops.march_fwd(ws.alpha, ws.rgb, ws.step if step_w is None else step_w, None, ws.ray_start, ws.N, cfg.bg,
                      ws.weights, ws.T, ws.alphainv_last, ws.i_end, ws.rgb_marched, ws.rgb_pre, ws.cum_weights,
                      ws.depth_acc, None)

    # -- backward ------------------------------------------------------------------------------------------
    def backward(self, ws, k0_cl, sdf, sdf_ab, rgbnet_p, warp_p, inv_s, pe_w, k0_grad_cl, sdf_ab_grad, rgbnet_grad,
                 warp_grad, g_depth=None, g_weights=None, g_gradient_ext=None, g_sdf_deform=None, g_grad_deform=None,
                 g_correction=None, g_alpha_ext=None, g_rgb_ext=None, after_k0_grad=None, defer_join=False, priors=None):
        """Consumes ws.g_rgbm / ws.g_last / ws.g_cw (+ optional per-sample upstream grads), accumulates parameter
        grads (atomic +=) and leaves d/d ray_pts in ws.g_pts and the per-sample viewdir grads in ws.g_view_s."""
        cfg, sc = self.cfg, self.cfg.pp
        ops.march_bwd(ws.alpha, ws.rgb, ws.step, ws.weights, ws.T, ws.alphainv_last, ws.ray_start, ws.i_end, ws.N, cfg.bg,
                      ws.rgb_pre, ws.g_rgbm, ws.g_cw, ws.g_last, g_depth, g_weights, ws.g_alpha, ws.g_rgb)
        if g_alpha_ext is not None:
            ws.g_alpha.add_(g_alpha_ext)
        if g_rgb_ext is not None:
            ws.g_rgb.add_(g_rgb_ext)
        ctx = self.ctx
        ops.rgbnet_bwd(rgbnet_p, ws.feat, ws.rgb_acts, ws.rgb, ws.g_rgb, ws.count, ws.cap, ws.scratch_rgb,
                       rgbnet_grad, ws.g_feat, ctx)
        ops.color_feat_bwd(sc, k0_cl, ws.pts, ws.viewdirs, ws.ray_id, ws.gradient, pe_w, ws.count, ws.cap, ws.g_feat,
                           k0_grad_cl, ws.g_pts, ws.g_gradient, ws.g_view_s)
        if after_k0_grad is not None:
            after_k0_grad()             # multi-GPU: the grid reduce-scatter overlaps the rest of the backward
        if priors is not None:
            # fused step: the sample-level priors (eikonal, deformation) are differentiated inside the geometry backward
            w_eik, w_dyn, ls, loss_out, batch_norm = priors
            ops.geometry_bwd_priors(sc, sdf, sdf_ab, ws.pts, ws.warp_out, ws.viewdirs, ws.ray_id, ws.count, ws.cap, inv_s,
                                    ws.g_alpha, ws.g_gradient, w_eik, w_dyn, ls, 1, ws.g_warp_out, ws.g_pts, ws.g_view_s,
                                    sdf_ab_grad, loss_out, batch_norm)
        else:
            if g_gradient_ext is not None:
                g_gradient_ext(ws)      # callable adding loss terms into ws.g_gradient etc.
            ops.geometry_bwd(sc, sdf, sdf_ab, ws.pts, ws.warp_out, ws.viewdirs, ws.ray_id, ws.count, ws.cap, inv_s,
                             ws.g_alpha, ws.g_gradient, None, g_sdf_deform, g_grad_deform, g_correction, 1, ws.g_warp_out,
                             ws.g_pts, ws.g_view_s, sdf_ab_grad)
        ops.context_join(ctx)           # rgbnet's weight-gradient kernel is done before the next register-hungry kernel
        ops.warp_bwd(warp_p, ws.pts, ws.warp_acts, ws.g_warp_out, ws.count, ws.cap, cfg.out_range, ws.scratch, warp_grad,
                     ws.g_pts, ctx)
        if not defer_join:
            ops.context_join(ctx)
        return ctx


class TrainEngine:
    """Fused object-branch train step on one GPU (rank-local shard of the rays when world_size > 1)."""

    LR = {'k0': 1e-1, 'rgbnet': 1e-3, 'warp': 1e-3, 'sdf_ab': 1e-2}     # configs/dtu_e2e/scan1.py:87-103

    def __init__(self, cfg: SceneConfig, n_views, H, W, n_rand, device='cuda', lr_pose=1e-3, lr_pose_end=1e-4,
                 pose_iters=1, lrate_decay=10, loss_scale=0.1, weight_main=1.0, weight_tv_k0=0.01, weight_mask=0.1,
                 fix_first=True, capacity=None, x_slab=None, dist_ctx=None, deterministic_scatter=False, options=None):
        """deterministic_scatter: the k0 gradient is accumulated per voxel in sample order (sorted scatter, ~0.2 ms instead of
        0.04 ms per step) instead of by float atomics - bit-identical gradient grids for identical inputs, and bit-identical
        replicas in the multi-GPU "samples" mode without the periodic re-broadcast.
        options: {name: value} of a PRIVATE pp_context for this engine's kernels (e.g. {'mlp_split': 0}: every MLP product on the
        fp32 MFMA instructions); None = the host's default context.  Engines with different options coexist in one process."""
        self.cfg, self.dev = cfg, torch.device(device)
        self.ctx = ops.Context(**options) if options else None
        self.deterministic_scatter = bool(deterministic_scatter)
        self._scatter_work = None
        self.V, self.H, self.W, self.N = n_views, H, W, n_rand
        cap = capacity or n_rand * cfg.n_samples
        self.ws = Workspace(n_rand, cap, self.dev, ctx=self.ctx)
        self.core = RenderCore(cfg, ctx=self.ctx)
        X, Y, Z = cfg.world_size
        f = dict(dtype=torch.float32, device=self.dev)
        self.k0 = [torch.zeros(X, Y, Z, cfg.k0_dim, **f), torch.zeros(X, Y, Z, cfg.k0_dim, **f)]   # ping-pong
        self.k0_cur = 0
        self.k0_grad, self.k0_m, self.k0_v = (torch.zeros(X, Y, Z, cfg.k0_dim, **f) for _ in range(3))
        self.sdf = torch.zeros(X, Y, Z, **f)
        # one byte per voxel, two parities: the scatter of step n marks [n & 1], the fused optimiser pass reads it (voxels that
        # were not reached keep a known-zero gradient: no read, no re-zeroing) and clears the other one for step n+1
        self.k0_touched = torch.zeros(2, X * Y * Z, dtype=torch.uint8, device=self.dev)
        self.touch_par = 0
        self._k0_marked = False
        self.flat = FlatParams(self.dev)
        self.se3 = torch.zeros(n_views, 6, **f)
        self.se3_grad, self.se3_m, self.se3_v = (torch.zeros(n_views, 6, **f) for _ in range(3))
        self.w2c_init = torch.zeros(n_views, 3, 4, **f)
        self.w2c, self.c2w = torch.zeros(n_views, 3, 4, **f), torch.zeros(n_views, 3, 4, **f)
        self.jac = torch.zeros(n_views, 12, 6, **f)
        self.c2w_grad = torch.zeros(n_views, 3, 4, **f)
        mask = torch.ones(n_views, dtype=torch.int32)
        if fix_first:
            mask[0] = 0                     # get_current_pose_pnp never refines view 0 (recon_scene.py:68)
        self.refine_mask = mask.to(self.dev)
        self.intr = torch.zeros(n_views, 4, **f)
        self.images = self.masks = None
        self.lr = dict(self.LR)
        self.lr_pose = lr_pose
        self.pose_gamma = (lr_pose_end / (1e-10 + lr_pose)) ** (1. / pose_iters)
        self.decay = 0.1 ** (1 / (lrate_decay * 1000))
        self.loss_scale, self.w_main, self.w_tv, self.w_mask = loss_scale, weight_main, weight_tv_k0, weight_mask
        self.n_step = 0
        self.x_slab = x_slab or (0, X)
        self.dist = dist_ctx
        segs = [self.flat.off[n][0] + _pad(self.flat.off[n][1]) for n, _ in FlatParams.SEG]
        self.seg_end = torch.tensor(segs, dtype=torch.int32, device=self.dev)
        self.pose_seg_end = torch.tensor([n_views * 6], dtype=torch.int32, device=self.dev)
        # per-step scalars travel in ONE async H2D copy from a pinned staging buffer:
        #   [0:npe) BARF weights | [npe:npe+3) lr of sdf_ab, rgbnet, warp | [npe+3] pose lr
        self.npe = cfg.posbase_pe + cfg.viewbase_pe
        # ring of pinned slots: the host may run many steps ahead of the GPU (no sync in the step), a slot must not be
        # rewritten before its async copy has executed; the launch queue never holds 256 steps
        self.step_host = torch.zeros(256, self.npe + 4, dtype=torch.float32)
        if self.dev.type == 'cuda':
            self.step_host = self.step_host.pin_memory()
        self._slot = 0
        self.step_dev = torch.zeros(self.npe + 4, **f)
        self.pe_w = self.step_dev[:self.npe]
        self.seg_lr = self.step_dev[self.npe:self.npe + 3]
        self.pose_seg_lr = self.step_dev[self.npe + 3:self.npe + 4]

    # ---- data / parameter loading ---------------------------------------------------------------------------
    def set_views(self, images, masks, Ks, w2c_init):
        """images [V,H,W,3], masks [V,H,W,1|none], Ks [V,3,3], w2c_init [V,3,4] (numpy or torch)."""
        t = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float32).to(self.dev).contiguous()
        self.images, self.masks = t(images), t(masks).reshape(self.V, self.H, self.W).contiguous()
        K = torch.as_tensor(np.asarray(Ks), dtype=torch.float32)
        self.intr.copy_(torch.stack([K[:, 0, 0], K[:, 1, 1], K[:, 0, 2], K[:, 1, 2]], -1))
        self.w2c_init.copy_(t(w2c_init))

    @property
    def k0_cl(self):
        return self.k0[self.k0_cur]

    def load_reference_params(self, k0, sdf, sdf_alpha, sdf_beta, rgbnet, warp, se3=None):
        """k0 [1,C,X,Y,Z], sdf [1,1,X,Y,Z] in the reference layout (state_dict tensors)."""
        with torch.no_grad():
            self.k0_cl.copy_(k0[0].permute(1, 2, 3, 0).to(self.dev))
            self.sdf.copy_(sdf[0, 0].to(self.dev))
            d = lambda x: x.detach().to(self.dev)
            self.flat.load_reference(d(sdf_alpha), d(sdf_beta), [(d(W), d(b)) for W, b in rgbnet],
                                     [(d(W), d(b)) for W, b in warp])
            if se3 is not None:
                self.se3.copy_(se3.to(self.dev))

    def k0_reference_layout(self, t=None):
        t = self.k0_cl if t is None else t
        return t.permute(3, 0, 1, 2)[None]

    # ---- one step -------------------------------------------------------------------------------------------
    def scatter_work(self, n_samples):
        """Workspace of the deterministic scatter (grown on demand, kept)."""
        need = ops.k0_scatter_sorted_workspace(n_samples)
        if self._scatter_work is None or self._scatter_work.numel() < need:
            self._scatter_work = torch.empty(need, dtype=torch.uint8, device=self.dev)
        return self._scatter_work

    def zero_grads(self):
        self.k0_grad.zero_()
        self.k0_touched.zero_()
        self._k0_marked = False
        self.flat.grad.zero_()
        self.se3_grad.zero_()

    def render_and_grads(self, ray_idx, jitter, global_step):
        """pose -> rays -> forward -> losses -> full backward.  Gradients are accumulated into k0_grad / flat.grad /
        se3_grad (which must be zero on entry)."""
        cfg, ws, sc = self.cfg, self.ws, self.cfg.pp
        ops.pose_fwd(self.se3, self.w2c_init, self.refine_mask, self.w2c, self.c2w, self.jac)
        ops.raygen_select_fwd(sc, ray_idx, self.c2w, self.intr, self.H, self.W, cfg.inverse_y, True, self.images, self.masks,
                              ws.rays_o, ws.rays_d, ws.viewdirs, ws.target, ws.mask_px)
        self.core.sample(ws, jitter)
        if self.dist is not None:
            # (sample count, masked-pixel count) of every rank: sizes the sample exchange exactly and normalises the losses
            # over the union batch; the 8-byte all-gather completes under the forward pass
            self.dist.start_batch_stats(ws.count, ws.mask_px)
        progress = global_step / cfg.N_iters
        self._upload_step_scalars(progress)
        s_val = cfg.s_val(global_step)
        inv_s = float(np.float32(1.0) / np.float32(s_val))
        P = self.flat
        self.core.forward(ws, self.k0_cl, self.sdf, P.view('sdf_ab'), P.view('rgbnet'), P.view('warp'), inv_s, self.pe_w,
                          before_k0_use=None if self.dist is None else (lambda: self.dist.wait_parameters(self)))
        ws.zero_block.zero_()
        w_dyn = dynamic_weight(1e-1, 1e-3, global_step, cfg.N_iters)
        ls = self.loss_scale
        batch_norm = None if self.dist is None else self.dist.wait_batch_stats()
        ops.loss_rays(ws.rgb_marched, ws.alphainv_last, ws.cum_weights, ws.target, ws.mask_px, ws.mask_sum, self.w_main,
                      0.01, self.w_mask, ls, ws.g_rgbm, ws.g_last, ws.g_cw, ws.loss_out, batch_norm)

        # The k0 scatter is issued from here (not inside colour-feature backward) so that it can mark the voxels it reaches:
        # single GPU = right after the colour-feature backward; multi-GPU "samples" mode = replayed for all ranks' gathered
        # samples at the end of the backward; "zero1" = dense reduce-scatter, no marking (every voxel may be non-zero).
        dense_exchange = self.dist is not None and self.dist.local_scatter
        k0_grad = self.k0_grad if dense_exchange else None
        self._k0_marked = not dense_exchange
        touched = self.k0_touched[self.touch_par]

        def after_k0():
            if self.dist is None:
                if self.deterministic_scatter:
                    ops.k0_scatter_samples_sorted(sc, ws.pts, ws.count, ws.cap, ws.g_feat, self.k0_grad, self.scatter_work(ws.cap), touched)
                else:
                    ops.k0_scatter_samples(sc, ws.pts, ws.count, ws.cap, ws.g_feat, self.k0_grad, touched)
            else:
                self.dist.start_grid_reduce(self)
        self.core.backward(ws, self.k0_cl, self.sdf, P.view('sdf_ab'), P.view('rgbnet'), P.view('warp'), inv_s, self.pe_w,
                           k0_grad, P.view('sdf_ab', 'grad'), P.view('rgbnet', 'grad'), P.view('warp', 'grad'),
                           priors=(1.0, w_dyn, ls, ws.loss_out, batch_norm),
                           after_k0_grad=after_k0, defer_join=True)
        ctx = self.ctx
        ops.raygen_select_bwd(sc, ray_idx, self.c2w, self.intr, self.H, self.W, cfg.inverse_y, ws.rays_o, ws.rays_d,
                              ws.t_min, ws.ray_start, ws.g_pts, ws.step, ws.g_view_s, None, None, None, None, None, None,
                              None, self.c2w_grad)
        ops.pose_bwd(self.jac, self.c2w_grad, self.se3_grad)
        ops.context_join(ctx)           # the warp net's weight gradients (side stream) before anything reads flat.grad
        return s_val, w_dyn

    def _upload_step_scalars(self, progress):
        """BARF weights for this step and the learning rates the optimiser will use at the END of this step (the
        per-step exponential decay precedes the optimiser step, recon_scene.py:742-768)."""
        h = self.step_host[self._slot]
        self._slot = (self._slot + 1) % self.step_host.shape[0]
        h[:self.npe] = torch.from_numpy(self.cfg.pe_weights(progress))
        d = self.decay
        h[self.npe + 0], h[self.npe + 1], h[self.npe + 2] = self.lr['sdf_ab'] * d, self.lr['rgbnet'] * d, self.lr['warp'] * d
        h[self.npe + 3] = self.lr_pose
        self.step_dev.copy_(h, non_blocking=True)

    def optimizer_step(self, optimize_pose=True, grad_scale=1.0):
        cfg = self.cfg
        self.n_step += 1
        for k in self.lr:                       # per-step exponential decay precedes the step (recon_scene.py:742-768)
            self.lr[k] *= self.decay
        self._k0_step(self.lr['k0'], self.n_step, grad_scale)
        if self.dist is not None:
            self.dist.wait_small()          # the small all-reduce (MLP / alpha-beta / pose gradients) ran beside the grid pass
        ops.adam_flat(self.flat.data, self.flat.grad, self.flat.m, self.flat.v, self.seg_end, self.seg_lr, grad_scale, 0.9,
                      0.99, 1e-8, self.n_step, 1)
        if optimize_pose:
            ops.adam_flat(self.se3.view(-1), self.se3_grad.view(-1), self.se3_m.view(-1), self.se3_v.view(-1),
                          self.pose_seg_end, self.pose_seg_lr, grad_scale, 0.9, 0.999, 1e-8, self.n_step, 1)
            self.lr_pose *= self.pose_gamma

    def _k0_step(self, lr_k0, n_step, grad_scale):
        """Fused TV + Adam pass over the colour grid (ping-pong buffers flip)."""
        cfg = self.cfg
        X, Y, Z = cfg.world_size
        tv_scale = self.loss_scale * self.w_tv / (3.0 * X * Y * Z * cfg.k0_dim)
        src, dst = self.k0[self.k0_cur], self.k0[1 - self.k0_cur]
        xb, xe = self.x_slab
        if self._k0_marked and (xb, xe) == (0, X):
            cur = self.touch_par
            ops.grid_tv_adam_step_sparse(src, dst, self.k0_grad, self.k0_m, self.k0_v, cfg.world_size, cfg.k0_dim, xb, xe,
                                         tv_scale, grad_scale, lr_k0, 0.9, 0.99, 1e-8, n_step, self.ws.tv_out,
                                         self.k0_touched[cur], self.k0_touched[1 - cur], self.ctx)
            self.touch_par = 1 - cur
            self._k0_marked = False
        else:
            ops.grid_tv_adam_step(src, dst, self.k0_grad, self.k0_m, self.k0_v, cfg.world_size, cfg.k0_dim, xb, xe, tv_scale,
                                  grad_scale, lr_k0, 0.9, 0.99, 1e-8, n_step, self.ws.tv_out, self.ctx)
        self.k0_cur = 1 - self.k0_cur

    def train_step(self, ray_idx, jitter, global_step, optimize_pose=True):
        """Gradients are zeroed by the optimiser kernels themselves after use; call zero_grads() once before the
        first step."""
        out = self.render_and_grads(ray_idx, jitter, global_step)
        self.grad_scale = 1.0
        if self.dist is not None:
            self.dist.reduce_gradients(self)          # sets x_slab and grad_scale = 1/world
        self.optimizer_step(optimize_pose, grad_scale=self.grad_scale)
        if self.dist is not None:
            self.dist.gather_parameters(self)
        return out

    # ---- checkpoint / resume (wire format of recon_scene.save_checkpoints, lib/recon_scene.py:779-790) -------------
    RGBNET_KEYS = ('rgbnet.0', 'rgbnet.2.0', 'rgbnet.3.0', 'rgbnet.4')

    def model_state_dict(self):
        """The trainable state under the reference's state_dict names and logical shapes (lib/voxurf_coarse.py module tree,
        SURVEY 8b), so that `Voxurf.load_state_dict(..., strict=False)` of either implementation accepts it."""
        c = lambda t: t.detach().clone().cpu()
        cfg, P = self.cfg, self.flat
        sd = {'sdf_alpha': c(P.view('sdf_ab')[0:1]), 'sdf_beta': c(P.view('sdf_ab')[1:2]),
              'xyz_min': torch.tensor(np.asarray(cfg.xyz_min, dtype=np.float32)),
              'xyz_max': torch.tensor(np.asarray(cfg.xyz_max, dtype=np.float32)),
              'sdf.grid': c(self.sdf)[None, None], 'k0.grid': c(self.k0_reference_layout()).contiguous()}
        for name in ('sdf', 'k0'):
            sd[name + '.xyz_min'], sd[name + '.xyz_max'] = sd['xyz_min'].clone(), sd['xyz_max'].clone()
        for key, (W, b) in zip(self.RGBNET_KEYS, unpack_rgbnet(P.view('rgbnet'))):
            sd[key + '.weight'], sd[key + '.bias'] = c(W).contiguous(), c(b)
        for i, (W, b) in enumerate(unpack_warp(P.view('warp'))):
            sd[f'warp_network.deform_net.net.net.{i}.0.weight'] = c(W)
            sd[f'warp_network.deform_net.net.net.{i}.0.bias'] = c(b)
        return sd

    def voxurf_view(self, model=None):
        """The drop-in `Voxurf` module over this engine's CURRENT parameters, for the callers either side of the train step
        that want the module API (surface queries, reprojection losses, inference): built once from `model_kwargs()`; every
        call re-points `k0.grid` at the live ping-pong buffer (zero-copy, channels-last) and the SDF template at the engine's,
        and copies the small parameters (alpha / beta, both MLPs: 370 KB) on the device.  Nothing goes through the host."""
        from . import voxurf_coarse
        if model is None:
            model = voxurf_coarse.Voxurf(**self.model_kwargs()).to(self.dev)
        P = self.flat
        with torch.no_grad():
            model.k0.grid.data = self.k0_reference_layout()
            model.sdf.grid.data = self.sdf[None, None]
            model.sdf_alpha.data.copy_(P.view('sdf_ab')[0:1])
            model.sdf_beta.data.copy_(P.view('sdf_ab')[1:2])
            sd = dict(model.named_parameters())
            for key, (W, b) in zip(self.RGBNET_KEYS, unpack_rgbnet(P.view('rgbnet'))):
                sd[key + '.weight'].data.copy_(W)
                sd[key + '.bias'].data.copy_(b)
            for i, (W, b) in enumerate(unpack_warp(P.view('warp'))):
                sd[f'warp_network.deform_net.net.net.{i}.0.weight'].data.copy_(W)
                sd[f'warp_network.deform_net.net.net.{i}.0.bias'].data.copy_(b)
        return model

    def load_model_state_dict(self, sd):
        rg = [(sd[k + '.weight'], sd[k + '.bias']) for k in self.RGBNET_KEYS]
        wp = [(sd[f'warp_network.deform_net.net.net.{i}.0.weight'], sd[f'warp_network.deform_net.net.net.{i}.0.bias'])
              for i in range(5)]
        self.load_reference_params(sd['k0.grid'], sd['sdf.grid'], sd['sdf_alpha'], sd['sdf_beta'], rg, wp)

    # ---- optimiser state in the reference's layout ---------------------------------------------------------------------
    # lib/recon_scene.py:779-791 stores `self.optimizer.state_dict()` of lib.utils.Adam (a torch.optim.Optimizer): 'state'
    # {param index: {'step', 'exp_avg', 'exp_avg_sq'}} + 'param_groups' [{'name', 'lr', 'params': [indices], ...}], one group
    # per `lrate_<name>` key, parameters in `module.parameters()` order (warp_network starts with its gradient-less
    # `progress`, which never gets a state entry).
    # Group order = the order of the `lrate_<name>` keys of the MERGED training config (lib/utils.py:320-341 walks cfg_train.keys()):
    # configs/default_fine_s.py contributes lrate_k0, lrate_rgbnet (coarse_train :34-35) and lrate_sdf (surf_train :77) first, the keys
    # configs/dtu_e2e/scan1.py:87-103 adds (lrate_sdf_alpha, lrate_sdf_beta, ..., lrate_warp_network) follow in its own order (mmengine
    # merges a child into its base: existing keys keep their place, new ones are appended).  `sdf` (lrate_sdf = 0.1 > 0) IS a group: its
    # one parameter, the frozen template, never receives a gradient, so it holds a parameter index but no state entry.
    GROUPS = ('k0', 'rgbnet', 'sdf', 'sdf_alpha', 'sdf_beta', 'warp_network')

    def _group_tensors(self, which):
        """{group name: list of tensors in the reference's parameter order and logical shapes} for 'm' or 'v'."""
        P = self.flat
        ab = P.view('sdf_ab', which)
        k0 = {'m': self.k0_m, 'v': self.k0_v}[which]
        rg = [t for Wb in unpack_rgbnet(P.view('rgbnet', which)) for t in Wb]
        wp = [t for Wb in unpack_warp(P.view('warp', which)) for t in Wb]
        return {'sdf_alpha': [ab[0:1]], 'sdf_beta': [ab[1:2]], 'k0': [self.k0_reference_layout(k0)], 'rgbnet': rg,
                'sdf': [None],                                     # None = a parameter without optimiser state (frozen template)
                'warp_network': [None] + wp}                       # None = warp_network.progress (no state)

    def group_order(self, cfg_train=None):
        """Optimiser group names in the order the reference builds them for `cfg_train` (any mapping whose keys include the
        `lrate_<name>` entries; None = the merged scan1 configuration).  Groups whose rate is not positive are frozen, not grouped."""
        if cfg_train is None:
            return list(self.GROUPS)
        known = set(self.GROUPS)
        return [k[6:] for k in cfg_train.keys() if k.startswith('lrate_') and k[6:] in known and float(cfg_train[k]) > 0]

    def optimizer_state_dict(self, cfg_train=None):
        c = lambda t: t.detach().clone().cpu().contiguous()
        m, v = self._group_tensors('m'), self._group_tensors('v')
        lr = {'sdf_alpha': self.lr['sdf_ab'], 'sdf_beta': self.lr['sdf_ab'], 'k0': self.lr['k0'], 'rgbnet': self.lr['rgbnet'],
              'warp_network': self.lr['warp'], 'sdf': self.lr['k0']}       # lrate_sdf = lrate_k0 = 0.1 in scan1.py, same decay
        state, groups, idx = {}, [], 0
        for name in self.group_order(cfg_train):
            ids = []
            for tm, tv in zip(m[name], v[name]):
                if tm is not None and self.n_step > 0:
                    state[idx] = {'step': self.n_step, 'exp_avg': c(tm), 'exp_avg_sq': c(tv)}
                ids.append(idx)
                idx += 1
            groups.append({'params': ids, 'lr': float(lr[name]), 'name': name, 'betas': (0.9, 0.99), 'eps': 1e-8,
                           'weight_decay': 0, 'amsgrad': False})
        return {'state': state, 'param_groups': groups}

    def pose_optimizer_state_dict(self):
        c = lambda t: t.detach().clone().cpu()
        state = {0: {'step': self.n_step, 'exp_avg': c(self.se3_m), 'exp_avg_sq': c(self.se3_v)}} if self.n_step > 0 else {}
        return {'state': state, 'param_groups': [{'params': [0], 'lr': float(self.lr_pose), 'betas': (0.9, 0.999), 'eps': 1e-8,
                                                  'weight_decay': 0, 'amsgrad': False}]}

    def load_optimizer_state_dict(self, sd, pose_sd=None):
        """Reads the reference's (or this engine's) torch-Adam layout into the flat moment buffers; groups are matched by
        `name`, parameters by position inside their group.  Groups that are absent keep zero moments."""
        d = lambda t: torch.as_tensor(t, dtype=torch.float32).to(self.dev)
        state = sd['state']
        m, v = self._group_tensors('m'), self._group_tensors('v')
        steps = []
        with torch.no_grad():
            for t in (self.k0_m, self.k0_v, self.flat.m, self.flat.v):
                t.zero_()
            for grp in sd['param_groups']:
                name = grp.get('name')
                if name not in m:
                    continue
                if len(grp['params']) != len(m[name]):
                    raise ValueError(f"optimizer group '{name}': {len(grp['params'])} parameters, expected {len(m[name])}")
                for i, tm, tv in zip(grp['params'], m[name], v[name]):
                    st = state.get(i, state.get(str(i)))
                    if st is None or tm is None:
                        continue
                    tm.copy_(d(st['exp_avg']).reshape(tm.shape))
                    tv.copy_(d(st['exp_avg_sq']).reshape(tv.shape))
                    steps.append(int(st['step']))
                key = {'sdf_alpha': 'sdf_ab', 'sdf_beta': 'sdf_ab', 'warp_network': 'warp'}.get(name, name)
                if key in self.lr:                                 # 'sdf': the frozen template has a group but no rate of ours
                    self.lr[key] = float(grp['lr'])
            if pose_sd is not None and pose_sd.get('state'):
                st = pose_sd['state'].get(0, pose_sd['state'].get('0'))
                self.se3_m.copy_(d(st['exp_avg']))
                self.se3_v.copy_(d(st['exp_avg_sq']))
                self.lr_pose = float(pose_sd['param_groups'][0]['lr'])
        if steps:
            if len(set(steps)) != 1:
                raise ValueError(f'optimizer state with differing step counts {sorted(set(steps))}: the fused Adam keeps one')
            self.n_step = steps[0]

    def load_training_state(self, tensors, se3, se3_m, se3_v, n_step, lr, lr_pose):
        """Resume from an IN-MEMORY training state given in the reference's terms: tensors = {name: (param, exp_avg, exp_avg_sq)}
        with names 'k0' ([1,C,X,Y,Z]), 'sdf_alpha', 'sdf_beta', 'rgbnet.<l>.weight|bias', 'warp.<l>.weight|bias' (logical
        shapes of the reference's modules); lr = {'k0', 'rgbnet', 'warp', 'sdf_ab'}.  Used to put this engine at the state of
        another trainer (teacher-forced parity runs, bench.py); the file-based twin is load_checkpoint."""
        d = lambda t: t.detach().to(self.dev, torch.float32)
        cl = lambda t: d(t)[0].permute(1, 2, 3, 0)
        with torch.no_grad():
            p, m, v = tensors['k0']
            self.k0_cl.copy_(cl(p)); self.k0_m.copy_(cl(m)); self.k0_v.copy_(cl(v))
            F = self.flat
            for which, pick in (('data', 0), ('m', 1), ('v', 2)):
                ab = F.view('sdf_ab', which)
                ab[0:1].copy_(d(tensors['sdf_alpha'][pick]).reshape(1)); ab[1:2].copy_(d(tensors['sdf_beta'][pick]).reshape(1))
                for li, (W, b) in enumerate(unpack_rgbnet(F.view('rgbnet', which))):
                    W.copy_(d(tensors[f'rgbnet.{li}.weight'][pick])); b.copy_(d(tensors[f'rgbnet.{li}.bias'][pick]))
                for li, (W, b) in enumerate(unpack_warp(F.view('warp', which))):
                    W.copy_(d(tensors[f'warp.{li}.weight'][pick])); b.copy_(d(tensors[f'warp.{li}.bias'][pick]))
            self.se3.copy_(d(se3)); self.se3_m.copy_(d(se3_m)); self.se3_v.copy_(d(se3_v))
        self.n_step = int(n_step)
        for k in self.lr:
            self.lr[k] = float(lr[k])
        self.lr_pose = float(lr_pose)

    def model_kwargs(self):
        """Constructor arguments of the drop-in `Voxurf` for this engine's configuration, as plain Python values (a checkpoint
        holding them loads weights-only; utils.load_model rebuilds the module from them)."""
        cfg = self.cfg
        rs = [float(cfg.out_range)] * 3
        return {'xyz_min': [float(x) for x in cfg.xyz_min], 'xyz_max': [float(x) for x in cfg.xyz_max],
                'num_voxels': int(cfg.num_voxels), 'num_voxels_base': int(cfg.num_voxels), 'alpha_init': 1e-2,
                'rgbnet_dim': int(cfg.k0_dim), 'rgbnet_direct': True, 'rgbnet_depth': 4, 'rgbnet_width': 128,
                'posbase_pe': int(cfg.posbase_pe), 'viewbase_pe': int(cfg.viewbase_pe), 'geo_rgb_dim': 3,
                's_ratio': float(cfg.s_ratio), 's_start': float(cfg.s_start), 'step_start': float(cfg.step_start),
                'barf_c2f': None if cfg.barf_c2f is None else [float(x) for x in cfg.barf_c2f], 'N_iters': int(cfg.N_iters),
                'i_train': list(range(self.V)), 'HW': [[int(self.H), int(self.W)]] * self.V, 'camera_noise': 0.0,
                'range_shape': rs, 'rect_size': rs}

    def save_checkpoint(self, path, global_step, cfg_train=None):
        """The reference's `last_ckpt.tar` (lib/recon_scene.py:779-791): `global_step`, `current_pose`, `model_kwargs`,
        `MaskCache_kwargs`, `model_state_dict`, `optimizer_state_dict`, `optimizer_pose_state_dict` - the optimiser entries in
        torch-Adam layout - plus this engine's pose parametrisation (`se3_refine`, `w2c_init`) and schedule state."""
        ops.pose_fwd(self.se3, self.w2c_init, self.refine_mask, self.w2c, self.c2w, self.jac)
        c = lambda t: t.detach().clone().cpu()
        cfg = self.cfg
        torch.save({'global_step': int(global_step), 'current_pose': c(self.w2c),
                    'model_kwargs': self.model_kwargs(),
                    'MaskCache_kwargs': {'xyz_min': [float(x) for x in cfg.xyz_min], 'xyz_max': [float(x) for x in cfg.xyz_max],
                                         'act_shift': float(np.log(1 / (1 - 1e-2) - 1)), 'voxel_size_ratio': 1.0, 'nearest': False},
                    'model_state_dict': self.model_state_dict(), 'optimizer_state_dict': self.optimizer_state_dict(cfg_train),
                    'optimizer_pose_state_dict': self.pose_optimizer_state_dict(),
                    'se3_refine': c(self.se3), 'w2c_init': c(self.w2c_init),
                    'engine': {'format': 'poseprobe_amd.TrainEngine/2', 'n_step': self.n_step, 'lr': dict(self.lr),
                               'lr_pose': self.lr_pose}}, path)

    def load_checkpoint(self, path, reload_optimizer=True):
        """-> global_step.  Accepts this engine's files and reference `last_ckpt.tar` files (model under the reference's names,
        optimiser in torch-Adam layout).  Weights-only load (numpy arrays in `model_kwargs` admitted): nothing is executed."""
        from .utils import load_checkpoint_file
        ck = load_checkpoint_file(path)
        self.load_model_state_dict(ck['model_state_dict'])
        d = lambda t: torch.as_tensor(t, dtype=torch.float32).to(self.dev)
        if 'se3_refine' in ck:
            self.se3.copy_(d(ck['se3_refine']))
            self.w2c_init.copy_(d(ck['w2c_init']))
        elif 'current_pose' in ck:                          # a reference file: start from its poses, refinement at zero
            cp = ck['current_pose']
            cp = torch.stack([torch.as_tensor(p) for p in cp]) if isinstance(cp, (list, tuple)) else torch.as_tensor(cp)
            self.w2c_init.copy_(d(cp)[:, :3, :4])
            self.se3.zero_()
        opt = ck.get('optimizer_state_dict', {})
        if reload_optimizer and opt.get('format') == 'poseprobe_amd.TrainEngine/1':     # round-1 files
            self.n_step, self.lr, self.lr_pose = int(opt['n_step']), dict(opt['lr']), float(opt['lr_pose'])
            self.k0_m.copy_(d(opt['k0.exp_avg'])); self.k0_v.copy_(d(opt['k0.exp_avg_sq']))
            self.flat.m.copy_(d(opt['flat.exp_avg'])); self.flat.v.copy_(d(opt['flat.exp_avg_sq']))
            self.se3_m.copy_(d(opt['se3.exp_avg'])); self.se3_v.copy_(d(opt['se3.exp_avg_sq']))
        elif reload_optimizer and 'param_groups' in opt:
            self.load_optimizer_state_dict(opt, ck.get('optimizer_pose_state_dict'))
            eng = ck.get('engine')
            if eng is not None:
                self.n_step, self.lr, self.lr_pose = int(eng['n_step']), dict(eng['lr']), float(eng['lr_pose'])
        self.zero_grads()
        return int(ck.get('global_step', 0))

    def losses(self):
        """dict of the unweighted loss scalars of the last step (one D2H copy)."""
        v = self.ws.loss_out.cpu().numpy()
        names = ['img_render', 'weight_entropy_last', 'grad_constraint', 'grad_deform_constraint',
                 'sdf_correct_constraint', 'sdf_deform_constraint', 'mask_render']
        return {n: float(v[i]) for i, n in enumerate(names)}
