"""Thin, allocation-explicit Python wrappers over the C ABI (include/poseprobe_hip.h).

One function per entry point; tensors are torch CUDA tensors used purely as device buffers (torch is
plumbing: memory + streams).  Error behaviour mirrors the reference extension
(lib/cuda/render_utils.cpp:46-48 CHECK_INPUT): non-CUDA or non-contiguous inputs raise RuntimeError.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import pp_scene


def _ptr(t, dtype=None, name='tensor'):
    if t is None:
        return None
    if not isinstance(t, torch.Tensor):
        raise TypeError(f'{name} must be a torch.Tensor')
    if not t.is_cuda:
        raise RuntimeError(f'{name} must be a CUDA tensor')
    if not t.is_contiguous():
        raise RuntimeError(f'{name} must be contiguous')
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError(f'{name} must be {dtype}, got {t.dtype}')
    return ctypes.c_void_p(t.data_ptr())


def _f(t, name='tensor'):
    return _ptr(t, torch.float32, name)


def _i(t, name='tensor'):
    return _ptr(t, torch.int32, name)


def _u8(t, name='tensor'):
    return _ptr(t, torch.uint8, name)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def make_scene(xyz_min, xyz_max, world_size, voxel_size, stepsize, near, far, bg, out_range=1.0, k0_dim=12,
               pos_pe=5, view_pe=1, sdf_index_exact=None):
    sc = pp_scene()
    for i in range(3):
        sc.xyz_min[i] = float(xyz_min[i])
        sc.xyz_max[i] = float(xyz_max[i])
        sc.size[i] = int(world_size[i])
    sc.voxel_size = float(np.float32(voxel_size))
    sc.stepsize = float(stepsize)
    sc.near_clip = float(near)
    sc.far_clip = float(far)
    sc.bg = float(bg)
    # voxurf_coarse.py:700
    sc.n_samples = int(np.linalg.norm(np.array([int(w) for w in world_size]) + 1) / stepsize) + 1
    sc.out_range = float(out_range)
    sc.k0_dim, sc.pos_pe, sc.view_pe = int(k0_dim), int(pos_pe), int(view_pe)
    if sdf_index_exact is None:         # host-side switch (PP_SDF_INDEX_EXACT=1): the mathematically intended voxel index above 2^24 voxels
        import os
        sdf_index_exact = os.environ.get('PP_SDF_INDEX_EXACT') == '1'
    sc.sdf_index_exact = int(bool(sdf_index_exact))
    return sc


RGBNET_PARAMS = 128 * 64 + 128 + 2 * (128 * 128 + 128) + 3 * 128 + 3
WARP_PARAMS = 128 * 3 + 128 + 3 * (128 * 128 + 128) + 4 * 128 + 4
FEAT_LD = 64


# ------------------------------------------------------------------------------------------- pose / rays
def pose_fwd(se3, w2c_init, refine_mask, w2c, c2w, jac):
    V = se3.shape[0]
    _lib.call('pp_pose_fwd', _f(se3), _f(w2c_init), _i(refine_mask), V, _f(w2c), _f(c2w), _f(jac), _stream())


def pose_bwd(jac, c2w_grad, se3_grad):
    _lib.call('pp_pose_bwd', _f(jac), _f(c2w_grad), se3_grad.shape[0], _f(se3_grad), _stream())


def raygen_select_fwd(sc, ray_idx, c2w, intr, H, W, inverse_y, normalize, images, masks, rays_o, rays_d, viewdirs,
                      target, mask_px):
    _lib.call('pp_raygen_select_fwd', ctypes.byref(sc), _i(ray_idx), ray_idx.shape[0], _f(c2w), _f(intr),
              c2w.shape[0], H, W, int(inverse_y), int(normalize), _f(images), _f(masks), _f(rays_o), _f(rays_d),
              _f(viewdirs), _f(target), _f(mask_px), _stream())


def raygen_select_bwd(sc, ray_idx, c2w, intr, H, W, inverse_y, rays_o, rays_d, t_min, ray_start, pts_grad, step,
                      vgrad_s, g_o, g_d, g_v, g_depth, g_o_out, g_d_out, g_v_out, c2w_grad):
    n_views = c2w.shape[0] if c2w is not None else 0
    _lib.call('pp_raygen_select_bwd', ctypes.byref(sc), _i(ray_idx), rays_o.shape[0], _f(c2w), _f(intr), n_views,
              H, W, int(inverse_y), _f(rays_o), _f(rays_d), _f(t_min), _i(ray_start), _f(pts_grad), _f(step),
              _f(vgrad_s), _f(g_o), _f(g_d), _f(g_v), _f(g_depth), _f(g_o_out), _f(g_d_out), _f(g_v_out),
              _f(c2w_grad), _stream())


def sample_dense(sc, rays_o, rays_d, jitter, capacity, t_min, t_max, ray_start, count, pts, ray_id, step_k, step,
                 mask_keep=None):
    _lib.call('pp_sample_dense', ctypes.byref(sc), _f(rays_o), _f(rays_d), _f(jitter), rays_o.shape[0], capacity,
              _f(t_min), _f(t_max), _i(ray_start), _i(count), _f(pts), _i(ray_id), _i(step_k), _f(step),
              _ptr(mask_keep, torch.uint8), _stream())


def sample_var(sc, rays_o, rays_d, capacity, t_min, t_max, n_steps, ray_start, count, pts, ray_id, step_id):
    _lib.call('pp_sample_var', ctypes.byref(sc), _f(rays_o), _f(rays_d), rays_o.shape[0], capacity, _f(t_min),
              _f(t_max), _i(n_steps), _i(ray_start), _i(count), _f(pts), _i(ray_id), _i(step_id), _stream())


# ------------------------------------------------------------------------------------------- scan / composite
def alpha2weight_fwd(alpha, ray_start, n_rays, weights, T, alphainv_last, i_end):
    _lib.call('pp_alpha2weight_fwd', _f(alpha), _i(ray_start), n_rays, _f(weights), _f(T), _f(alphainv_last),
              _i(i_end), _stream())


def alpha2weight_bwd(alpha, weights, T, alphainv_last, ray_start, i_end, n_rays, grad_weights, grad_last, grad_alpha):
    _lib.call('pp_alpha2weight_bwd', _f(alpha), _f(weights), _f(T), _f(alphainv_last), _i(ray_start), _i(i_end),
              n_rays, _f(grad_weights), _f(grad_last), _f(grad_alpha), _stream())


def march_fwd(alpha, rgb, step_w, nrm_in, ray_start, n_rays, bg, weights, T, alphainv_last, i_end, rgb_marched,
              rgb_pre, cum_weights, depth_acc, normal_marched):
    _lib.call('pp_march_fwd', _f(alpha), _f(rgb), _f(step_w), _f(nrm_in), _i(ray_start), n_rays, float(bg),
              _f(weights), _f(T), _f(alphainv_last), _i(i_end), _f(rgb_marched), _f(rgb_pre), _f(cum_weights),
              _f(depth_acc), _f(normal_marched), _stream())


def march_bwd(alpha, rgb, step_w, weights, T, alphainv_last, ray_start, i_end, n_rays, bg, rgb_pre, g_rgbm, g_cw,
              g_last, g_depth, g_weights, grad_alpha, grad_rgb):
    _lib.call('pp_march_bwd', _f(alpha), _f(rgb), _f(step_w), _f(weights), _f(T), _f(alphainv_last), _i(ray_start),
              _i(i_end), n_rays, float(bg), _f(rgb_pre), _f(g_rgbm), _f(g_cw), _f(g_last), _f(g_depth),
              _f(g_weights), _f(grad_alpha), _f(grad_rgb), _stream())


# ------------------------------------------------------------------------------------------- geometry / colour
def geometry_fwd(sc, sdf_grid, sdf_ab, pts, warp_out, viewdirs, ray_id, count, capacity, inv_s, alpha, gradient,
                 sdf_final, sdf_deform, grad_deform):
    _lib.call('pp_geometry_fwd', ctypes.byref(sc), _f(sdf_grid), _f(sdf_ab), _f(pts), _f(warp_out), _f(viewdirs),
              _i(ray_id), _i(count), capacity, float(inv_s), _f(alpha), _f(gradient), _f(sdf_final), _f(sdf_deform),
              _f(grad_deform), _stream())


def geometry_bwd(sc, sdf_grid, sdf_ab, pts, warp_out, viewdirs, ray_id, count, capacity, inv_s, g_alpha, g_gradient,
                 g_sdf_final, g_sdf_deform, g_grad_deform, g_correction, accumulate, warp_out_grad, pts_grad, vgrad_s,
                 sdf_ab_grad):
    _lib.call('pp_geometry_bwd', ctypes.byref(sc), _f(sdf_grid), _f(sdf_ab), _f(pts), _f(warp_out), _f(viewdirs),
              _i(ray_id), _i(count), capacity, float(inv_s), _f(g_alpha), _f(g_gradient), _f(g_sdf_final),
              _f(g_sdf_deform), _f(g_grad_deform), _f(g_correction), int(accumulate), _f(warp_out_grad),
              _f(pts_grad), _f(vgrad_s), _f(sdf_ab_grad), _stream())


def color_feat_fwd(sc, k0_cl, pts, viewdirs, ray_id, gradient, pe_w, count, capacity, feat):
    _lib.call('pp_color_feat_fwd', ctypes.byref(sc), _f(k0_cl), _f(pts), _f(viewdirs), _i(ray_id), _f(gradient),
              _f(pe_w), _i(count), capacity, _f(feat), _stream())


def color_feat_bwd(sc, k0_cl, pts, viewdirs, ray_id, gradient, pe_w, count, capacity, feat_grad, k0_grad_cl, pts_grad,
                   gradient_grad, vgrad_s):
    _lib.call('pp_color_feat_bwd', ctypes.byref(sc), _f(k0_cl), _f(pts), _f(viewdirs), _i(ray_id), _f(gradient),
              _f(pe_w), _i(count), capacity, _f(feat_grad), _f(k0_grad_cl), _f(pts_grad), _f(gradient_grad),
              _f(vgrad_s), _stream())


def geometry_bwd_priors(sc, sdf_grid, sdf_ab, pts, warp_out, viewdirs, ray_id, count, capacity, inv_s, g_alpha, g_gradient,
                        w_eikonal, w_deform, loss_scale, accumulate, warp_out_grad, pts_grad, vgrad_s, sdf_ab_grad, loss_out,
                        batch_norm=None):
    _lib.call('pp_geometry_bwd_priors', ctypes.byref(sc), _f(sdf_grid), _f(sdf_ab), _f(pts), _f(warp_out), _f(viewdirs),
              _i(ray_id), _i(count), capacity, float(inv_s), _f(g_alpha), _f(g_gradient), float(w_eikonal), float(w_deform),
              float(loss_scale), int(accumulate), _f(warp_out_grad), _f(pts_grad), _f(vgrad_s), _f(sdf_ab_grad),
              _f(loss_out), _f(batch_norm), _stream())


def k0_pack_samples(pts, feat_grad, count, capacity, k0_dim, packed):
    _lib.call('pp_k0_pack_samples', _f(pts), _f(feat_grad), _i(count), capacity, int(k0_dim), _f(packed), _stream())


def k0_scatter_packed(sc, packed, n_shards, capacity, k0_grad_cl, touched=None):
    _lib.call('pp_k0_scatter_packed', ctypes.byref(sc), _f(packed), int(n_shards), capacity, _f(k0_grad_cl), _u8(touched),
              _stream())


def k0_scatter_samples(sc, pts, count, capacity, feat_grad, k0_grad_cl, touched=None):
    _lib.call('pp_k0_scatter_samples', ctypes.byref(sc), _f(pts), _i(count), capacity, _f(feat_grad), _f(k0_grad_cl),
              _u8(touched), _stream())


def k0_scatter_sorted_workspace(n_samples):
    """Bytes of device workspace the deterministic scatters need for n_samples = n_shards * capacity samples."""
    b = ctypes.c_int64()
    _lib.call('pp_k0_scatter_sorted_workspace', int(n_samples), ctypes.byref(b))
    return b.value


def k0_scatter_samples_sorted(sc, pts, count, capacity, feat_grad, k0_grad_cl, work, touched=None):
    """Deterministic pp_k0_scatter_samples: contributions sorted by voxel and added in sample order (work: uint8 tensor)."""
    _lib.call('pp_k0_scatter_samples_sorted', ctypes.byref(sc), _f(pts), _i(count), capacity, _f(feat_grad), _f(k0_grad_cl),
              _u8(touched), _u8(work), int(work.numel()), _stream())


def k0_scatter_packed_sorted(sc, packed, n_shards, capacity, k0_grad_cl, work, touched=None):
    _lib.call('pp_k0_scatter_packed_sorted', ctypes.byref(sc), _f(packed), int(n_shards), capacity, _f(k0_grad_cl), _u8(touched),
              _u8(work), int(work.numel()), _stream())


# ------------------------------------------------------------------------------------------- MLPs
# Every wrapper of an option-dependent entry point takes `ctx`: an _lib.Context, or None = the host's default context.
Context = _lib.Context
_h = _lib.handle


def rgbnet_fwd(params, feat, count, capacity, acts, rgb, ctx=None):
    _lib.call('pp_rgbnet_fwd', _f(params), _f(feat), _i(count), capacity, _f(acts), _f(rgb), _h(ctx), _stream())


def context_join(ctx=None):
    """Current stream waits for every weight-gradient kernel deferred onto the context's auxiliary stream (option side_stream)."""
    _lib.call('pp_context_join', _h(ctx), _stream())


def rgbnet_bwd(params, feat, acts, rgb, rgb_grad, count, capacity, scratch, params_grad, feat_grad, ctx=None):
    _lib.call('pp_rgbnet_bwd', _f(params), _f(feat), _f(acts), _f(rgb), _f(rgb_grad), _i(count), capacity,
              _f(scratch), _f(params_grad), _f(feat_grad), _h(ctx), _stream())


def warp_fwd(params, pts, count, capacity, out_range, acts, out, ctx=None):
    _lib.call('pp_warp_fwd', _f(params), _f(pts), _i(count), capacity, float(out_range), _f(acts), _f(out), _h(ctx), _stream())


def warp_bwd(params, pts, acts, out_grad, count, capacity, out_range, scratch, params_grad, pts_grad, ctx=None):
    _lib.call('pp_warp_bwd', _f(params), _f(pts), _f(acts), _f(out_grad), _i(count), capacity, float(out_range),
              _f(scratch), _f(params_grad), _f(pts_grad), _h(ctx), _stream())


def mlp_workspaces(capacity):
    """-> {'rgbnet': (acts, scratch), 'warp': (acts, scratch)} in floats, from the library's workspace queries."""
    out = {}
    for name in ('rgbnet', 'warp'):
        a, s = ctypes.c_int64(), ctypes.c_int64()
        _lib.call(f'pp_{name}_workspace', int(capacity), ctypes.byref(a), ctypes.byref(s))
        out[name] = (a.value, s.value)
    return out


def warp_bwd_data(params, pts, acts, out_grad, count, capacity, out_range, scratch, params_grad, pts_grad, ctx=None):
    """-> stage2: the value to hand to warp_bwd_weights (which stage owns the hidden layers' bias gradients)."""
    stage2 = ctypes.c_int32()
    _lib.call('pp_warp_bwd_data', _f(params), _f(pts), _f(acts), _f(out_grad), _i(count), capacity, float(out_range),
              _f(scratch), _f(params_grad), _f(pts_grad), ctypes.byref(stage2), _h(ctx), _stream())
    return stage2.value


def warp_bwd_weights(acts, scratch, count, capacity, params_grad, stage2, ctx=None):
    _lib.call('pp_warp_bwd_weights', _f(acts), _f(scratch), _i(count), capacity, _f(params_grad), int(stage2), _h(ctx), _stream())


def rgbnet_bwd_data(params, acts, rgb, rgb_grad, count, capacity, scratch, params_grad, feat_grad, ctx=None):
    stage2 = ctypes.c_int32()
    _lib.call('pp_rgbnet_bwd_data', _f(params), _f(acts), _f(rgb), _f(rgb_grad), _i(count), capacity, _f(scratch),
              _f(params_grad), _f(feat_grad), ctypes.byref(stage2), _h(ctx), _stream())
    return stage2.value


def rgbnet_bwd_weights(feat, acts, scratch, count, capacity, params_grad, stage2, ctx=None):
    _lib.call('pp_rgbnet_bwd_weights', _f(feat), _f(acts), _f(scratch), _i(count), capacity, _f(params_grad), int(stage2),
              _h(ctx), _stream())


# ------------------------------------------------------------------------------------------- losses / optimiser
def loss_rays(rgb_marched, alphainv_last, cum_weights, target, mask_px, mask_sum, w_main, w_entropy, w_mask,
              loss_scale, g_rgbm, g_last, g_cw, loss_out, batch_norm=None):
    _lib.call('pp_loss_rays', _f(rgb_marched), _f(alphainv_last), _f(cum_weights), _f(target), _f(mask_px),
              _f(mask_sum), rgb_marched.shape[0], float(w_main), float(w_entropy), float(w_mask), float(loss_scale),
              _f(g_rgbm), _f(g_last), _f(g_cw), _f(loss_out), _f(batch_norm), _stream())


def loss_samples(gradient, grad_deform, warp_out, sdf_deform, count, capacity, w_eik, w_deform, loss_scale, g_gradient,
                 g_grad_deform, g_correction, g_sdf_deform, loss_out, batch_norm=None):
    _lib.call('pp_loss_samples', _f(gradient), _f(grad_deform), _f(warp_out), _f(sdf_deform), _i(count), capacity,
              float(w_eik), float(w_deform), float(loss_scale), _f(g_gradient), _f(g_grad_deform), _f(g_correction),
              _f(g_sdf_deform), _f(loss_out), _f(batch_norm), _stream())


def grid_tv_adam_step(p_in, p_out, grad, exp_avg, exp_avg_sq, size, channels, x_begin, x_end, tv_scale, grad_scale, lr,
                      beta1, beta2, eps, step, tv_out, ctx=None):
    _lib.call('pp_grid_tv_adam_step', _f(p_in), _f(p_out), _f(grad), _f(exp_avg), _f(exp_avg_sq),
              int(size[0]), int(size[1]), int(size[2]), channels, x_begin, x_end, float(tv_scale), float(grad_scale),
              float(lr), float(beta1), float(beta2), float(eps), int(step), _f(tv_out), _h(ctx), _stream())


def grid_tv_adam_step_sparse(p_in, p_out, grad, exp_avg, exp_avg_sq, size, channels, x_begin, x_end, tv_scale, grad_scale,
                             lr, beta1, beta2, eps, step, tv_out, touched, touched_clear, ctx=None):
    _lib.call('pp_grid_tv_adam_step_sparse', _f(p_in), _f(p_out), _f(grad), _f(exp_avg), _f(exp_avg_sq),
              int(size[0]), int(size[1]), int(size[2]), channels, x_begin, x_end, float(tv_scale), float(grad_scale),
              float(lr), float(beta1), float(beta2), float(eps), int(step), _f(tv_out), _u8(touched), _u8(touched_clear),
              _h(ctx), _stream())


def grid_tv_value(p, size, channels, out):
    _lib.call('pp_grid_tv_value', _f(p), int(size[0]), int(size[1]), int(size[2]), channels, _f(out), _stream())


def adam_flat(p, grad, exp_avg, exp_avg_sq, seg_end, seg_lr, grad_scale, beta1, beta2, eps, step, zero_grad):
    _lib.call('pp_adam_flat', _f(p), _f(grad), _f(exp_avg), _f(exp_avg_sq), p.numel(), _i(seg_end), _f(seg_lr),
              seg_end.numel(), float(grad_scale), float(beta1), float(beta2), float(eps), int(step), int(zero_grad),
              _stream())


def grid_sample_fwd(sc, grid_cl, channels, pts, border, out):
    _lib.call('pp_grid_sample_fwd', ctypes.byref(sc), _f(grid_cl), channels, _f(pts), pts.shape[0], int(border), _f(out),
              _stream())


def grid_sample_bwd(sc, grid_cl, channels, pts, border, out_grad, grid_grad_cl, pts_grad):
    _lib.call('pp_grid_sample_bwd', ctypes.byref(sc), _f(grid_cl), channels, _f(pts), pts.shape[0], int(border),
              _f(out_grad), _f(grid_grad_cl), _f(pts_grad), _stream())


def grid_tv_grad(p, size, channels, scale, g_scalar, grad):
    _lib.call('pp_grid_tv_grad', _f(p), int(size[0]), int(size[1]), int(size[2]), channels, float(scale), _f(g_scalar),
              _f(grad), _stream())


def sdf_first_crossing(sdf, ray_start, step_k, n_rays, n_samples, dist, t_min, rays_o, rays_d, sdf_dense, pts, mask, zval):
    _lib.call('pp_sdf_first_crossing', _f(sdf), _i(ray_start), _i(step_k), n_rays, n_samples, float(dist), _f(t_min),
              _f(rays_o), _f(rays_d), _f(sdf_dense), _f(pts), _ptr(mask, torch.uint8), _f(zval), _stream())


def sdf_crossing_dense_bwd(sc, sdf_grid, rays_o, rays_d, t_min, jitter, n_rays, n_samples, dist, sdf_dense, g_pts, g_sdf_dense,
                           g_rays_o, g_rays_d, g_t_min):
    _lib.call('pp_sdf_crossing_dense_bwd', ctypes.byref(sc), _f(sdf_grid), _f(rays_o), _f(rays_d), _f(t_min), _f(jitter),
              n_rays, n_samples, float(dist), _f(sdf_dense), _f(g_pts), _f(g_sdf_dense), _f(g_rays_o), _f(g_rays_d),
              _f(g_t_min), _stream())


def feat_generic_fwd(sc, k0_cl, pts, viewdirs, ray_id, gradient, pe_w, sel, k0_skip, ld, count, capacity, feat, k0_raw):
    _lib.call('pp_feat_generic_fwd', ctypes.byref(sc), _f(k0_cl), _f(pts), _f(viewdirs), _i(ray_id), _f(gradient), _f(pe_w),
              _ptr(sel, torch.uint8), int(k0_skip), int(ld), _i(count), capacity, _f(feat), _f(k0_raw), _stream())


def feat_generic_bwd_k0(sc, pts, sel, k0_skip, ld, count, capacity, feat_grad, k0_raw_grad, k0_grad_cl):
    _lib.call('pp_feat_generic_bwd_k0', ctypes.byref(sc), _f(pts), _ptr(sel, torch.uint8), int(k0_skip), int(ld), _i(count),
              capacity, _f(feat_grad), _f(k0_raw_grad), _f(k0_grad_cl), _stream())


def mlp_fwd(params, feat, in_ld, n_gemm, count, capacity, logit_add, logit_add_ld, acts, out, ctx=None):
    _lib.call('pp_mlp_fwd', _f(params), _f(feat), int(in_ld), int(n_gemm), _i(count), capacity, _f(logit_add),
              int(logit_add_ld), _f(acts), _f(out), _h(ctx), _stream())


def mlp_bwd(params, feat, in_ld, n_gemm, acts, out, out_grad, count, capacity, scratch, params_grad, feat_grad,
            logit_add_grad, logit_add_ld, ctx=None):
    _lib.call('pp_mlp_bwd', _f(params), _f(feat), int(in_ld), int(n_gemm), _f(acts), _f(out), _f(out_grad), _i(count),
              capacity, _f(scratch), _f(params_grad), _f(feat_grad), _f(logit_add_grad), int(logit_add_ld), _h(ctx),
              _stream())


def march_dvgo_fwd(alpha, rgb, step_w, ray_start, n_rays, weights, T, alphainv_last, i_end, rgb_acc, cum_weights, depth_acc):
    _lib.call('pp_march_dvgo_fwd', _f(alpha), _f(rgb), _f(step_w), _i(ray_start), n_rays, _f(weights), _f(T),
              _f(alphainv_last), _i(i_end), _f(rgb_acc), _f(cum_weights), _f(depth_acc), _stream())


# ------------------------------------------------------------------------------------------- scene branch (NeRF)
def nerf_layout():
    """Offsets (floats) of {W0,b0,...,W7,b7,wd,bd,R0,br0,R1,br1} in the packed parameter block and its total size."""
    off = (ctypes.c_int64 * 23)()
    _lib.call('pp_nerf_layout', off)
    return list(off)


def nerf_workspace(n_samples, n_rays):
    a, s = ctypes.c_int64(), ctypes.c_int64()
    _lib.call('pp_nerf_workspace', ctypes.c_int64(n_samples), ctypes.c_int64(n_rays), ctypes.byref(a), ctypes.byref(s))
    return a.value, s.value


def nerf_fwd(params, center, ray, depth, bands, count, n_rays, n_samples, acts, rgb_samples, density_samples, ctx=None):
    _lib.call('pp_nerf_fwd', _f(params), _f(center), _f(ray), _f(depth), _f(bands), _i(count), int(n_rays), int(n_samples),
              _f(acts), _f(rgb_samples), _f(density_samples), _h(ctx), _stream())


def nerf_bwd(params, ray, depth, count, n_rays, n_samples, acts, rgb_samples, g_rgb_samples, g_density_samples, scratch,
             params_grad, g_center, g_ray, ctx=None):
    _lib.call('pp_nerf_bwd', _f(params), _f(ray), _f(depth), _i(count), int(n_rays), int(n_samples), _f(acts), _f(rgb_samples),
              _f(g_rgb_samples), _f(g_density_samples), _f(scratch), _f(params_grad), _f(g_center), _f(g_ray), _h(ctx), _stream())


def nerf_composite_fwd(rgb_samples, density_samples, depth, ray, n_rays, n_samples, white_bg, rgb, depth_out, opacity, weights,
                       all_cumulated, rgb_var, depth_var):
    _lib.call('pp_nerf_composite_fwd', _f(rgb_samples), _f(density_samples), _f(depth), _f(ray), int(n_rays), int(n_samples),
              int(bool(white_bg)), _f(rgb), _f(depth_out), _f(opacity), _f(weights), _f(all_cumulated), _f(rgb_var),
              _f(depth_var), _stream())


def nerf_composite_bwd(rgb_samples, density_samples, depth, ray, weights, n_rays, n_samples, white_bg, g_rgb, g_depth, g_opacity,
                       g_weights, g_rgb_samples, g_density_samples, g_ray):
    _lib.call('pp_nerf_composite_bwd', _f(rgb_samples), _f(density_samples), _f(depth), _f(ray), _f(weights), int(n_rays),
              int(n_samples), int(bool(white_bg)), _f(g_rgb), _f(g_depth), _f(g_opacity), _f(g_weights), _f(g_rgb_samples),
              _f(g_density_samples), _f(g_ray), _stream())


def nerf_band_weights(progress, start, end, l_3d, l_view, bands):
    _lib.call('pp_nerf_band_weights', _f(progress), ctypes.c_float(start), ctypes.c_float(float(end) - float(start)), int(l_3d), int(l_view), _f(bands),
              _stream())


def nerf_huber_loss(pred, label, delta, weight, loss, g_pred):
    """loss[0] = weight * huber(pred, label, delta, mean); g_pred = its gradient (base_losses.py:155-156)."""
    _lib.call('pp_nerf_huber_loss', _f(pred), _f(label), int(pred.numel()), ctypes.c_float(delta), ctypes.c_float(weight), _f(loss),
              _f(g_pred), _stream())
