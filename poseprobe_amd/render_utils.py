"""Function-level mirror of the reference's native extension modules
(`render_utils_cuda`, `adam_upd_cuda`, `total_variation_cuda`, `ub360_utils_cuda`; lib/cuda/*.cpp) over the C ABI:
same names, argument meaning and return tuples, torch CUDA tensors in / out."""
import ctypes

import torch

from . import _lib, ops
from .ops import _f, _i, _ptr, _stream


def _u8(t):
    return _ptr(t, torch.uint8)


def _scene_from(xyz_min, xyz_max):
    return ops.make_scene(xyz_min.tolist(), xyz_max.tolist(), [2, 2, 2], 1.0, 1.0, 0., 1., 0.)


# ---- render_utils_cuda --------------------------------------------------------------------------------------
def alpha2weight(alpha, ray_id, N):
    """-> weight, T, alphainv_last, i_start, i_end   (lib/cuda/render_utils_kernel.cu:619-651)"""
    from .voxurf_coarse import ray_start_from_ids
    alpha = alpha.contiguous().float()
    M = alpha.shape[0]
    rs = ray_start_from_ids(ray_id, N)
    w, T, last = torch.empty_like(alpha), torch.empty_like(alpha), torch.empty(N, device=alpha.device)
    i_end = torch.empty(N, device=alpha.device, dtype=torch.int32)
    if M > 0:
        ops.alpha2weight_fwd(alpha, rs, N, w, T, last, i_end)
    else:
        last.fill_(1.0)
        i_end.zero_()
    return w, T, last, rs[:-1].long(), i_end.long()


def alpha2weight_backward(alpha, weight, T, alphainv_last, i_start, i_end, n_rays, grad_weights, grad_last):
    g = torch.zeros_like(alpha)
    if alpha.numel() == 0:
        return g
    rs = torch.cat([i_start.int(), torch.tensor([alpha.numel()], dtype=torch.int32, device=alpha.device)])
    # ray_start must be the exclusive prefix: empty rays carry the next ray's start
    for_fix = rs.clone()
    nonempty = torch.zeros(n_rays + 1, dtype=torch.bool, device=alpha.device)
    nonempty[:-1] = i_end > i_start
    nonempty[-1] = True
    idx = torch.arange(n_rays + 1, device=alpha.device)
    nxt = torch.where(nonempty, idx, torch.full_like(idx, n_rays + 1)).flip(0).cummin(0).values.flip(0)
    rs = for_fix[nxt.clamp(max=n_rays)].contiguous()
    ops.alpha2weight_bwd(alpha.contiguous().float(), weight.contiguous(), T.contiguous(), alphainv_last.contiguous(), rs,
                         i_end.int().contiguous(), n_rays, grad_weights.contiguous().float(),
                         grad_last.contiguous().float(), g)
    return g


def raw2alpha(density, shift, interval):
    """-> exp_d, alpha   (render_utils_kernel.cu:431-504; tensor `interval` selects raw2alpha_nonuni)"""
    d = density.contiguous().float()
    e, a = torch.empty_like(d), torch.empty_like(d)
    if d.numel() == 0:
        return e, a                      # render_utils_kernel.cu:468-470
    iv = interval.contiguous().float() if isinstance(interval, torch.Tensor) else None
    _lib.call('pp_raw2alpha_fwd', _f(d), float(shift), 0.0 if iv is not None else float(interval), _f(iv), d.numel(),
              _f(e), _f(a), _stream())
    return e, a


raw2alpha_nonuni = raw2alpha


def raw2alpha_backward(exp_d, grad_back, interval):
    g = torch.empty_like(exp_d)
    if exp_d.numel() == 0:
        return g
    iv = interval.contiguous().float() if isinstance(interval, torch.Tensor) else None
    _lib.call('pp_raw2alpha_bwd', _f(exp_d.contiguous()), _f(grad_back.contiguous().float()),
              0.0 if iv is not None else float(interval), _f(iv), exp_d.numel(), _f(g), _stream())
    return g


raw2alpha_nonuni_backward = raw2alpha_backward


def maskcache_lookup(world, xyz, xyz2ijk_scale, xyz2ijk_shift):
    """world[X,Y,Z] bool, xyz[n,3] -> bool[n]   (render_utils_kernel.cu:374-424)"""
    w8 = world.contiguous().to(torch.uint8)
    xyz = xyz.contiguous().float()
    out = torch.zeros(xyz.shape[0], dtype=torch.uint8, device=xyz.device)
    s, t = [float(v) for v in xyz2ijk_scale.tolist()], [float(v) for v in xyz2ijk_shift.tolist()]
    _lib.call('pp_maskcache_lookup', _u8(w8), _f(xyz), *[int(v) for v in world.shape], *s, *t, xyz.shape[0], _u8(out),
              _stream())
    return out.bool()


def sample_ndc_pts_on_rays(rays_o, rays_d, xyz_min, xyz_max, N_samples):
    """-> rays_pts[N,S,3], mask_outbbox[N,S]   (render_utils_kernel.cu:245-293)"""
    ro, rd = rays_o.contiguous().float(), rays_d.contiguous().float()
    N = ro.shape[0]
    pts = torch.empty(N, N_samples, 3, device=ro.device)
    mask = torch.empty(N, N_samples, dtype=torch.uint8, device=ro.device)
    sc = _scene_from(xyz_min, xyz_max)
    _lib.call('pp_sample_ndc', ctypes.byref(sc), _f(ro), _f(rd), N, int(N_samples), _f(pts), _u8(mask), _stream())
    return pts, mask.bool()


def sample_bg_pts_on_rays(rays_o, rays_d, t_max, bg_preserve, N_samples):
    """-> rays_pts[N,S,3]   (render_utils_kernel.cu:301-360)"""
    ro, rd = rays_o.contiguous().float(), rays_d.contiguous().float()
    N = ro.shape[0]
    pts = torch.empty(N, N_samples, 3, device=ro.device)
    _lib.call('pp_sample_bg', _f(ro), _f(rd), _f(t_max.contiguous().float()), float(bg_preserve), N, int(N_samples),
              _f(pts), _stream())
    return pts


# ---- adam_upd_cuda (lib/cuda/adam_upd.cpp) --------------------------------------------------------------------
def _adam(param, grad, exp_avg, exp_avg_sq, perlr, step, beta1, beta2, lr, eps, mode):
    for t in (param, grad, exp_avg, exp_avg_sq):
        if not t.is_contiguous():
            raise RuntimeError('adam_upd: tensors must be contiguous')
    _lib.call('pp_adam_upd', _f(param), _f(grad), _f(exp_avg), _f(exp_avg_sq), _f(perlr), param.numel(), int(step),
              float(beta1), float(beta2), float(lr), float(eps), mode, _stream())


def adam_upd_multi(tensors, step, beta1, beta2, lr, eps):
    """adam_upd over a list of (param, grad, exp_avg, exp_avg_sq) tuples of contiguous fp32 CUDA tensors that share step / betas /
    lr / eps (one optimiser group), 32 per launch (pp_adam_upd_multi)."""
    import ctypes
    for i in range(0, len(tensors), 32):
        chunk = tensors[i:i + 32]
        n = len(chunk)
        arr = lambda k: (ctypes.c_void_p * n)(*[c[k].data_ptr() for c in chunk])
        for c in chunk:
            for t in c:
                if not (t.is_cuda and t.is_contiguous() and t.dtype == torch.float32):
                    raise RuntimeError('adam_upd_multi: tensors must be contiguous fp32 CUDA tensors')
        sizes = (ctypes.c_int32 * n)(*[c[0].numel() for c in chunk])
        _lib.call('pp_adam_upd_multi', arr(0), arr(1), arr(2), arr(3), sizes, n, int(step), float(beta1), float(beta2), float(lr),
                  float(eps), _stream())


def adam_upd(param, grad, exp_avg, exp_avg_sq, step, beta1, beta2, lr, eps):
    _adam(param, grad, exp_avg, exp_avg_sq, None, step, beta1, beta2, lr, eps, 0)


def masked_adam_upd(param, grad, exp_avg, exp_avg_sq, step, beta1, beta2, lr, eps):
    _adam(param, grad, exp_avg, exp_avg_sq, None, step, beta1, beta2, lr, eps, 1)


def adam_upd_with_perlr(param, grad, exp_avg, exp_avg_sq, perlr, step, beta1, beta2, lr, eps):
    _adam(param, grad, exp_avg, exp_avg_sq, perlr.contiguous().float(), step, beta1, beta2, lr, eps, 2)


# ---- total_variation_cuda (lib/cuda/total_variation.cpp) ------------------------------------------------------
def total_variation_add_grad(param, grad, wx, wy, wz, dense_mode, mask=None):
    """param/grad (and mask): logical [1,C,X,Y,Z] stored channels_last_3d. In place on grad."""
    from .grid import channels_last_view
    C, (X, Y, Z) = param.shape[1], param.shape[2:]
    _lib.call('pp_tv_add_grad', _f(channels_last_view(param)), _f(channels_last_view(grad)),
              _f(None if mask is None else channels_last_view(mask)), X, Y, Z, C, float(wx), float(wy), float(wz),
              int(bool(dense_mode)), _stream())


total_variation_add_grad_new = total_variation_add_grad


# ---- ub360_utils_cuda (lib/cuda/ub360_utils.cpp) ----------------------------------------------------------------
def cumdist_thres(dist, thres):
    d = dist.contiguous().float()
    mask = torch.zeros(d.shape, dtype=torch.uint8, device=d.device)
    _lib.call('pp_cumdist_thres', _f(d), float(thres), d.shape[0], d.shape[1], _u8(mask), _stream())
    return mask.bool()
