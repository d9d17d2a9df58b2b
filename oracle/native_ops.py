"""ctypes binding of oracle/native_ops.c (the C restatement of lib/cuda/render_utils_kernel.cu).

TEST INFRASTRUCTURE ONLY (see the header of native_ops.c).  Operates on CPU torch tensors.
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, '_build', 'libpp_oracle.so')
_lib = None


def build(force=False):
    src = os.path.join(_HERE, 'native_ops.c')
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(_SO), exist_ok=True)
        subprocess.check_call(['gcc', '-O2', '-ffp-contract=off', '-fPIC', '-shared', '-o', _SO, src, '-lm'])
    return _SO


def _load():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.pp_oracle_sample_rays_count.restype = ctypes.c_int64
    return _lib


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def alpha2weight(alpha, ray_id, n_rays):
    """-> weights[M], T[M], alphainv_last[N], i_start[N], i_end[N]   (render_utils_kernel.cu:619-651)"""
    lib = _load()
    alpha = alpha.detach().contiguous().float()
    ray_id = ray_id.contiguous().long()
    M = alpha.numel()
    w = torch.empty(M)
    T = torch.empty(M)
    last = torch.empty(n_rays)
    i_s = torch.empty(n_rays, dtype=torch.int64)
    i_e = torch.empty(n_rays, dtype=torch.int64)
    lib.pp_oracle_alpha2weight(_p(alpha), _p(ray_id), ctypes.c_int64(M), ctypes.c_int64(n_rays),
                               _p(w), _p(T), _p(last), _p(i_s), _p(i_e))
    return w, T, last, i_s, i_e


def alpha2weight_backward(alpha, weight, T, alphainv_last, i_start, i_end, n_rays, grad_weights, grad_last):
    """-> grad_alpha[M]   (render_utils_kernel.cu:654-707)"""
    lib = _load()
    M = alpha.numel()
    g = torch.empty(M)
    args = [t.detach().contiguous() for t in (alpha, weight, T, alphainv_last, i_start, i_end)]
    gw = grad_weights.detach().contiguous().float()
    gl = grad_last.detach().contiguous().float()
    lib.pp_oracle_alpha2weight_backward(*[_p(t) for t in args], ctypes.c_int64(M), ctypes.c_int64(n_rays),
                                        _p(gw), _p(gl), _p(g))
    return g


def sample_pts_on_rays(rays_o, rays_d, xyz_min, xyz_max, near, far, stepdist):
    """-> rays_pts, mask_outbbox, ray_id, step_id, N_steps, t_min, t_max, rays_start, rays_dir
    (render_utils_kernel.cu:196-242)"""
    lib = _load()
    rays_o = rays_o.detach().contiguous().float()
    rays_d = rays_d.detach().contiguous().float()
    xyz_min = xyz_min.detach().contiguous().float()
    xyz_max = xyz_max.detach().contiguous().float()
    N = rays_o.shape[0]
    t_min, t_max = torch.empty(N), torch.empty(N)
    n_steps = torch.empty(N, dtype=torch.int64)
    start, dirs = torch.empty(N, 3), torch.empty(N, 3)
    f = ctypes.c_float
    total = lib.pp_oracle_sample_rays_count(_p(rays_o), _p(rays_d), _p(xyz_min), _p(xyz_max), f(float(near)),
                                            f(float(far)), f(float(stepdist)), ctypes.c_int64(N), _p(t_min),
                                            _p(t_max), _p(n_steps), _p(start), _p(dirs))
    pts = torch.empty(total, 3)
    mask = torch.empty(total, dtype=torch.uint8)
    ray_id = torch.empty(total, dtype=torch.int64)
    step_id = torch.empty(total, dtype=torch.int64)
    lib.pp_oracle_sample_rays_fill(_p(start), _p(dirs), _p(xyz_min), _p(xyz_max), _p(n_steps), f(float(stepdist)),
                                   ctypes.c_int64(N), _p(pts), _p(mask), _p(ray_id), _p(step_id))
    return pts, mask.bool(), ray_id, step_id, n_steps, t_min, t_max, start, dirs
