"""numpy restatements of the reference's DVGO-surface native operators (TEST INFRASTRUCTURE ONLY).

Followed from the .cu text: lib/cuda/render_utils_kernel.cu:245-360 (NDC / background samplers), :374-424 (maskcache
lookup), :431-574 (raw2alpha*), lib/cuda/adam_upd_kernel.cu:8-133, lib/cuda/total_variation_kernel.cu:13-134,
lib/cuda/ub360_utils_kernel.cu:12-48.  None of these kernels is exercised by the reference's live loop and the
reference ships no tests for them: "parity unpinned" beyond the source text."""
import numpy as np

f32 = np.float32


def raw2alpha(density, shift, interval):
    e = np.exp((density + f32(shift)).astype(f32)).astype(f32)
    iv = np.asarray(interval, dtype=f32)
    with np.errstate(over='ignore'):
        a = (f32(1) - np.power(f32(1) + e, -iv)).astype(f32)
    return e, a


def raw2alpha_backward(exp_d, grad_back, interval):
    iv = np.asarray(interval, dtype=f32)
    with np.errstate(over='ignore', invalid='ignore'):
        return (np.minimum(exp_d, f32(1e10)) * np.power(f32(1) + exp_d, -iv - f32(1)) * iv * grad_back).astype(f32)


def maskcache_lookup(world, xyz, scale, shift):
    ijk = np.rint(xyz * scale + shift).astype(np.int64)       # round-half-away in C, rint here: inputs avoid .5 ties
    ok = np.all((ijk >= 0) & (ijk < np.array(world.shape)), axis=1)
    out = np.zeros(len(xyz), dtype=bool)
    out[ok] = world[ijk[ok, 0], ijk[ok, 1], ijk[ok, 2]]
    return out


def sample_ndc(rays_o, rays_d, xyz_min, xyz_max, S):
    dist = (np.arange(S, dtype=f32) / f32(S - 1)).astype(f32)
    pts = rays_o[:, None, :] + rays_d[:, None, :] * dist[None, :, None]
    mask = ((xyz_min > pts) | (xyz_max < pts)).any(-1)
    return pts.astype(f32), mask


def sample_bg(rays_o, rays_d, t_max, bg_preserve, S):
    step = (np.arange(S, dtype=f32) / f32(S)).astype(np.float64)
    t_outer0 = (t_max.astype(np.float64)[:, None] - 1. + 1. / (1. - step)[None]).astype(f32)
    p = (rays_o[:, None, :] + rays_d[:, None, :] * t_outer0[..., None]).astype(f32)
    t_outer = np.sqrt((p * p).sum(-1)).astype(f32)
    R = t_outer / np.abs(p).max(-1)
    o2i = ((R * R / (t_outer * t_outer)).astype(np.float64) * (1. - bg_preserve) + (R / t_outer * f32(bg_preserve))).astype(f32)
    return (p * o2i[..., None]).astype(f32)


def adam_upd(p, g, m, v, step, b1, b2, lr, eps, mode=0, perlr=None):
    step_size = f32(lr) * np.sqrt(f32(1) - np.power(f32(b2), f32(step))) / (f32(1) - np.power(f32(b1), f32(step)))
    sel = (g != 0) if mode == 1 else np.ones_like(g, dtype=bool)
    m2 = np.where(sel, f32(b1) * m + (f32(1) - f32(b1)) * g, m).astype(f32)
    v2 = np.where(sel, f32(b2) * v + (f32(1) - f32(b2)) * g * g, v).astype(f32)
    lr_e = step_size * (perlr if mode == 2 else f32(1))
    p2 = np.where(sel, p - lr_e * m2 / (np.sqrt(v2) + f32(eps)), p).astype(f32)
    return p2, m2, v2


def tv_add_grad(param, grad, wx, wy, wz, dense_mode, mask=None):
    """param/grad/mask: [1,C,X,Y,Z]. Returns the new grad. Quirks of the two reference kernels kept."""
    P = param[0]
    add = np.zeros_like(P)
    M = np.ones_like(P) if mask is None else mask[0]
    w_last = wz if mask is None else wx          # unmasked kernel: wz on the last axis (wx unused)
    for ax, w in ((3, w_last), (2, wy), (1, wz)):
        d = np.clip(np.diff(P, axis=ax), -1, 1)               # P[i+1]-P[i]
        mm = np.take(M, range(1, P.shape[ax]), axis=ax) * np.take(M, range(0, P.shape[ax] - 1), axis=ax)
        hi = [slice(None)] * 4
        lo = [slice(None)] * 4
        hi[ax] = slice(1, None)
        lo[ax] = slice(0, -1)
        add[tuple(hi)] += w * d * mm
        add[tuple(lo)] -= w * d * mm
    sel = np.ones_like(P, dtype=bool) if dense_mode else (grad[0] != 0)
    return (grad[0] + np.where(sel, add, 0))[None].astype(f32)


def cumdist_thres(dist, thres):
    mask = np.zeros(dist.shape, dtype=bool)
    for r in range(dist.shape[0]):
        cum = f32(0)
        for i in range(dist.shape[1]):
            cum = f32(cum + dist[r, i])
            over = cum > f32(thres)
            if over:
                cum = f32(0)
            mask[r, i] = over
    return mask
