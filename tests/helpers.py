"""Shared helpers for the parity tests: load golden fixtures into oracle structures."""
import os

import numpy as np
import torch

from oracle import voxurf_oracle as O
from poseprobe_amd import synthetic as syn

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


def scene_for(G, **kw):
    rs = syn.range_shape()
    return O.Scene(syn.XYZ_MIN, syn.XYZ_MAX, int(G) ** 3, output_range=float(rs.max()), rect_size=rs.tolist(), **kw)


def params_from_npz(d):
    P = {'k0': torch.tensor(d['P.k0']), 'sdf': torch.tensor(d['P.sdf']),
         'sdf_alpha': torch.tensor(d['P.sdf_alpha']), 'sdf_beta': torch.tensor(d['P.sdf_beta']),
         'rgbnet': [], 'warp': []}
    for li in range(4):
        P['rgbnet'].append((torch.tensor(d[f'P.rgbnet.{li}.weight']), torch.tensor(d[f'P.rgbnet.{li}.bias'])))
    for li in range(5):
        P['warp'].append((torch.tensor(d[f'P.warp.{li}.weight']), torch.tensor(d[f'P.warp.{li}.bias'])))
    return P


def oracle_step_from_golden(d):
    """Re-run the oracle on a forward_* fixture's inputs: returns (out, S, loss, P, se3)."""
    G = int(d['G'])
    scene = scene_for(G)
    P = O.params_require_grad(params_from_npz(d))
    se3 = torch.tensor(d['se3'], requires_grad=True)
    w2c = O.current_pose_pnp(se3, torch.tensor(d['w2c_init']))
    c2w = O.pose_invert(w2c)
    ro, rd, vd, target, mask = O.select_training_rays(torch.tensor(d['ray_idx']), torch.tensor(d['images']),
                                                      torch.tensor(d['masks']), torch.tensor(d['Ks']), c2w)
    gs = int(d['global_step'])
    out = O.voxurf_forward(P, scene, ro, rd, vd, jitter=torch.tensor(d['jitter']), global_step=gs)
    S, Wt, loss = O.object_losses(out, target, mask, gs, scene.N_iters)
    (loss * 0.1).backward()
    return out, S, loss, P, se3, dict(rays_o=ro, rays_d=rd, viewdirs=vd, target=target, mask=mask, c2w=c2w, w2c=w2c)


def assert_close(a, b, rtol=1e-5, atol=1e-6, name='', scaled=0.0):
    """|a-b| <= atol + rtol*|b| + scaled*max|b|.  `scaled` expresses an error budget relative to the
    largest magnitude in the tensor (fp32 sums of large terms leave absolute errors on small entries)."""
    a = np.asarray(a.detach().cpu() if isinstance(a, torch.Tensor) else a, dtype=np.float64)
    b = np.asarray(b.detach().cpu() if isinstance(b, torch.Tensor) else b, dtype=np.float64)
    assert a.shape == b.shape, f'{name}: shape {a.shape} vs {b.shape}'
    if a.size == 0:
        return
    err = np.abs(a - b)
    tol = atol + rtol * np.abs(b) + scaled * np.abs(b).max()
    bad = err > tol
    assert not bad.any(), (f'{name}: {bad.sum()}/{a.size} mismatches, max abs err {err.max():.3e}, '
                           f'max |ref| {np.abs(b).max():.3e}')


def assert_mostly_close(a, b, rtol, scaled, name='', outlier_frac=0.03, outlier_scaled=3e-2):
    """For gradients of ReLU networks compared across two fp32 implementations: a pre-activation that lies within rounding
    distance of zero flips its mask in one of them and changes that sample's contribution discretely (a handful of the ~1e7
    activations of a test do).  Rows (first dimension) are therefore allowed to miss the tight tolerance in at most
    `outlier_frac` of the cases, and every element has to meet the loose bound `outlier_scaled` * max|b|."""
    a = np.asarray(a.detach().cpu() if isinstance(a, torch.Tensor) else a, dtype=np.float64)
    b = np.asarray(b.detach().cpu() if isinstance(b, torch.Tensor) else b, dtype=np.float64)
    assert a.shape == b.shape, f'{name}: shape {a.shape} vs {b.shape}'
    err = np.abs(a - b)
    mx = np.abs(b).max()
    loose = err > outlier_scaled * mx + rtol * np.abs(b)
    assert not loose.any(), f'{name}: {loose.sum()}/{a.size} beyond the loose bound, max abs err {err.max():.3e}, max |ref| {mx:.3e}'
    bad = (err > rtol * np.abs(b) + scaled * mx).reshape(a.shape[0], -1).any(axis=1)
    assert bad.mean() <= outlier_frac, (f'{name}: {bad.sum()}/{bad.size} rows miss the tight tolerance '
                                        f'(max abs err {err.max():.3e}, max |ref| {mx:.3e})')


def assert_close_but(a, b, rtol, atol, name, frac=1e-3, loose_atol=5e-4):
    """Per-sample quantities at the large grids: all entries within `loose_atol`, and all but a fraction `frac` within the
    stated tight tolerance (the NeuS alpha is a quotient of sigmoid differences; where sigma(prev / s) is tiny, fp32
    cancellation leaves 1e-4-level absolute noise on a handful of the ~1e5 samples, in the reference's own fp32 as well)."""
    a = np.asarray(a.detach().cpu() if isinstance(a, torch.Tensor) else a, dtype=np.float64)
    b = np.asarray(b.detach().cpu() if isinstance(b, torch.Tensor) else b, dtype=np.float64)
    assert a.shape == b.shape, f'{name}: shape {a.shape} vs {b.shape}'
    err = np.abs(a - b)
    assert err.max() <= loose_atol + rtol * np.abs(b).max(), f'{name}: max abs err {err.max():.3e}'
    bad = err > atol + rtol * np.abs(b)
    assert bad.mean() <= frac, f'{name}: {bad.sum()}/{a.size} outside rtol {rtol} / atol {atol} (max abs err {err.max():.3e})'
