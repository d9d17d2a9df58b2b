"""CPU restatement of the scene branch's render path (TEST INFRASTRUCTURE ONLY - never imported by the product).

Follows the reference's lib/bg_nerf/source/models/frequency_nerf.py:
  * positional encoding (:42-69) with BARF's coarse-to-fine band weights (:239-266),
  * 8 x 256 feature MLP with the skip at layer 4 and the density channel in row 0 of the last layer (:152-170),
  * softplus density, view-dependent colour head 283 -> 128 -> 3 + sigmoid (:189-227),
  * quadrature compositing with exp(-cumsum) transmittance (:290-343),
and the photometric loss 2 * huber(delta = 0.5) of training/core/base_losses.py:155-156, :304-305.

Pinned by tests/golden/scene_b2.npz, which oracle/make_golden.py produced by executing the reference itself
(tests/test_oracle_vs_golden.py).  Plain torch on the CPU; gradients come from autograd over this restatement.
"""
import math

import torch
import torch.nn.functional as F

L_3D, L_VIEW = 10, 4
N_FEAT_LAYERS, SKIP = 8, (4,)


def band_weights(progress, barf_c2f, L, dtype=torch.float32):
    """Raised-cosine window over the frequency bands; all ones without a schedule."""
    if barf_c2f is None:
        return torch.ones(L, dtype=dtype)
    start, end = barf_c2f
    a = (progress - start) / (end - start) * L
    k = torch.arange(L, dtype=dtype)
    return (1 - torch.cos(math.pi * (a - k).clamp(0, 1))) / 2


def encode(x, L, w):
    """x [..., 3] -> [..., 3 + 6L] = (x, per coordinate: L sines then L cosines), band l scaled by w[l]."""
    freq = (2.0 ** torch.arange(L, dtype=x.dtype)) * math.pi
    ph = x[..., None] * freq                                        # [..., 3, L]
    enc = torch.stack([ph.sin() * w, ph.cos() * w], dim=-2)         # [..., 3, 2, L]
    return torch.cat([x, enc.flatten(-3)], dim=-1)


def mlp(params, pts, ray, progress, barf_c2f, masks=None, hidden=None):
    """pts [R, S, 3], ray [R, 3] -> rgb_samples [R, S, 3], density [R, S].
    Test instrumentation (never used by the fixtures): `hidden` (a list) collects the nine post-ReLU activations; `masks`
    (nine boolean tensors) REPLACES every ReLU by a multiplication with the given 0 / 1 pattern, i.e. evaluates the network
    on the linear piece another implementation chose - two fp32 implementations differ in the ReLU state of the few
    pre-activations that lie within rounding distance of zero, and on a common piece their gradients agree tightly."""
    act = (lambda x, i: F.relu(x)) if masks is None else (lambda x, i: x * masks[i].reshape(x.shape).to(x.dtype))
    e = encode(pts, L_3D, band_weights(progress, barf_c2f, L_3D, pts.dtype))
    h = e
    for li in range(N_FEAT_LAYERS):
        if li in SKIP:
            h = torch.cat([h, e], -1)
        h = F.linear(h, params[f'mlp_feat.{li}.weight'], params[f'mlp_feat.{li}.bias'])
        if li == N_FEAT_LAYERS - 1:
            raw, h = h[..., 0], h[..., 1:]
        h = act(h, li)
        if hidden is not None:
            hidden.append(h)
    density = F.softplus(raw)
    unit = ray / ray.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    ve = encode(unit, L_VIEW, band_weights(progress, barf_c2f, L_VIEW, pts.dtype))
    h = torch.cat([h, ve[:, None, :].expand(*h.shape[:-1], ve.shape[-1])], -1)
    h = act(F.linear(h, params['mlp_rgb.0.weight'], params['mlp_rgb.0.bias']), N_FEAT_LAYERS)
    if hidden is not None:
        hidden.append(h)
    rgb = torch.sigmoid(F.linear(h, params['mlp_rgb.1.weight'], params['mlp_rgb.1.bias']))
    return rgb, density


def composite(rgb_s, density, depth, ray, white_bg=False):
    """depth [R, S]; returns rgb [R,3], depth [R], opacity [R], weights [R,S], all_cumulated [R], rgb_var [R], depth_var [R]."""
    intv = torch.cat([depth[:, 1:] - depth[:, :-1], torch.full_like(depth[:, :1], 1e10)], 1)
    sd = density * intv * ray.norm(dim=-1, keepdim=True)
    alpha = 1 - torch.exp(-sd)
    T = torch.exp(-torch.cat([torch.zeros_like(sd[:, :1]), sd[:, :-1]], 1).cumsum(1))
    w = T * alpha
    d = (depth * w).sum(1)
    rgb = (rgb_s * w[..., None]).sum(1)
    out = dict(weights=w, depth=d, opacity=w.sum(1), all_cumulated=T[:, -2],
               depth_var=(w * (depth - d[:, None]) ** 2).sum(1),
               rgb_var=((rgb_s - rgb[:, None]).sum(-1) * w).sum(1))
    out['rgb'] = rgb + (1 - out['opacity'])[:, None] if white_bg else rgb
    return out


def render(params, center, ray, depth, progress, barf_c2f, white_bg=False, masks=None, hidden=None):
    """center, ray [R, 3]; depth [R, S]."""
    pts = center[:, None] + ray[:, None] * depth[..., None]
    rgb_s, dens = mlp(params, pts, ray, progress, barf_c2f, masks, hidden)
    out = composite(rgb_s, dens, depth, ray, white_bg)
    out.update(rgb_samples=rgb_s, density_samples=dens)
    return out


def photometric_loss(rgb, image):
    return 2. * F.huber_loss(rgb, image, reduction='mean', delta=0.5)
