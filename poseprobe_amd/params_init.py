"""Random initialisation with the reference's architecture and init rules, for synthetic runs (no checkpoints
exist offline): cube-init SDF template (lib/voxurf_coarse.py:153-170), nn.Linear default init for rgbnet with zero
last bias (:208-216), kaiming-normal ReLU MLP for the warp net (lib/deformation/modules.py:136-139) whose last layer
is perturbed N(0,1e-2) instead of zero (a zero last layer makes the Jacobian path trivial)."""
import math

import numpy as np
import torch


def cube_sdf(cfg, rect_size):
    lo, hi = np.asarray(cfg.xyz_min, dtype=np.float32), np.asarray(cfg.xyz_max, dtype=np.float32)
    X, Y, Z = cfg.world_size
    lo_t, hi_t = torch.tensor(lo), torch.tensor(hi)
    x, y, z = np.mgrid[lo_t[0].item():hi_t[0].item():X * 1j, lo_t[1].item():hi_t[1].item():Y * 1j,
                       lo_t[2].item():hi_t[2].item():Z * 1j]
    c = ((hi_t + lo_t) / 2).tolist()
    d2 = 0
    inside = np.ones_like(x, dtype=bool)
    for ax, g in enumerate((x, y, z)):
        r = rect_size[ax]
        d = np.minimum(np.abs(g - (r / 2 - c[ax])), np.abs(g - (r / 2 + c[ax])))
        d2 = d2 + d ** 2
        inside &= (g >= (c[ax] - r / 2)) & (g <= (c[ax] + r / 2))
    sdf = torch.from_numpy(d2 ** 0.5)
    sdf[torch.from_numpy(inside)] *= -1
    return sdf.float()[None, None]


def reference_like_params(cfg, seed=0, k0_std=0.1, warp_last_std=1e-2):
    from . import synthetic as syn
    g = torch.Generator().manual_seed(seed)
    X, Y, Z = cfg.world_size
    P = {'sdf': cube_sdf(cfg, syn.range_shape().tolist()),
         'k0': torch.randn(1, cfg.k0_dim, X, Y, Z, generator=g) * k0_std,
         'sdf_alpha': torch.tensor([10.0]), 'sdf_beta': torch.tensor([2.0]), 'rgbnet': [], 'warp': []}
    dim0 = (3 + 6 * cfg.posbase_pe) + (3 + 6 * cfg.viewbase_pe) + cfg.k0_dim + 3
    dims = [dim0, 128, 128, 128, 3]
    for li in range(4):
        b = 1 / math.sqrt(dims[li])
        Wt = (torch.rand(dims[li + 1], dims[li], generator=g) * 2 - 1) * b
        bias = (torch.rand(dims[li + 1], generator=g) * 2 - 1) * b if li < 3 else torch.zeros(3)
        P['rgbnet'].append((Wt, bias))
    wd = [3, 128, 128, 128, 128, 4]
    for li in range(5):
        if li < 4:
            Wt = torch.randn(wd[li + 1], wd[li], generator=g) * math.sqrt(2.0 / wd[li])
            bias = (torch.rand(wd[li + 1], generator=g) * 2 - 1) / math.sqrt(wd[li])
        else:
            Wt = torch.randn(4, 128, generator=g) * warp_last_std
            bias = torch.randn(4, generator=g) * warp_last_std
        P['warp'].append((Wt, bias))
    return P
