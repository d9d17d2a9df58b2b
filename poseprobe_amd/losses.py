"""object_losses with the reference's signature (lib/losses.py:6-74) for the drop-in autograd path.

These are O(N)+O(M) reductions on the render outputs; the fused train step (engine.TrainEngine) uses the HIP loss
kernels (pp_loss_rays / pp_loss_samples) instead and never calls this module.
"""
import math

import torch
import torch.nn.functional as F


class _AttrDict(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def deform_implicit_loss(model_output, use_deform):
    g = model_output['gradient']
    d = {'grad_constraint': torch.abs(g.norm(dim=-1) - 1).mean()}
    if use_deform:
        d.update({'grad_deform_constraint': model_output['grad_deform'].norm(dim=-1).mean(),
                  'sdf_correct_constraint': torch.abs(model_output['sdf_correct']).mean(),
                  'sdf_deform_constraint': torch.abs(model_output['sdf_deform']).mean()})
    return d


def rendering_loss(rgb_marched, target, mask):
    return F.mse_loss(rgb_marched * mask, target * mask, reduction='sum') / (mask.sum() * 3)


def dynamic_weight(initial_weight, final_weight, iteration, total_iterations):
    return initial_weight * math.exp(math.log(final_weight / initial_weight) / total_iterations * iteration)


def object_losses(model_output, cfg_train, target, mask, iteration, total_iterations, use_deform):
    S, Wt = _AttrDict(), _AttrDict()
    S.img_render = rendering_loss(model_output['rgb_marched'], target, mask)
    Wt.img_render = cfg_train.weight_main
    pout = model_output['alphainv_cum'].clamp(1e-6, 1 - 1e-6)
    S.weight_entropy_last = -(pout * torch.log(pout) + (1 - pout) * torch.log(1 - pout)).mean()
    Wt.weight_entropy_last = 0.01
    if cfg_train.weight_tv_k0 > 0:
        S.tv_k0 = model_output['k0_tv']
        Wt.tv_k0 = cfg_train.weight_tv_k0
    imp = deform_implicit_loss(model_output, use_deform)
    S.grad_constraint = imp['grad_constraint']
    Wt.grad_constraint = 1.0
    if use_deform:
        w = dynamic_weight(1e-1, 1e-3, iteration, total_iterations)
        for k in ('grad_deform_constraint', 'sdf_correct_constraint', 'sdf_deform_constraint'):
            S[k] = imp[k]
            Wt[k] = w
    S.mask_render = F.binary_cross_entropy(model_output['cum_weights'].clip(1e-3, 1.0 - 1e-3), mask)
    Wt.mask_render = cfg_train.weight_mask
    loss = 0
    for k, v in S.items():
        loss = loss + v * Wt[k]
    return S, Wt, loss
