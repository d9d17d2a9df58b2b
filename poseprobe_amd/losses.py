"""object_losses with the reference's signature (lib/losses.py:6-74) for the drop-in autograd path.

These are O(N)+O(M) reductions on the render outputs.  On CUDA tensors `object_losses` evaluates them and the gradient of their
weighted sum with the two HIP loss kernels the fused train step uses (pp_loss_rays / pp_loss_samples): 6 launches instead of ~40.
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


def _object_losses_torch(model_output, cfg_train, target, mask, iteration, total_iterations, use_deform):
    """The reference's expressions, op by op (lib/losses.py:34-74): ~40 small torch launches; CPU tensors, use_deform=False and
    callers that differentiate individual loss scalars take this path."""
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


_KERNEL_SCALARS = ('img_render', 'weight_entropy_last', 'grad_constraint', 'grad_deform_constraint', 'sdf_correct_constraint',
                   'sdf_deform_constraint', 'mask_render')          # order of pp_loss_rays / pp_loss_samples' loss_out[0..6]


class _FusedObjectLosses(torch.autograd.Function):
    """The seven ray- / sample-level terms of object_losses as TWO kernels (pp_loss_rays, pp_loss_samples - the ones the fused
    engine step uses): values AND the gradient of their weighted sum in one pass.  Outputs: the weighted sum (differentiable) and
    the seven unweighted scalars.  The common case - only the total is back-propagated, as lib/recon_scene.py:649 does - costs one
    scaling launch in backward; when a caller differentiates an individual scalar the backward re-evaluates the reference's
    expressions through torch autograd instead (same result, the slow way)."""

    @staticmethod
    def forward(ctx, rgb_marched, alphainv_cum, cum_weights, gradient, grad_deform, sdf_correct, sdf_deform, target, mask, weights):
        from . import ops
        w_main, w_ent, w_mask, w_eik, w_dyn = weights
        dev = rgb_marched.device
        N, M = rgb_marched.shape[0], gradient.shape[0]
        f = dict(dtype=torch.float32, device=dev)
        c = lambda t: t.detach().contiguous().float()
        # ONE buffer for every gradient, so that backward scales them all with one launch
        sizes = (N * 3, N, N, M * 3, M * 9, M, M)
        flat = torch.zeros(sum(sizes) + 16, **f)                   # tail: loss_out[8] | mask_sum[1]
        views, o = [], 0
        for n in sizes:
            views.append(flat[o:o + n]); o += n
        loss_out, mask_sum = flat[o:o + 8], flat[o + 8:o + 9]
        ops.loss_rays(c(rgb_marched), c(alphainv_cum), c(cum_weights).reshape(-1), c(target), c(mask).reshape(-1), mask_sum, w_main,
                      w_ent, w_mask, 1.0, views[0].view(N, 3), views[1], views[2], loss_out, None)
        if M > 0:
            wo = torch.empty(M, 16, **f)                           # the kernel reads the correction at column 3 of a warp_out row
            wo[:, 3] = sdf_correct.detach().reshape(M)
            count = torch.full((1,), M, dtype=torch.int32, device=dev)
            ops.loss_samples(c(gradient), c(grad_deform).reshape(M, 9), wo, c(sdf_deform), count, M, w_eik, w_dyn, 1.0,
                             views[3].view(M, 3), views[4].view(M, 9), views[5], views[6], loss_out, None)
        scalars = loss_out[:7].clone()
        total = loss_out[7].clone()                                # the kernels' own weighted sum (no host round trip for the weights)
        ctx.flat, ctx.sizes, ctx.shapes = flat, sizes, (rgb_marched.shape, alphainv_cum.shape, cum_weights.shape, gradient.shape,
                                                         grad_deform.shape, sdf_correct.shape, sdf_deform.shape)
        ctx.weights = weights
        ctx.save_for_backward(rgb_marched, alphainv_cum, cum_weights, gradient, grad_deform, sdf_correct, sdf_deform, target, mask)
        ctx.set_materialize_grads(False)
        return total, scalars

    @staticmethod
    def backward(ctx, g_total, g_scalars):
        if g_scalars is not None:
            return _FusedObjectLosses._backward_torch(ctx, g_total, g_scalars)
        if g_total is None:
            return (None,) * 10
        n = sum(ctx.sizes)
        scaled = ctx.flat[:n] * g_total
        out, o = [], 0
        for size, shape in zip(ctx.sizes, ctx.shapes):
            out.append(scaled[o:o + size].view(shape)); o += size
        return (*out, None, None, None)

    @staticmethod
    def _backward_torch(ctx, g_total, g_scalars):
        saved = ctx.saved_tensors
        ins = [t.detach().requires_grad_(True) for t in saved[:7]]
        target, mask = saved[7], saved[8]
        w_main, w_ent, w_mask, w_eik, w_dyn = ctx.weights
        with torch.enable_grad():
            mo = dict(rgb_marched=ins[0], alphainv_cum=ins[1], cum_weights=ins[2], gradient=ins[3], grad_deform=ins[4],
                      sdf_correct=ins[5], sdf_deform=ins[6])
            cfg = _AttrDict(weight_main=w_main, weight_tv_k0=0.0, weight_mask=w_mask)
            S, _, _ = _object_losses_torch(mo, cfg, target, mask, 0, 1, True)
            vec = torch.stack([S[k] for k in _KERNEL_SCALARS])
            wvec = torch.tensor([w_main, w_ent, w_eik, w_dyn, w_dyn, w_dyn, w_mask], device=vec.device)
            eff = g_scalars + (wvec * g_total if g_total is not None else 0.0)
            grads = torch.autograd.grad((vec * eff).sum(), ins, allow_unused=True)
        return (*grads, None, None, None)


def object_losses(model_output, cfg_train, target, mask, iteration, total_iterations, use_deform):
    """lib/losses.py:34-74: (loss_scalars, loss_weight, loss).  CUDA inputs with use_deform take the two-kernel path
    (_FusedObjectLosses); everything else the op-by-op torch expressions."""
    rgb = model_output['rgb_marched']
    if not (use_deform and rgb.is_cuda and all(k in model_output for k in ('grad_deform', 'sdf_correct', 'sdf_deform'))):
        return _object_losses_torch(model_output, cfg_train, target, mask, iteration, total_iterations, use_deform)
    w_dyn = dynamic_weight(1e-1, 1e-3, iteration, total_iterations)
    weights = (float(cfg_train.weight_main), 0.01, float(cfg_train.weight_mask), 1.0, float(w_dyn))
    total, scalars = _FusedObjectLosses.apply(rgb, model_output['alphainv_cum'], model_output['cum_weights'], model_output['gradient'],
                                              model_output['grad_deform'], model_output['sdf_correct'], model_output['sdf_deform'],
                                              target, mask, weights)
    S, Wt = _AttrDict(), _AttrDict()
    vals = dict(zip(_KERNEL_SCALARS, scalars.unbind(0)))
    S.img_render, Wt.img_render = vals['img_render'], cfg_train.weight_main
    S.weight_entropy_last, Wt.weight_entropy_last = vals['weight_entropy_last'], 0.01
    loss = total
    if cfg_train.weight_tv_k0 > 0:
        S.tv_k0, Wt.tv_k0 = model_output['k0_tv'], cfg_train.weight_tv_k0
        loss = loss + S.tv_k0 * Wt.tv_k0
    S.grad_constraint, Wt.grad_constraint = vals['grad_constraint'], 1.0
    for k in ('grad_deform_constraint', 'sdf_correct_constraint', 'sdf_deform_constraint'):
        S[k], Wt[k] = vals[k], w_dyn
    S.mask_render, Wt.mask_render = vals['mask_render'], cfg_train.weight_mask
    return S, Wt, loss
