"""Optimiser / checkpoint helpers with the reference's surface (lib/utils.py:53-198 `Adam`, :316-362
`create_optimizer_or_freeze_model`, `create_optimizer_pose`, :416-438 `load_model`, `mse2psnr`).

`Adam.step()` runs the HIP kernel pp_adam_upd (plain / per-voxel-lr form) on every parameter; parameters stored
channels-last (the k0 grid) are updated in their physical layout, moments share that layout.
"""
import numpy as np
import torch
import torch.nn as nn

from . import render_utils

mse2psnr = lambda x: -10. * torch.log10(x)


class Adam(torch.optim.Optimizer):
    """Extended Adam with optional per-voxel learning rate (lib/utils.py:53-198)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        if weight_decay != 0 or amsgrad:
            raise NotImplementedError('weight_decay / amsgrad are never used by the reference configs')
        for name, v, ok in (('learning rate', lr, lr >= 0.0), ('epsilon', eps, eps >= 0.0),
                            ('beta1', betas[0], 0.0 <= betas[0] < 1.0), ('beta2', betas[1], 0.0 <= betas[1] < 1.0)):
            if not ok:
                raise ValueError(f'Invalid {name}: {v}')
        self.per_lr = None
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad))

    def set_pervoxel_lr(self, count):
        assert self.param_groups[0]['params'][0].shape == count.shape
        self.per_lr = count.float() / count.max()

    @staticmethod
    def _dense(t):
        """(tensor usable as a flat buffer, is_view): channels-last 5-D tensors are dense in their own layout."""
        if t.is_contiguous():
            return t
        if t.dim() == 5 and t.is_contiguous(memory_format=torch.channels_last_3d):
            return t.permute(0, 2, 3, 4, 1)
        return None

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            beta1, beta2 = group['betas']
            for p in group['params']:
                if p.grad is None:
                    continue
                if not p.is_cuda:
                    raise RuntimeError('poseprobe_amd.utils.Adam runs on the HIP path only (CUDA parameters)')
                state = self.state[p]
                if len(state) == 0:
                    state['step'] = 0
                    state['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    state['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                state['step'] += 1
                pd, md, vd = self._dense(p.data), self._dense(state['exp_avg']), self._dense(state['exp_avg_sq'])
                g = p.grad
                if pd is None or md is None or vd is None:
                    raise RuntimeError('Adam: parameter / state tensors must be dense')
                gd = self._dense(g)
                if gd is None or gd.stride() != pd.stride():
                    g = g.contiguous(memory_format=torch.channels_last_3d) if pd.dim() == 5 and not p.is_contiguous() else g.contiguous()
                    gd = self._dense(g)
                # torch's Adam divides sqrt(v) by sqrt(bias_correction2) BEFORE adding eps (lib/utils.py:184-188), the
                # kernel folds sqrt(bc2) into the step size like lib/cuda/adam_upd.cpp:72 -> scale eps accordingly
                bc2 = 1 - beta2 ** state['step']
                eps = group['eps'] * np.sqrt(bc2)
                if self.per_lr is not None and p.shape == self.per_lr.shape:
                    per = self.per_lr.to(p.device)
                    per = per.contiguous(memory_format=torch.channels_last_3d).permute(0, 2, 3, 4, 1) if pd.dim() == 5 and not p.is_contiguous() else per.contiguous()
                    render_utils.adam_upd_with_perlr(pd, gd, md, vd, per, state['step'], beta1, beta2, group['lr'], eps)
                else:
                    render_utils.adam_upd(pd, gd, md, vd, state['step'], beta1, beta2, group['lr'], eps)
        return loss


def create_optimizer_or_freeze_model(model, cfg_train, global_step):
    """lib/utils.py:316-342: one param group per `lrate_<attr>` key, lr decayed to `global_step`, betas (0.9, 0.99)."""
    decay_steps = cfg_train.lrate_decay * 1000
    decay_factor = 0.1 ** (global_step / decay_steps)
    param_group = []
    for k in cfg_train.keys():
        if not k.startswith('lrate_'):
            continue
        k = k[len('lrate_'):]
        if not hasattr(model, k):
            continue
        param = getattr(model, k)
        if param is None:
            continue
        lr = getattr(cfg_train, f'lrate_{k}') * decay_factor
        if lr > 0:
            if isinstance(param, nn.Module):
                param = param.parameters()
            param_group.append({'params': param, 'lr': lr, 'name': k})
        else:
            if isinstance(param, nn.Module):
                for q in param.parameters():
                    q.requires_grad = False
            else:
                param.requires_grad = False
    return Adam(param_group, betas=(0.9, 0.99))


def create_optimizer_pose(model, cfg_train, max_iter=1, align=False, index=None):
    """lib/utils.py:347-362"""
    if align is False:
        optim_pose = Adam([dict(params=model.se3_refine, lr=cfg_train.lr_pose)])
    else:
        optim_pose = Adam([dict(params=model.se3_align_refine, lr=1e-5)])
    if cfg_train.sched_pose is None:
        return optim_pose, None
    assert cfg_train.sched_pose == 'ExponentialLR'
    gamma = (cfg_train.lr_pose_end / (1e-10 + cfg_train.lr_pose)) ** (1. / max_iter)
    return optim_pose, torch.optim.lr_scheduler.ExponentialLR(optim_pose, gamma=gamma)


def load_model(model_class, ckpt_path, strict=True, device='cuda'):
    """lib/utils.py:416-438: rebuild from `model_kwargs` and load `model_state_dict` (files written by this package;
    reference checkpoints contain pickled numpy arrays in `model_kwargs` and need weights_only=False on the user's side)."""
    ckpt = torch.load(ckpt_path, map_location='cpu', weights_only=False)
    model = model_class(**ckpt['model_kwargs'])
    sd = ckpt['model_state_dict']
    sd.pop('s_val', None)        # only present when the reference model was built on CPU (voxurf_coarse.py:94)
    model.load_state_dict(sd, strict=strict)
    return model.to(device)
