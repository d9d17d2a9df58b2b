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
            small = {}                                  # step -> [(p, g, m, v)]: small dense tensors of the group go out in ONE launch
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
                elif pd.numel() <= (1 << 20) and pd.is_contiguous() and gd.is_contiguous() and pd.dtype == torch.float32 and gd.dtype == torch.float32:
                    small.setdefault(state['step'], []).append((pd, gd, md, vd))
                else:
                    render_utils.adam_upd(pd, gd, md, vd, state['step'], beta1, beta2, group['lr'], eps)
            for step, tensors in small.items():         # same arithmetic as adam_upd, up to 32 tensors per launch
                render_utils.adam_upd_multi(tensors, step, beta1, beta2, group['lr'], group['eps'] * np.sqrt(1 - beta2 ** step))
        return loss


def _optimised_attributes(model, cfg_train):
    """(name, attribute, base lr) for every `lrate_<name>` entry of the training config that names an existing,
    non-None attribute of the model - the coupling the reference's optimiser builder relies on."""
    rows = []
    for key in cfg_train.keys():
        name = key[6:] if key.startswith('lrate_') else None
        attr = getattr(model, name, None) if name else None
        if attr is not None:
            rows.append((name, attr, getattr(cfg_train, key)))
    return rows


def create_optimizer_or_freeze_model(model, cfg_train, global_step):
    """Counterpart of lib/utils.py:316-342.  One Adam group (betas 0.9 / 0.99) per trainable attribute, its learning rate
    already decayed to `global_step` (factor 0.1 per `lrate_decay` thousand steps); attributes whose rate is not positive
    are frozen instead of grouped."""
    decayed = 0.1 ** (global_step / (cfg_train.lrate_decay * 1000))
    groups = []
    for name, attr, base_lr in _optimised_attributes(model, cfg_train):
        tensors = list(attr.parameters()) if isinstance(attr, nn.Module) else [attr]
        if base_lr * decayed > 0:
            groups.append(dict(params=tensors, lr=base_lr * decayed, name=name))
        else:
            for t in tensors:
                t.requires_grad = False
    return Adam(groups, betas=(0.9, 0.99))


def create_optimizer_pose(model, cfg_train, max_iter=1, align=False, index=None):
    """Counterpart of lib/utils.py:347-362: Adam on the 6-DoF refinement (or, with align=True, on the alignment refinement
    at 1e-5) and, when the config asks for it, an exponential schedule that reaches `lr_pose_end` after `max_iter` steps."""
    target, lr = (model.se3_align_refine, 1e-5) if align else (model.se3_refine, cfg_train.lr_pose)
    optim = Adam([dict(params=target, lr=lr)])
    kind = cfg_train.sched_pose
    if kind is None:
        return optim, None
    if kind != 'ExponentialLR':
        raise ValueError(f'sched_pose={kind!r}: only ExponentialLR is used by the reference configurations')
    ratio = cfg_train.lr_pose_end / (1e-10 + cfg_train.lr_pose)
    return optim, torch.optim.lr_scheduler.ExponentialLR(optim, gamma=ratio ** (1. / max_iter))


def plain_kwargs(kwargs):
    """Constructor kwargs with numpy arrays / scalars and tensors turned into plain Python lists and numbers, so that a
    checkpoint holding them loads with `weights_only=True` and nothing else."""
    def conv(v):
        if isinstance(v, (np.ndarray, np.generic)):
            return v.tolist()
        if isinstance(v, torch.Tensor):
            return v.detach().cpu().tolist()
        if isinstance(v, dict):
            return {k: conv(x) for k, x in v.items()}
        if isinstance(v, (list, tuple)):
            return [conv(x) for x in v]
        return v
    return {k: conv(v) for k, v in kwargs.items()}


def _numpy_array_globals():
    """The reconstructors a pickled numpy array needs - and nothing else: a reference checkpoint keeps `xyz_min`, `HW`,
    `i_train`, ... as numpy arrays inside `model_kwargs` (lib/voxurf_coarse.py get_kwargs)."""
    try:
        from numpy._core import multiarray as ma
    except ImportError:                                    # numpy < 2
        from numpy.core import multiarray as ma
    allowed = [np.ndarray, np.dtype, ma._reconstruct, ma.scalar]
    allowed += [type(np.dtype(t)) for t in (np.float32, np.float64, np.int32, np.int64, np.uint8, np.bool_)]
    return allowed


def load_checkpoint_file(path):
    """torch.load with weights_only=True; numpy arrays are the one extra thing admitted.  Nothing in the file is executed:
    a checkpoint that needs any other global fails with torch's UnpicklingError instead of being unpickled."""
    with torch.serialization.safe_globals(_numpy_array_globals()):
        return torch.load(path, map_location='cpu', weights_only=True)


def save_model(model, path, global_step=0, optimizer=None, **extra):
    """Checkpoint in the reference's key layout (lib/recon_scene.py:779-791) with plain-list constructor kwargs."""
    ck = {'global_step': int(global_step), 'model_kwargs': plain_kwargs(model.get_kwargs()),
          'MaskCache_kwargs': plain_kwargs(model.get_MaskCache_kwargs()), 'model_state_dict': model.state_dict()}
    if optimizer is not None:
        ck['optimizer_state_dict'] = optimizer.state_dict()
    ck.update(extra)
    torch.save(ck, path)


def load_model(model_class, ckpt_path, strict=True, device='cuda'):
    """Counterpart of lib/utils.py:416-438: rebuild the model from `model_kwargs`, load `model_state_dict`.  Reads files
    written by this package (save_model, TrainEngine.save_checkpoint) and reference checkpoints alike, always weights-only."""
    ckpt = load_checkpoint_file(ckpt_path)
    model = model_class(**ckpt['model_kwargs'])
    sd = dict(ckpt['model_state_dict'])
    sd.pop('s_val', None)        # only present when the reference model was built on CPU (voxurf_coarse.py:94)
    model.load_state_dict(sd, strict=strict)
    return model.to(device)
