"""Minimal counterpart of the reference's joint training loop (lib/recon_scene.py:534-791) around the two HIP engines: the
schedules that decide WHAT a step does, the batch samplers and the scene branch's snapshot file.  Not a re-implementation of
the trainer (view selection, PnP re-initialisation, matching, logging and evaluation stay with the reference): it exists to
drive `joint.DualBranchEngine` the way `scene_rep_reconstruction` drives the two models, and to read / write the files that
loop reads / writes.

Schedules (all pure functions of the step, unit-tested on the CPU):
  * scene learning rate   ExponentialLR from `optim.lr` to `optim.lr_end` over `max_iter` (lib/utils.py:300-313,
                          default_config.py:176-180),
  * coarse-to-fine window `progress = iteration_nerf / max_iter` written into both NeRFs (renderer.py:399-402),
  * fine network          from `ratio_start_fine_sampling_at_x * max_iter` on (renderer.py:580-584),
  * pose refinement       while `step < ratio_end_joint_nerf_pose_refinement * max_iter` (sparf.py:33, recon_scene.py:770).
"""
import os

import numpy as np
import torch

from . import bg_nerf
from .joint import DualBranchEngine


def scene_lr(step, lr=1e-3, lr_end=1e-4, max_iter=60000):
    """Learning rate the scheduler holds AFTER `step` optimiser steps."""
    gamma = (lr_end / lr) ** (1.0 / max_iter)
    return lr * gamma ** step


def fine_phase(step, max_iter, ratio_start_fine=0.3, fine_sampling=True):
    return bool(fine_sampling) and not (ratio_start_fine is not None and step < max_iter * ratio_start_fine)


def pose_phase(step, max_iter, ratio_end_pose=0.3):
    return step < max_iter * ratio_end_pose


def c2f_progress(iteration_nerf, max_iter):
    return iteration_nerf / max_iter


def active_views(global_step, n_views, incremental_step, incremental=True, start=2):
    """Number of views in play at `global_step`: the incremental schedule of lib/recon_scene.py:552-576 starts from the first
    two views and takes in one more whenever the step count reaches a multiple of `cfg.camera.incremental_step` (the
    check runs before the step's batch is drawn, never at step 0) until all views are in."""
    if not incremental or incremental_step <= 0:
        return n_views
    return int(min(n_views, start + max(global_step, 0) // incremental_step))


def object_phase(global_step, n_iters_object, start_object=0):
    """The object branch is optimised while start_object <= step <= cfg_train.N_iters (lib/recon_scene.py:584)."""
    return start_object <= global_step <= n_iters_object


class ReprojectionTerm:
    """The reprojection + near-surface pose terms of the live loop (lib/recon_scene.py:624-637) as a `pose_terms` entry of
    DualBranchTrainer: one matched view pair is drawn per step among the ACTIVE views, every matched pixel is lifted to the
    current surface, moved into the other view and compared with its match (recon_utils.get_project_error).  The surface
    point comes from the zero-crossing query of the raw template while at most two views are active
    (`pose_use_deform = optimize_object_nerf and len(selected_i_train) > 2`, :584 - i.e. at the start of EVERY run) and from
    the rendered depth afterwards; both are differentiated by the HIP backward (pp_sdf_crossing_dense_bwd / the render
    chain), so the pose receives the reference's gradient in either phase.

    pairs: list of (i, j, coord_i [P,2], coord_j [P,2], conf [P]) - matcher output, an input here as in bg_losses.
    weight_projection / weight_near_surface: cfg_train.projection_dis_error / cfg_train.weight_near_surface
    (configs/dtu_e2e/scan1.py:60-61: 1e-3 / 1e-1)."""
    __name__ = 'reprojection'

    def __init__(self, obj_engine, pairs, nl, weight_projection, weight_near_surface, pixel_thre=200, inverse_y=True,
                 flip_x=False, flip_y=False, seed=0):
        self.e, self.pairs = obj_engine, list(pairs)
        self.nl, self.w_proj, self.w_near, self.pixel_thre = float(nl), float(weight_projection), float(weight_near_surface), pixel_thre
        self.flags = dict(inverse_y=inverse_y, flip_x=flip_x, flip_y=flip_y)
        self.rng = np.random.RandomState(seed)
        self.model = obj_engine.voxurf_view()
        self.global_step = 0
        self.last = {}

    def __call__(self, se3, w2c_init, n_active):
        from . import camera, recon_utils
        e = self.e
        live = [p for p in self.pairs if p[0] < n_active and p[1] < n_active]
        if not live:
            return 0.0, se3.sum() * 0.0
        i, j, ci, cj, conf = live[self.rng.randint(len(live))]
        use_deform = n_active > 2                                   # recon_scene.py:584
        if use_deform:
            e.voxurf_view(self.model)                               # the rendered-depth query reads the current parameters
        w2c, _ = camera.current_pose_c2w(se3, w2c_init, fix_first=bool(e.refine_mask[0] == 0))
        cfg = e.cfg
        Ks = torch.zeros(e.V, 3, 3, device=e.dev)
        Ks[:, 0, 0], Ks[:, 1, 1], Ks[:, 0, 2], Ks[:, 1, 2], Ks[:, 2, 2] = e.intr[:, 0], e.intr[:, 1], e.intr[:, 2], e.intr[:, 3], 1.
        err, near = recon_utils.get_project_error(
            self.model, Ks, np.array([[e.H, e.W]] * e.V), self.nl, self.global_step, w2c, cj[None].to(e.dev), ci[None].to(e.dev),
            np.array([j]), np.array([i]), conf[None].to(e.dev), use_deform=use_deform, pixel_thre=self.pixel_thre,
            near=cfg.near, far=cfg.far, bg=cfg.bg, stepsize=cfg.stepsize, **self.flags)
        self.last = dict(projection_dis_error=float(err.detach()), loss_near_surface=float(near.detach()), use_deform=use_deform,
                         pair=(i, j), hits=int(recon_utils.get_project_error.last_valid.sum()))
        return 1.0, self.w_near * near + self.w_proj * err


class DualBranchTrainer:
    def __init__(self, obj_engine, opt, max_iter=60000, lr=1e-3, lr_end=1e-4, ratio_start_fine=0.3, ratio_end_pose=0.3,
                 depth_range=(0.5, 3.0), seed=0, incremental_step=0, pose_initialiser=None, pose_terms=()):
        """incremental_step > 0: the incremental view schedule (`active_views`); a view that joins gets its initial pose from
        `pose_initialiser(view, w2c_of_previous_view [3,4]) -> w2c [3,4]` - the reference's PnP hand-off (cv2.solvePnPRansac on
        matcher output, lib/recon_scene.py:202-214, :276-310) plugs in here; the default is its `use_identical` variant (the
        previous view's current pose).  pose_terms: extra pose-only loss terms mixed into the object loss as the reference
        mixes its reprojection / near-surface terms (:616-637): callables `f(se3 [V,6] requiring grad, w2c_init, n_active)
        -> (weight, scalar loss)`, differentiated by torch autograd through camera.current_pose_c2w."""
        self.opt, self.max_iter = opt, max_iter
        self.incremental_step, self.pose_initialiser, self.pose_terms = incremental_step, pose_initialiser, tuple(pose_terms)
        self.n_active = None
        self.lr, self.lr_end = lr, lr_end
        self.ratio_start_fine, self.ratio_end_pose = ratio_start_fine, ratio_end_pose
        dev = obj_engine.dev
        self.nerf = bg_nerf.NeRF(opt, device=dev)
        self.nerf_fine = bg_nerf.NeRF(opt, is_fine_network=True, device=dev) if opt.nerf.fine_sampling else None
        self.joint = DualBranchEngine(obj_engine, self.nerf, lr_scene=lr, depth_range=depth_range, scene_net_fine=self.nerf_fine)
        self.iteration = 0            # == the reference's Graph.iteration_nerf
        self.gen = torch.Generator(device=dev).manual_seed(seed)
        self.dev = dev

    # ---- batches (recon_scene.py:598-606 for the object branch, sampling_strategies.py:132-170 for the scene branch) ------
    def sample_batch(self, n_active=None):
        """Object-branch rays are a prefix of a permutation of the ACTIVE views' pixels (views 0 .. n_active - 1 are
        contiguous in the flattened [V, H, W] order); the scene branch draws rand_rays // n_active pixels per active view."""
        e = self.joint.obj
        k = e.V if n_active is None else n_active
        n_total = k * e.H * e.W
        ray_idx = torch.randperm(n_total, device=self.dev, generator=self.gen)[:e.N].to(torch.int32)
        jitter = torch.rand(e.N, device=self.dev, generator=self.gen)
        n_pix = self.opt.nerf.rand_rays // k
        flat = torch.randperm(e.H * e.W, device=self.dev, generator=self.gen)[:n_pix]
        py, px = flat // e.W, flat % e.W
        image = e.images[:k, py, px]                                   # [k, n_pix, 3] ground-truth colours at those pixels
        pixels = torch.stack([px.float() + 0.5, py.float() + 0.5], dim=-1)
        return ray_idx, jitter, pixels, image

    def _admit_views(self, global_step):
        """Incremental schedule: views joining at this step start from a handed-over pose with zero refinement."""
        e = self.joint.obj
        k = active_views(global_step, e.V, self.incremental_step, self.incremental_step > 0)
        if self.n_active is None:
            self.n_active = k if self.incremental_step <= 0 else min(k, 2)
        while self.n_active < k:
            v = self.n_active
            from . import ops
            ops.pose_fwd(e.se3, e.w2c_init, e.refine_mask, e.w2c, e.c2w, e.jac)          # current pose of the previous view
            prev = e.w2c[v - 1].detach().clone()
            init = prev if self.pose_initialiser is None else torch.as_tensor(self.pose_initialiser(v, prev.cpu()),
                                                                              dtype=torch.float32).to(e.dev)
            with torch.no_grad():
                e.w2c_init[v].copy_(init[:3, :4])
                e.se3[v].zero_(); e.se3_m[v].zero_(); e.se3_v[v].zero_()
            self.n_active += 1
        return self.n_active

    def _mix_pose_terms(self, k):
        """loss += w_i * L_i(poses) for the extra pose-only terms: their se3 gradient joins the object branch's (which the
        engine scales by loss_scale = 0.1, lib/recon_scene.py:648)."""
        if not self.pose_terms:
            return {}
        e = self.joint.obj
        se3 = e.se3.detach().clone().requires_grad_(True)
        values, total = {}, 0.
        for i, term in enumerate(self.pose_terms):
            if hasattr(term, 'global_step'):
                term.global_step = self.global_step
            w, val = term(se3, e.w2c_init, k)
            values[getattr(term, '__name__', f'term{i}')] = float(val.detach())
            total = total + w * val
        (total * e.loss_scale).backward()
        e.se3_grad += se3.grad * e.refine_mask[:, None].to(se3.grad.dtype)
        return values

    def train_step(self, global_step):
        """One iteration of the joint loop; returns (object-branch summary, scene loss)."""
        self.iteration += 1
        self.global_step = global_step
        k = self._admit_views(global_step)
        fine = fine_phase(global_step, self.max_iter, self.ratio_start_fine, self.nerf_fine is not None)
        ray_idx, jitter, pixels, image = self.sample_batch(k)
        self.last_pose_terms = {}
        out = self.joint.train_step(ray_idx, jitter, global_step, pixels, image, fine=fine,
                                    optimize_pose=pose_phase(global_step, self.max_iter, self.ratio_end_pose), n_views=k,
                                    before_step=(lambda: self.last_pose_terms.update(self._mix_pose_terms(k))) if self.pose_terms else None)
        self.joint.scene.set_lr(scene_lr(self.iteration, self.lr, self.lr_end, self.max_iter))
        p = c2f_progress(self.iteration, self.max_iter)                # takes effect from the next iteration (renderer.py:399)
        self.nerf.progress.data.fill_(p)
        if self.nerf_fine is not None:
            self.nerf_fine.progress.data.fill_(p)
        return out

    # ---- `model_last.pth.tar` (renderer.py:1028-1051 save_snapshot, recon_scene.py:827-838 load) ----------------------------
    def _adam_state_dict(self):
        """torch.optim.Adam.state_dict() layout over [nerf.parameters(), nerf_fine.parameters()] (lib/utils.py:294-299)."""
        state, groups, idx = {}, [], 0
        for st in self.joint.scene.states:
            ids = []
            views_m, views_v = st.net._views(st.m), st.net._views(st.v)
            for m, v in zip(views_m, views_v):
                if st.steps > 0:
                    state[idx] = {'step': torch.tensor(float(st.steps)), 'exp_avg': m.detach().clone().cpu(),
                                  'exp_avg_sq': v.detach().clone().cpu()}
                ids.append(idx)
                idx += 1
            ids.append(idx)                                            # `progress`: a Parameter without gradient, no state
            idx += 1
            groups.append({'lr': float(self.joint.scene.lr), 'betas': (0.9, 0.999), 'eps': 1e-8, 'weight_decay': 0,
                           'amsgrad': False, 'params': ids})
        return {'state': state, 'param_groups': groups}

    def state_dict(self):
        sd = {'nerf.' + k: v.detach().clone().cpu() for k, v in self.nerf.state_dict().items()}
        if self.nerf_fine is not None:
            sd.update({'nerf_fine.' + k: v.detach().clone().cpu() for k, v in self.nerf_fine.state_dict().items()})
        return sd

    def save_snapshot(self, directory, filename='model_last.pth.tar'):
        os.makedirs(directory, exist_ok=True)
        e = self.joint.obj
        torch.save({'current_pose': e.w2c.detach().clone().cpu(), 'epoch': 0, 'iteration': self.iteration,
                    'iteration_nerf': self.iteration, 'state_dict': self.state_dict(), 'best_val': None,
                    'epoch_of_best_val': None, 'optimizer': self._adam_state_dict(),
                    'scheduler': {'last_epoch': self.iteration, 'gamma': (self.lr_end / self.lr) ** (1.0 / self.max_iter),
                                  '_last_lr': [float(self.joint.scene.lr)] * len(self.joint.scene.states)}},
                   os.path.join(directory, filename))

    def load_snapshot(self, path):
        """Accepts a file written by save_snapshot or by the reference's Graph.save_snapshot (tensors only are read)."""
        ck = torch.load(path, map_location='cpu', weights_only=True)
        sd = ck['state_dict']
        nets = [('nerf.', self.nerf)] + ([('nerf_fine.', self.nerf_fine)] if self.nerf_fine is not None else [])
        for prefix, net in nets:
            net.load_state_dict({k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}, strict=True)
        self.iteration = int(ck['iteration_nerf'])
        opt_state, groups = ck['optimizer']['state'], ck['optimizer']['param_groups']
        for st, grp in zip(self.joint.scene.states, groups):
            ids = grp['params'][:-1]
            st.m.zero_(), st.v.zero_()
            st.steps = 0
            for i, m, v in zip(ids, st.net._views(st.m), st.net._views(st.v)):
                if i in opt_state:
                    m.copy_(opt_state[i]['exp_avg'])
                    v.copy_(opt_state[i]['exp_avg_sq'])
                    st.steps = int(opt_state[i]['step'])
        self.joint.scene.set_lr(scene_lr(self.iteration, self.lr, self.lr_end, self.max_iter))
        return ck.get('current_pose')
