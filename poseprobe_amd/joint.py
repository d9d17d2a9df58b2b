"""Dual-branch optimisation step: object branch (voxel SDF renderer, engine.TrainEngine) and scene branch (NeRF MLP,
bg_nerf.SceneEngine) driven by ONE set of camera poses, `loss = 0.1 * L_obj + L_bg` (lib/recon_scene.py:645-649), one
optimiser step for each parameter group (:765-771).

The pose is shared the way the reference shares it (`model_bg.data_dict.poses_w2c = current_pose[train_idx]`, :639): the
object engine's pose kernel produces c2w and its Jacobian d c2w / d se3 once per step; the scene branch's rays are the
(per-ray, tiny) camera algebra of lib/bg_nerf/source/utils/camera.py:384-416 on that c2w, and the scene branch's ray
gradients are folded back through the same Jacobian, so se3 receives the sum of both branches' gradients before its Adam
step.
"""
import torch

from . import bg_nerf, ops


class DualBranchEngine:
    def __init__(self, obj_engine, scene_net, lr_scene=1e-3, depth_range=(0.5, 3.0), scene_net_fine=None):
        self.obj = obj_engine
        self.scene = bg_nerf.SceneEngine(scene_net, lr=lr_scene, net_fine=scene_net_fine)
        self.depth_range = depth_range
        e = obj_engine
        self._se3_tmp = torch.zeros_like(e.se3_grad)
        self.last_scene_loss = None

    def scene_rays(self, pixels, n_views=None):
        """pixels [N, 2] (x, y; the same for every view, as the reference's sampler draws them) -> center, ray [V, N, 3] and
        the camera-frame directions [V, N, 3] for the pose chain (the first n_views views when the trainer's incremental
        schedule has not admitted all of them yet)."""
        e = self.obj
        k = e.V if n_views is None else n_views
        fx, fy, cx, cy = (e.intr[:k, i][:, None] for i in range(4))
        x, y = pixels[None, :, 0], pixels[None, :, 1]
        dir_cam = torch.stack([(x - cx) / fx, (y - cy) / fy, torch.ones_like((x - cx) / fx)], dim=-1)
        c2w = e.c2w[:k]
        ray = dir_cam @ c2w[:, :, :3].transpose(-1, -2)
        center = c2w[:, None, :, 3].expand_as(ray)
        return center, ray, dir_cam

    def forward_backward(self, ray_idx, jitter, global_step, pixels, image, depth_rand=None, fine=False, fine_grid=None,
                         n_views=None):
        """ray_idx / jitter: the object branch's batch (engine.TrainEngine.train_step); pixels [N, 2] + image [V, N, 3]: the
        scene branch's batch; depth_rand [V, N, S, 1] / fine_grid [Nf + 1] optionally replay the samplers' draws.  On return every gradient
        buffer (object engine's k0 / MLPs / se3 - the pose gradient of BOTH branches - and the scene engine's block) is
        filled; the object engine's gradients must be zero on entry (its optimiser kernels leave them so)."""
        e, sc = self.obj, self.scene
        out = e.render_and_grads(ray_idx, jitter, global_step)            # also refreshes e.c2w / e.jac for this step
        opt = sc.net.opt
        V, N, S = (e.V if n_views is None else n_views), pixels.shape[0], opt.nerf.sample_intvs
        center, ray, dir_cam = self.scene_rays(pixels, V)
        if depth_rand is None:
            depth = bg_nerf.sample_depth(opt, V, N, S, self.depth_range, mode='train', device=pixels.device)
        else:
            jit = depth_rand + torch.arange(S, device=pixels.device)[None, None, :, None].float()
            depth = jit / S * (self.depth_range[1] - self.depth_range[0]) + self.depth_range[0]
        loss_bg, g_center, g_ray = sc.forward_backward(center.reshape(V * N, 3).contiguous(), ray.reshape(V * N, 3).contiguous(),
                                                       depth.reshape(V * N, S).contiguous(), image.reshape(V * N, 3),
                                                       fine=fine, depth_range=self.depth_range, fine_grid=fine_grid)
        # fold the ray gradients into d L_bg / d c2w and through the object engine's pose Jacobian
        g_ray, g_center = g_ray.view(V, N, 3), g_center.view(V, N, 3)
        g_c2w = torch.cat([torch.einsum('vni,vnj->vij', g_ray, dir_cam), g_center.sum(1)[..., None]], dim=-1)
        if V < e.V:                                     # views that are not in play yet receive no scene gradient
            g_c2w = torch.cat([g_c2w, torch.zeros(e.V - V, 3, 4, device=g_c2w.device)], dim=0)
        g_c2w = g_c2w.contiguous()
        ops.pose_bwd(e.jac, g_c2w, self._se3_tmp)
        e.se3_grad += self._se3_tmp
        self.last_scene_loss = loss_bg
        return out, loss_bg

    def train_step(self, ray_idx, jitter, global_step, pixels, image, depth_rand=None, optimize_pose=True, fine=False,
                   fine_grid=None, n_views=None, before_step=None):
        """fine=True: the scene branch also runs its fine network (after ratio_start_fine_sampling_at_x of the schedule).
        before_step: callable run after both branches' backward and before the optimiser step (the trainer mixes its extra
        pose-only loss terms into se3_grad there)."""
        out = self.forward_backward(ray_idx, jitter, global_step, pixels, image, depth_rand, fine, fine_grid, n_views)
        if before_step is not None:
            before_step()
        e = self.obj
        e.grad_scale = 1.0
        if e.dist is not None:
            # ray-sharded data parallelism: the object engine's exchange already carries se3_grad, which holds
            # the pose gradient of BOTH branches; the scene networks add one 2 MB all-reduce each (averaged by grad_scale)
            e.dist.reduce_gradients(e)
            for st in self.scene.states:
                if st.has_grad:
                    e.dist.all_reduce_tensor(st.grad)
        e.optimizer_step(optimize_pose, grad_scale=e.grad_scale)
        self.scene.optimizer_step(grad_scale=e.grad_scale)
        if e.dist is not None:
            e.dist.gather_parameters(e)
        return out
