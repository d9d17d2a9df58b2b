"""Camera / pose algebra with the reference's surface (lib/camera.py:51-188: `pose`, `lie` singletons).

`lie.se3_to_SE3` and the fused `current_pose_c2w` run on the HIP pose kernel (pp_pose_fwd / pp_pose_bwd: forward-mode
Jacobian d c2w / d se3, consumed in the backward); compose / invert of [...,3,4] matrices are two tiny matmuls and
stay torch expressions (they are bookkeeping on V<=8 matrices, not part of the per-ray path).
"""
import torch

from . import ops


class Pose:
    def __call__(self, R=None, t=None):
        assert R is not None or t is not None
        if R is None:
            if not isinstance(t, torch.Tensor):
                t = torch.tensor(t)
            R = torch.eye(3, device=t.device).repeat(*t.shape[:-1], 1, 1)
        elif t is None:
            if not isinstance(R, torch.Tensor):
                R = torch.tensor(R)
            t = torch.zeros(R.shape[:-1], device=R.device)
        else:
            if not isinstance(R, torch.Tensor):
                R = torch.tensor(R)
            if not isinstance(t, torch.Tensor):
                t = torch.tensor(t)
        assert R.shape[:-1] == t.shape and R.shape[-2:] == (3, 3)
        return torch.cat([R.float(), t.float()[..., None]], dim=-1)

    def invert(self, pose, use_inverse=False):
        R, t = pose[..., :3], pose[..., 3:]
        R_inv = R.inverse() if use_inverse else R.transpose(-1, -2)
        return self(R=R_inv, t=(-R_inv @ t)[..., 0])

    def compose(self, pose_list):
        new = pose_list[0]
        for p in pose_list[1:]:
            new = self.compose_pair(new, p)
        return new

    def compose_pair(self, pose_a, pose_b):
        R_a, t_a = pose_a[..., :3], pose_a[..., 3:]
        R_b, t_b = pose_b[..., :3], pose_b[..., 3:]
        return self(R=R_b @ R_a, t=(R_b @ t_a + t_b)[..., 0])


class _PoseChain(torch.autograd.Function):
    """se3[V,6], w2c_init[V,3,4] -> (w2c, c2w) with refine o init and the inversion fused (recon_scene.py:62-74,:444)."""

    @staticmethod
    def forward(ctx, se3, w2c_init, refine_mask):
        V = se3.shape[0]
        f = dict(device=se3.device, dtype=torch.float32)
        w2c, c2w, jac = torch.empty(V, 3, 4, **f), torch.empty(V, 3, 4, **f), torch.empty(V, 12, 6, **f)
        ops.pose_fwd(se3.contiguous().float(), w2c_init.contiguous().float(), refine_mask, w2c, c2w, jac)
        ctx.save_for_backward(jac, w2c)
        return w2c, c2w

    @staticmethod
    def backward(ctx, g_w2c, g_c2w):
        jac, w2c = ctx.saved_tensors
        V = jac.shape[0]
        g = torch.zeros(V, 3, 4, device=jac.device) if g_c2w is None else g_c2w.contiguous().float().clone()
        if g_w2c is not None:
            # w2c = [R | t], c2w = [R^T | -R^T t]  =>  fold a gradient on w2c into the equivalent one on c2w
            R, t = w2c[..., :3], w2c[..., 3:]
            gR, gt = g_w2c[..., :3].float(), g_w2c[..., 3:].float()
            c = -(R.transpose(-1, -2) @ t)                       # camera centre = c2w[:, :, 3]
            g[..., :3] += gR.transpose(-1, -2) - c @ gt.transpose(-1, -2)
            g[..., 3:] += -(R.transpose(-1, -2) @ gt)
        se3_grad = torch.empty(V, 6, device=jac.device)
        ops.pose_bwd(jac, g.contiguous(), se3_grad)
        return se3_grad, None, None


def current_pose_c2w(se3_refine, w2c_init, fix_first=True):
    """(w2c, c2w) of get_current_pose_pnp + camera.pose.invert, differentiable w.r.t. se3_refine."""
    V = se3_refine.shape[0]
    mask = torch.ones(V, dtype=torch.int32, device=se3_refine.device)
    if fix_first:
        mask[0] = 0
    return _PoseChain.apply(se3_refine, w2c_init, mask)


class Lie:
    def se3_to_SE3(self, wu):
        """[...,6] -> [...,3,4] (lib/camera.py:127-142) through the HIP pose kernel (identity init, no fix)."""
        try:
            wu = torch.cat(wu, dim=0)
        except Exception:
            pass
        shape = wu.shape[:-1]
        flat = wu.reshape(-1, 6)
        eye = torch.eye(3, 4, device=wu.device).expand(flat.shape[0], 3, 4).contiguous()
        w2c, _ = _PoseChain.apply(flat, eye, None)
        return w2c.reshape(*shape, 3, 4)

    def skew_symmetric(self, w):
        w0, w1, w2 = w.unbind(dim=-1)
        O = torch.zeros_like(w0)
        return torch.stack([torch.stack([O, -w2, w1], dim=-1), torch.stack([w2, O, -w0], dim=-1),
                            torch.stack([-w1, w0, O], dim=-1)], dim=-2)


pose = Pose()
lie = Lie()
