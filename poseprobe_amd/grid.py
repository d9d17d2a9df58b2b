"""DenseGrid with the reference's module surface (lib/grid.py:38-89) on a channels-last physical layout.

The parameter keeps the reference's logical shape [1,C,X,Y,Z] (state_dict compatible) but is stored with
torch.channels_last_3d strides, i.e. physically [X,Y,Z,C]: one 8-corner stencil is 8 contiguous C*4-byte reads
for the HIP kernels and `load_state_dict` / optimisers keep working on the logical view.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops


def create_grid(type, **kwargs):
    if type == 'DenseGrid':
        return DenseGrid(**kwargs)
    raise NotImplementedError(f'{type}: only DenseGrid is constructed on the live path (lib/voxurf_coarse.py:121,194)')


def channels_last_view(param):
    """[1,C,X,Y,Z] channels_last_3d tensor -> its physical [X,Y,Z,C] contiguous view (no copy)."""
    t = param[0].permute(1, 2, 3, 0)
    if not t.is_contiguous():
        raise RuntimeError('grid parameter is not stored channels-last; call DenseGrid.ensure_layout()')
    return t


class _GridSample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, grid, xyz, sc, border):
        C = grid.shape[1]
        g_cl = channels_last_view(grid)
        pts = xyz.reshape(-1, 3).contiguous().float()
        out = torch.empty(pts.shape[0], C, device=grid.device)
        ops.grid_sample_fwd(sc, g_cl, C, pts, border, out)
        ctx.save_for_backward(grid, pts)
        ctx.sc, ctx.border, ctx.xyz_shape = sc, border, xyz.shape
        return out

    @staticmethod
    def backward(ctx, g_out):
        grid, pts = ctx.saved_tensors
        C = grid.shape[1]
        g_grid = g_pts = None
        if ctx.needs_input_grad[0]:
            g_grid = torch.zeros_like(grid, memory_format=torch.channels_last_3d)
        if ctx.needs_input_grad[1]:
            g_pts = torch.empty_like(pts)
        ops.grid_sample_bwd(ctx.sc, channels_last_view(grid), C, pts, ctx.border, g_out.contiguous(),
                            None if g_grid is None else channels_last_view(g_grid), g_pts)
        return g_grid, None if g_pts is None else g_pts.reshape(ctx.xyz_shape), None, None


class DenseGrid(nn.Module):
    def __init__(self, channels, world_size, xyz_min, xyz_max, **kwargs):
        super().__init__()
        self.channels = channels
        self.world_size = torch.as_tensor(world_size).long()
        self.register_buffer('xyz_min', torch.Tensor(xyz_min))
        self.register_buffer('xyz_max', torch.Tensor(xyz_max))
        ws = [int(v) for v in self.world_size.tolist()]
        self.grid = nn.Parameter(torch.zeros([1, channels, *ws]).contiguous(memory_format=torch.channels_last_3d))

    def ensure_layout(self):
        if not self.grid[0].permute(1, 2, 3, 0).is_contiguous():
            self.grid.data = self.grid.data.contiguous(memory_format=torch.channels_last_3d)

    def _apply(self, fn, *a, **k):      # .cuda()/.to() may drop the memory format
        r = super()._apply(fn, *a, **k)
        self.ensure_layout()
        return r

    def pp_scene(self, stepsize=1.0):
        ws = [int(v) for v in self.grid.shape[2:]]
        return ops.make_scene(self.xyz_min.tolist(), self.xyz_max.tolist(), ws, 1.0, stepsize, 0., 1., 0., 1.0,
                              max(self.channels, 1))

    def forward(self, xyz):
        """xyz: global coordinates to query (lib/grid.py:47-58: bilinear, align_corners, zeros padding)."""
        self.ensure_layout()
        shape = xyz.shape[:-1]
        out = _GridSample.apply(self.grid, xyz, self.pp_scene(), 0)
        out = out.reshape(*shape, self.channels)
        return out.squeeze(-1) if self.channels == 1 else out

    @torch.no_grad()
    def scale_volume_grid(self, new_world_size):
        ws = tuple(int(v) for v in new_world_size)
        if self.channels == 0:
            self.grid = nn.Parameter(torch.zeros([1, self.channels, *ws], device=self.grid.device))
        else:
            g = F.interpolate(self.grid.data.contiguous(), size=ws, mode='trilinear', align_corners=True)
            self.grid = nn.Parameter(g.contiguous(memory_format=torch.channels_last_3d))
        self.world_size = torch.as_tensor(ws).long()

    def get_dense_grid(self):
        return self.grid

    @torch.no_grad()
    def __isub__(self, val):
        self.grid.data -= val
        return self

    def extra_repr(self):
        return f'channels={self.channels}, world_size={self.world_size.tolist()}'
