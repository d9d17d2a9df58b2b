"""Stub harness that lets the *reference's own Python* (read-only at /root/reference) be imported on
CPU in the build container, so that golden vectors can be generated from it.

TEST INFRASTRUCTURE ONLY.  Nothing here ships: it is used by oracle/make_golden.py (run by hand in
the build container, where /root/reference exists) and never on the GPU box.

What is stubbed and why
  * cv2, mcubes, termcolor, ipdb, easydict, plyfile, skimage, trimesh, imageio : absent third-party
    modules only touched by mesh / vis / IO code that the hot path never calls.
  * torch_scatter.segment_coo(src, index, out, reduce='sum') : third-party (un-vendored, unpinned);
    restated from its documented semantics as out.index_add_(0, index, src)  -> "parity unpinned"
    for summation ORDER only (values compared with tolerance).
  * torch.utils.cpp_extension.load : the reference JIT-compiles lib/cuda/*.cu with nvcc at import
    (lib/voxurf_coarse.py:19-24, lib/grid.py:12-24).  nvcc/CUDA do not exist here, so the loader is
    replaced by an object exposing alpha2weight / alpha2weight_backward / sample_pts_on_rays that
    call oracle.native_ops (our restatement of lib/cuda/render_utils_kernel.cu).
  * torch.Tensor.cuda : identity (lib/voxurf_coarse.py:94,:495 hard-code .cuda()).
"""
import sys
import types

import torch

REF_ROOT = '/root/reference'


class _EasyDict(dict):
    def __init__(self, d=None, **kw):
        super().__init__()
        d = dict(d or {}, **kw)
        for k, v in d.items():
            self[k] = v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def install():
    """Install the stubs and put the reference on sys.path. Idempotent."""
    if getattr(install, '_done', False):
        return
    sys.dont_write_bytecode = True
    for name in ('cv2', 'mcubes', 'termcolor', 'trimesh', 'imageio'):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules['termcolor'].colored = lambda s, *a, **k: s
    ipdb = types.ModuleType('ipdb')
    ipdb.set_trace = lambda *a, **k: None
    sys.modules.setdefault('ipdb', ipdb)
    ed = types.ModuleType('easydict')
    ed.EasyDict = _EasyDict
    sys.modules.setdefault('easydict', ed)
    ply = types.ModuleType('plyfile')
    ply.PlyData = ply.PlyElement = object
    sys.modules.setdefault('plyfile', ply)
    sk = types.ModuleType('skimage')
    sk.measure = types.ModuleType('skimage.measure')
    sys.modules.setdefault('skimage', sk)
    sys.modules.setdefault('skimage.measure', sk.measure)

    ts = types.ModuleType('torch_scatter')

    def segment_coo(src, index, out=None, reduce='sum'):
        assert reduce == 'sum' and out is not None
        return out.index_add_(0, index, src)

    ts.segment_coo = segment_coo
    sys.modules.setdefault('torch_scatter', ts)

    from oracle import native_ops

    class _FakeExt:
        @staticmethod
        def alpha2weight(alpha, ray_id, n_rays):
            return native_ops.alpha2weight(alpha, ray_id, n_rays)

        @staticmethod
        def alpha2weight_backward(alpha, weight, T, alphainv_last, i_start, i_end, n_rays, gw, gl):
            return native_ops.alpha2weight_backward(alpha, weight, T, alphainv_last, i_start, i_end,
                                                    n_rays, gw, gl)

        @staticmethod
        def sample_pts_on_rays(rays_o, rays_d, xyz_min, xyz_max, near, far, stepdist):
            return native_ops.sample_pts_on_rays(rays_o, rays_d, xyz_min, xyz_max, near, far, stepdist)

    import torch.utils.cpp_extension as cpp_ext
    cpp_ext.load = lambda **kw: _FakeExt
    torch.Tensor.cuda = lambda self, *a, **k: self
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    install._done = True


def import_reference():
    """Returns the reference modules (lib.voxurf_coarse, lib.dvgo_ori, lib.camera, lib.losses)."""
    install()
    import importlib
    V = importlib.import_module('lib.voxurf_coarse')
    D = importlib.import_module('lib.dvgo_ori')
    C = importlib.import_module('lib.camera')
    L = importlib.import_module('lib.losses')
    return V, D, C, L
