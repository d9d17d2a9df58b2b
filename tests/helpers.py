"""Shared helpers for the parity tests: load golden fixtures into oracle structures."""
import os

import numpy as np
import torch

from oracle import voxurf_oracle as O
from poseprobe_amd import synthetic as syn

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


def scene_for(G, **kw):
    rs = syn.range_shape()
    return O.Scene(syn.XYZ_MIN, syn.XYZ_MAX, int(G) ** 3, output_range=float(rs.max()), rect_size=rs.tolist(), **kw)


def params_from_npz(d):
    P = {'k0': torch.tensor(d['P.k0']), 'sdf': torch.tensor(d['P.sdf']),
         'sdf_alpha': torch.tensor(d['P.sdf_alpha']), 'sdf_beta': torch.tensor(d['P.sdf_beta']),
         'rgbnet': [], 'warp': []}
    for li in range(4):
        P['rgbnet'].append((torch.tensor(d[f'P.rgbnet.{li}.weight']), torch.tensor(d[f'P.rgbnet.{li}.bias'])))
    for li in range(5):
        P['warp'].append((torch.tensor(d[f'P.warp.{li}.weight']), torch.tensor(d[f'P.warp.{li}.bias'])))
    return P


def oracle_step_from_golden(d):
    """Re-run the oracle on a forward_* fixture's inputs: returns (out, S, loss, P, se3)."""
    G = int(d['G'])
    scene = scene_for(G)
    P = O.params_require_grad(params_from_npz(d))
    se3 = torch.tensor(d['se3'], requires_grad=True)
    w2c = O.current_pose_pnp(se3, torch.tensor(d['w2c_init']))
    c2w = O.pose_invert(w2c)
    ro, rd, vd, target, mask = O.select_training_rays(torch.tensor(d['ray_idx']), torch.tensor(d['images']),
                                                      torch.tensor(d['masks']), torch.tensor(d['Ks']), c2w)
    gs = int(d['global_step'])
    out = O.voxurf_forward(P, scene, ro, rd, vd, jitter=torch.tensor(d['jitter']), global_step=gs)
    S, Wt, loss = O.object_losses(out, target, mask, gs, scene.N_iters)
    (loss * 0.1).backward()
    return out, S, loss, P, se3, dict(rays_o=ro, rays_d=rd, viewdirs=vd, target=target, mask=mask, c2w=c2w, w2c=w2c)


def assert_close(a, b, rtol=1e-5, atol=1e-6, name='', scaled=0.0):
    """|a-b| <= atol + rtol*|b| + scaled*max|b|.  `scaled` expresses an error budget relative to the
    largest magnitude in the tensor (fp32 sums of large terms leave absolute errors on small entries)."""
    a = np.asarray(a.detach().cpu() if isinstance(a, torch.Tensor) else a, dtype=np.float64)
    b = np.asarray(b.detach().cpu() if isinstance(b, torch.Tensor) else b, dtype=np.float64)
    assert a.shape == b.shape, f'{name}: shape {a.shape} vs {b.shape}'
    if a.size == 0:
        return
    err = np.abs(a - b)
    tol = atol + rtol * np.abs(b) + scaled * np.abs(b).max()
    bad = err > tol
    assert not bad.any(), (f'{name}: {bad.sum()}/{a.size} mismatches, max abs err {err.max():.3e}, '
                           f'max |ref| {np.abs(b).max():.3e}')


def assert_mostly_close(a, b, rtol, scaled, name='', outlier_frac=0.03, outlier_scaled=3e-2):
    """For gradients of ReLU networks compared across two fp32 implementations: a pre-activation that lies within rounding
    distance of zero flips its mask in one of them and changes that sample's contribution discretely (a handful of the ~1e7
    activations of a test do).  Rows (first dimension) are therefore allowed to miss the tight tolerance in at most
    `outlier_frac` of the cases, and every element has to meet the loose bound `outlier_scaled` * max|b|."""
    a = np.asarray(a.detach().cpu() if isinstance(a, torch.Tensor) else a, dtype=np.float64)
    b = np.asarray(b.detach().cpu() if isinstance(b, torch.Tensor) else b, dtype=np.float64)
    assert a.shape == b.shape, f'{name}: shape {a.shape} vs {b.shape}'
    err = np.abs(a - b)
    mx = np.abs(b).max()
    loose = err > outlier_scaled * mx + rtol * np.abs(b)
    assert not loose.any(), f'{name}: {loose.sum()}/{a.size} beyond the loose bound, max abs err {err.max():.3e}, max |ref| {mx:.3e}'
    bad = (err > rtol * np.abs(b) + scaled * mx).reshape(a.shape[0], -1).any(axis=1)
    assert bad.mean() <= outlier_frac, (f'{name}: {bad.sum()}/{bad.size} rows miss the tight tolerance '
                                        f'(max abs err {err.max():.3e}, max |ref| {mx:.3e})')


def assert_close_but(a, b, rtol, atol, name, frac=1e-3, loose_atol=5e-4):
    """Per-sample quantities at the large grids: all entries within `loose_atol`, and all but a fraction `frac` within the
    stated tight tolerance (the NeuS alpha is a quotient of sigmoid differences; where sigma(prev / s) is tiny, fp32
    cancellation leaves 1e-4-level absolute noise on a handful of the ~1e5 samples, in the reference's own fp32 as well)."""
    a = np.asarray(a.detach().cpu() if isinstance(a, torch.Tensor) else a, dtype=np.float64)
    b = np.asarray(b.detach().cpu() if isinstance(b, torch.Tensor) else b, dtype=np.float64)
    assert a.shape == b.shape, f'{name}: shape {a.shape} vs {b.shape}'
    err = np.abs(a - b)
    assert err.max() <= loose_atol + rtol * np.abs(b).max(), f'{name}: max abs err {err.max():.3e}'
    bad = err > atol + rtol * np.abs(b)
    assert bad.mean() <= frac, f'{name}: {bad.sum()}/{a.size} outside rtol {rtol} / atol {atol} (max abs err {err.max():.3e})'


def assert_normals_close(a, b, name='gradient', frac=1e-4):
    """SDF normals [M,3] of two fp32 implementations: the trilinear interpolant is continuous but its gradient jumps at cell
    faces, so a deformed sample within rounding distance of a face (|u - round(u)| ~ 1e-5 voxels: a couple of the ~1e5
    coordinates of a step) reads its normal from the neighbouring cell in one of them.  All rows but a fraction `frac` meet the
    tight bound (rtol 1e-4 + 2e-5 of the largest entry); the exceptions stay below the jump a cell face can cause."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, f'{name}: shape {a.shape} vs {b.shape}'
    err, mx = np.abs(a - b), np.abs(b).max()
    bad = (err > 1e-5 + 1e-4 * np.abs(b) + 2e-5 * mx).any(axis=-1)
    assert bad.mean() <= frac, f'{name}: {bad.sum()}/{bad.size} rows outside the tight bound (max abs err {err.max():.3e}, max |ref| {mx:.3e})'
    assert err.max() <= 2e-2 * mx, f'{name}: max abs err {err.max():.3e} (max |ref| {mx:.3e})'


def ref96_inputs(d):
    """Inputs of tests/golden/forward_ref96.npz regenerated from the seeds it stores (the reference's real configuration:
    96^3 voxels, 113 samples per ray, 1024 rays of 3 views 400 x 400; oracle/make_golden.py::gen_forward_ref96)."""
    from poseprobe_amd.engine import SceneConfig
    from poseprobe_amd.params_init import reference_like_params
    G, H, W, nv, N = (int(d[k]) for k in ('G', 'H', 'W', 'n_views', 'n_rand'))
    rs = syn.range_shape()
    cfg = SceneConfig(syn.XYZ_MIN, syn.XYZ_MAX, G ** 3, stepsize=float(d['stepsize']), out_range=float(rs.max()))
    assert cfg.n_samples == int(d['n_samples']) == 113
    idx, jit = syn.step_randomness(nv * H * W, N, seed=int(d['batch_seed']))
    return dict(cfg=cfg, views=syn.make_views(nv, H, W), P=reference_like_params(cfg, int(d['param_seed'])),
                se3=syn.se3_perturbation(nv), idx=idx, jit=jit, gs=int(d['global_step']), G=G, H=H, W=W, nv=nv, N=N,
                scene=O.Scene(syn.XYZ_MIN, syn.XYZ_MAX, G ** 3, stepsize=float(d['stepsize']), output_range=float(rs.max()),
                              rect_size=rs.tolist()))


def check_against_ref96(d, get, tol_pix=dict(rtol=1e-4, atol=1e-5), tol_grad=dict(rtol=1e-3, scaled=5e-5), frac=2e-3):
    """Compare one step's results with the reference's stored outputs.  `get(name)` returns the candidate's numpy array for:
    rgb_marched, alphainv_cum, cum_weights [N], depth, weights [M], raw_alpha [M], raw_rgb [M,3], gradient [M,3], sdf_deform [M],
    samples_per_ray, loss.<k>, grad.se3, grad.sdf_alpha, grad.sdf_beta, grad.rgbnet.<l>.weight|bias, grad.warp.<l>.weight|bias,
    k0_grad (callable result: [X*Y*Z, 12] rows for given voxel ids), k0g.sum / abs_sum / n_touched."""
    assert np.array_equal(get('samples_per_ray'), d['samples_per_ray']), 'per-ray in-bbox sample counts (indices) differ'
    assert_close_but(get('rgb_marched'), d['out.rgb_marched'], name='rgb_marched', frac=frac, **tol_pix)
    assert_close_but(get('alphainv_cum'), d['out.alphainv_cum'], name='alphainv_cum', frac=frac, **tol_pix)
    assert_close_but(get('cum_weights'), d['out.cum_weights'][:, 0], name='cum_weights', frac=frac, **tol_pix)
    assert_close_but(get('depth'), d['out.depth'], name='depth', frac=frac, **tol_pix)
    assert_close_but(get('weights')[::8], d['out8.weights'][:, 0], rtol=1e-4, atol=1e-6, name='weights')
    assert_close_but(get('raw_alpha')[::8], d['out8.raw_alpha'][:, 0], rtol=1e-4, atol=1e-6, name='raw_alpha')
    assert_close_but(get('raw_rgb')[::8], d['out8.raw_rgb'], rtol=1e-4, atol=1e-5, name='raw_rgb')
    assert_normals_close(get('gradient')[::8], d['out8.gradient'], frac=0.0 if frac == 0.0 else 2e-4)
    assert_close_but(get('sdf_deform')[::8], d['out8.sdf_deform'][:, 0], rtol=1e-4, atol=1e-6, name='sdf_deform')
    for k in ('img_render', 'weight_entropy_last', 'grad_constraint', 'grad_deform_constraint', 'sdf_correct_constraint',
              'sdf_deform_constraint', 'mask_render'):
        assert_close(np.float32(get('loss.' + k)), d['loss.' + k], rtol=2e-4, atol=1e-7, name='loss.' + k)
    assert_close(get('grad.se3'), d['grad.se3'], atol=1e-6, name='g.se3', **tol_grad)
    assert_close(get('grad.sdf_alpha'), d['grad.sdf_alpha'], atol=1e-7, name='g.sdf_alpha', **tol_grad)
    assert_close(get('grad.sdf_beta'), d['grad.sdf_beta'], atol=1e-7, name='g.sdf_beta', **tol_grad)
    for li in range(4):
        for w in ('weight', 'bias'):
            assert_close(get(f'grad.rgbnet.{li}.{w}'), d[f'grad.rgbnet.{li}.{w}'], atol=1e-8, name=f'g.rgbnet{li}.{w}', **tol_grad)
    for li in range(5):
        for w in ('weight', 'bias'):
            assert_close(get(f'grad.warp.{li}.{w}'), d[f'grad.warp.{li}.{w}'], atol=2e-7, name=f'g.warp{li}.{w}', **tol_grad)
    rows = get('k0_grad')(d['k0g.voxels'])
    assert_close(rows, d['k0g.values'], atol=1e-9, name='k0 gradient at the stored voxels', **tol_grad)
    # support size: a contribution that underflows to exact zero in one implementation only may differ, nothing else
    assert abs(int(get('k0g.n_touched')) - int(d['k0g.n_touched'])) <= 1e-3 * int(d['k0g.n_touched']), 'number of voxels with a colour-grid gradient'
    assert_close(np.float64(get('k0g.abs_sum')), d['k0g.abs_sum'], rtol=1e-4, name='sum |k0 grad|')
    assert_close(np.float64(get('k0g.sum')), d['k0g.sum'], rtol=1e-3, atol=1e-5 * float(d['k0g.abs_sum']), name='sum k0 grad')


# ---------------------------------------------------------------------------------------------------- trajectories
LR_OF = {'k0': 1e-1, 'se3': 1e-3, 'sdf_ab': 1e-2}         # configs/dtu_e2e/scan1.py:87-103; every MLP tensor: 1e-3


def engine_vs_oracle_tensors(eng, st, P):
    """{name: (engine value, oracle value)} of every trained tensor (numpy float64), reference layouts."""
    from poseprobe_amd.engine import unpack_rgbnet, unpack_warp
    c = lambda t: t.detach().cpu().double().numpy()
    out = {'k0': (c(eng.k0_reference_layout()), c(P['k0'])), 'se3': (c(eng.se3), c(st.se3)),
           'sdf_ab': (c(eng.flat.view('sdf_ab')), np.concatenate([c(P['sdf_alpha']), c(P['sdf_beta'])]))}
    for li, (Wt, b) in enumerate(unpack_rgbnet(eng.flat.view('rgbnet'))):
        out[f'rgbnet{li}.W'], out[f'rgbnet{li}.b'] = (c(Wt), c(P['rgbnet'][li][0])), (c(b), c(P['rgbnet'][li][1]))
    for li, (Wt, b) in enumerate(unpack_warp(eng.flat.view('warp'))):
        out[f'warp{li}.W'], out[f'warp{li}.b'] = (c(Wt), c(P['warp'][li][0])), (c(b), c(P['warp'][li][1]))
    return out


def put_engine_at_oracle_state(eng, st):
    lr = {g['name']: g['lr'] for g in st.groups}
    eng.load_training_state({g['name']: (g['p'], g['m'], g['v']) for g in st.groups}, st.se3, st.pose_m, st.pose_v, st.n_step,
                            {'k0': lr['k0'], 'rgbnet': lr['rgbnet.0.weight'], 'warp': lr['warp.0.weight'], 'sdf_ab': lr['sdf_alpha']},
                            st.lr_pose)


def assert_trajectory_close(now, start, n_steps, rtol, crossed=None, what='', coupled=False):
    """`now` / `start`: engine_vs_oracle_tensors() after / before `n_steps` optimiser steps.  Every entry of every tensor must
    satisfy  |engine - oracle| <= rtol * |oracle's movement| + 1e-2 * lr  (1e-4 * lr for the colour grid, whose entries move by
    whole lr-sized steps) - EXCEPT the explicitly identified sign-flip set: Adam's update lr * m / (sqrt(v) + eps) is sign-like,
    so an entry whose gradient passes through zero (relative to its own history: `crossed[name]` = min_t |g_t| / max_t |g_t|
    < 1e-2, recorded from the ORACLE's gradients) may take a different +-lr step in two fp32 implementations.  Such entries are
    counted: at most 5e-4 of a tensor (a handful of a 128 x 128 matrix), each within 2 * lr * n_steps.  Nothing else is excused - except, with coupled=True (long
    free-running horizons), entries COUPLED to a flipped one: a colour-grid voxel whose neighbour took the other +-0.1 step sees a
    different total-variation sign sum, a sample through it feeds different colours to the MLPs; those cannot be told from the
    oracle's own gradient history, so there the count (<= 5e-4) and the magnitude bound alone apply."""
    for name, (a, b) in now.items():
        lr = LR_OF.get(name, 1e-3)
        move = np.abs(b - start[name][1])
        tol = rtol * move + (1e-4 if name == 'k0' else 1e-2) * lr
        bad = np.abs(a - b) > tol
        if not bad.any():
            continue
        assert crossed is not None and name in crossed, f'{what}{name}: {bad.sum()} entries beyond rtol {rtol} of the movement (max dev {np.abs(a - b).max():.2e})'
        assert bad.mean() <= 5e-4, f'{what}{name}: {bad.sum()}/{bad.size} entries deviate'
        assert coupled or (crossed[name][bad] < 1e-2).all(), f'{what}{name}: a deviating entry never had a near-zero gradient'
        assert np.abs(a - b)[bad].max() <= 2 * lr * n_steps, f'{what}{name}: outlier beyond 2 lr per step'
