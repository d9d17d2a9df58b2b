"""GPU tests of the DVGO-surface operators (poseprobe_amd.render_utils mirrors the reference's extension modules)
against numpy restatements of the .cu text (oracle/dvgo_ops.py)."""
import numpy as np
import pytest
import torch

from tests.helpers import assert_close

pytestmark = pytest.mark.gpu
cu = lambda a: torch.as_tensor(np.asarray(a)).cuda()


def test_raw2alpha_uniform_and_nonuniform():
    from oracle import dvgo_ops as D
    from poseprobe_amd import render_utils as R
    rng = np.random.RandomState(0)
    d = (rng.randn(10007) * 4).astype(np.float32)
    d[:3] = [90.0, -90.0, 0.0]               # exp overflow -> inf handled like the reference
    gb = rng.randn(10007).astype(np.float32)
    for interval in (0.5, rng.rand(10007).astype(np.float32) + 0.1):
        e0, a0 = D.raw2alpha(d, -4.595, interval)
        iv = interval if np.isscalar(interval) else cu(interval)
        e, a = R.raw2alpha(cu(d), -4.595, iv)
        assert_close(e.cpu(), e0, rtol=2e-6, atol=0, name='exp_d')
        assert_close(a.cpu(), a0, rtol=1e-5, atol=1e-7, name='alpha')
        g0 = D.raw2alpha_backward(e0, gb, interval)
        g = R.raw2alpha_backward(e, cu(gb), iv)
        ok = np.isfinite(g0)
        assert_close(g.cpu().numpy()[ok], g0[ok], rtol=1e-4, atol=1e-10, name='grad')
    e, a = R.raw2alpha(torch.zeros(0, device='cuda'), 0.0, 1.0)          # empty input (kernel.cu:468-470)
    assert e.numel() == 0 and a.numel() == 0


def test_maskcache_lookup_and_samplers():
    from oracle import dvgo_ops as D
    from poseprobe_amd import render_utils as R
    rng = np.random.RandomState(1)
    world = rng.rand(9, 7, 5) > 0.5
    xyz = (rng.rand(5000, 3) * 1.4 - 0.2).astype(np.float32)
    scale, shift = np.array([8., 6., 4.], dtype=np.float32), np.array([0.25, 0.25, 0.25], dtype=np.float32)
    out = R.maskcache_lookup(cu(world), cu(xyz), cu(scale), cu(shift))
    assert np.array_equal(out.cpu().numpy(), D.maskcache_lookup(world, xyz, scale, shift))
    ro = (rng.randn(257, 3) * 0.1).astype(np.float32)
    rd = rng.randn(257, 3).astype(np.float32)
    lo, hi = np.array([-1, -1, -1], dtype=np.float32), np.array([1, 1, 1], dtype=np.float32)
    pts, mask = R.sample_ndc_pts_on_rays(cu(ro), cu(rd), cu(lo), cu(hi), 33)
    p0, m0 = D.sample_ndc(ro, rd, lo, hi, 33)
    assert_close(pts.cpu(), p0, rtol=1e-6, atol=1e-7, name='ndc pts')
    assert (mask.cpu().numpy() != m0).mean() < 1e-3
    t_max = (rng.rand(257) + 1.0).astype(np.float32)
    bg = R.sample_bg_pts_on_rays(cu(ro), cu(rd), cu(t_max), 0.5, 17)
    assert_close(bg.cpu(), D.sample_bg(ro, rd, t_max, 0.5, 17), rtol=2e-5, atol=1e-6, name='bg pts')


@pytest.mark.parametrize('mode', [0, 1, 2])
def test_adam_upd_variants(mode):
    from oracle import dvgo_ops as D
    from poseprobe_amd import render_utils as R
    rng = np.random.RandomState(2)
    n = 4099
    p, m, v = rng.randn(n).astype(np.float32), (rng.randn(n) * 1e-2).astype(np.float32), (rng.rand(n) * 1e-3).astype(np.float32)
    g = rng.randn(n).astype(np.float32)
    g[rng.rand(n) < 0.3] = 0
    perlr = rng.rand(n).astype(np.float32)
    P, G, M, V = cu(p), cu(g), cu(m), cu(v)
    if mode == 0:
        R.adam_upd(P, G, M, V, 7, 0.9, 0.99, 0.1, 1e-8)
    elif mode == 1:
        R.masked_adam_upd(P, G, M, V, 7, 0.9, 0.99, 0.1, 1e-8)
    else:
        R.adam_upd_with_perlr(P, G, M, V, cu(perlr), 7, 0.9, 0.99, 0.1, 1e-8)
    p0, m0, v0 = D.adam_upd(p, g, m, v, 7, 0.9, 0.99, 0.1, 1e-8, mode, perlr)
    assert_close(P.cpu(), p0, rtol=1e-5, atol=1e-6, name='param')
    assert_close(M.cpu(), m0, rtol=1e-5, atol=1e-8, name='exp_avg')
    assert_close(V.cpu(), v0, rtol=1e-5, atol=1e-10, name='exp_avg_sq')


@pytest.mark.parametrize('masked,dense', [(False, True), (False, False), (True, True)])
def test_total_variation_add_grad(masked, dense):
    from oracle import dvgo_ops as D
    from poseprobe_amd import render_utils as R
    rng = np.random.RandomState(3)
    shape = (1, 4, 6, 5, 7)
    p = (rng.randn(*shape) * 2).astype(np.float32)
    g = rng.randn(*shape).astype(np.float32)
    g[rng.rand(*shape) < 0.5] = 0
    mask = (rng.rand(*shape) > 0.3).astype(np.float32) if masked else None
    cl = lambda a: cu(a).contiguous(memory_format=torch.channels_last_3d)
    G = cl(g)
    R.total_variation_add_grad(cl(p), G, 0.3, 0.5, 0.7, dense, None if mask is None else cl(mask))
    assert_close(G.cpu(), D.tv_add_grad(p, g, 0.3, 0.5, 0.7, dense, mask), rtol=1e-5, atol=1e-6, name='tv grad')


def test_cumdist_thres_and_extension_level_alpha2weight():
    from oracle import dvgo_ops as D, native_ops
    from poseprobe_amd import render_utils as R
    rng = np.random.RandomState(4)
    dist = rng.rand(70, 90).astype(np.float32) * 0.1
    assert np.array_equal(R.cumdist_thres(cu(dist), 0.37).cpu().numpy(), D.cumdist_thres(dist, 0.37))
    # reference-extension signatures incl. i_start / i_end and empty rays
    lens = np.array([3, 0, 70, 0, 0, 5])
    ray_id = np.repeat(np.arange(6), lens)
    alpha = (rng.rand(int(lens.sum())) * 0.4).astype(np.float32)
    w0, T0, l0, is0, ie0 = native_ops.alpha2weight(torch.tensor(alpha), torch.tensor(ray_id), 6)
    w, T, last, i_s, i_e = R.alpha2weight(cu(alpha), cu(ray_id), 6)
    assert np.array_equal(w.cpu().numpy(), w0.numpy()) and np.array_equal(last.cpu().numpy(), l0.numpy())
    gw, gl = rng.randn(len(alpha)).astype(np.float32), rng.randn(6).astype(np.float32)
    g0 = native_ops.alpha2weight_backward(torch.tensor(alpha), w0, T0, l0, is0, ie0, 6, torch.tensor(gw), torch.tensor(gl))
    g = R.alpha2weight_backward(cu(alpha), w, T, last, i_s, i_e, 6, cu(gw), cu(gl))
    assert np.array_equal(g.cpu().numpy(), g0.numpy())
