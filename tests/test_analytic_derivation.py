"""The hand-derived (autograd-free) forward/backward that the HIP kernels implement, checked on CPU
against gradients produced by the reference itself (golden fixtures)."""
import pytest
import torch

from tests import analytic_model as AM
from tests.helpers import assert_close, load, params_from_npz, scene_for


@pytest.mark.parametrize('tag', ['g8_s10', 'g24_s10', 'g24_s7000'])
def test_analytic_step_matches_reference_gradients(tag):
    d = load(f'forward_{tag}.npz')
    scene = scene_for(d['G'])
    P = params_from_npz(d)
    r = AM.train_step_analytic(P, scene, d)
    assert_close(r['loss'], d['loss'], rtol=2e-5, name='loss')
    for k in ('rgb_marched', 'alphainv_cum', 'weights', 'raw_alpha', 'raw_rgb', 'gradient', 'grad_deform',
              'sdf_deform'):
        assert_close(r[k], d['out.' + k], rtol=1e-4, atol=2e-6, scaled=1e-6, name=k)
    assert_close(r['g_k0'], d['grad.k0'], rtol=1e-3, atol=1e-8, scaled=1e-5, name='g_k0')
    assert_close(r['g_sdf_alpha'], d['grad.sdf_alpha'][0], rtol=1e-3, atol=1e-7, scaled=1e-5, name='g_alpha')
    assert_close(r['g_sdf_beta'], d['grad.sdf_beta'][0], rtol=1e-3, atol=1e-7, scaled=1e-5, name='g_beta')
    for li in range(4):
        assert_close(r['g_rgbnet'][li][0], d[f'grad.rgbnet.{li}.weight'], rtol=1e-3, atol=1e-8, scaled=1e-5, name=f'rgb W{li}')
        assert_close(r['g_rgbnet'][li][1], d[f'grad.rgbnet.{li}.bias'], rtol=1e-3, atol=1e-8, scaled=1e-5, name=f'rgb b{li}')
    for li in range(5):
        assert_close(r['g_warp'][li][0], d[f'grad.warp.{li}.weight'], rtol=1e-3, atol=2e-7, scaled=1e-5, name=f'warp W{li}')
        assert_close(r['g_warp'][li][1], d[f'grad.warp.{li}.bias'], rtol=1e-3, atol=2e-7, scaled=1e-5, name=f'warp b{li}')
    g_se3 = AM.pose_chain_backward(torch.tensor(d['se3']), torch.tensor(d['w2c_init']), r['c2w_bar'])
    assert_close(g_se3, d['grad.se3'], rtol=1e-3, atol=1e-6, scaled=1e-5, name='g_se3')
