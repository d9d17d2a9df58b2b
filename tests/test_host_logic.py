"""CPU tests of host-side logic mirrored from the reference: config loader (`_base_` inheritance), parameter packing,
BARF weights, s_val / dynamic-weight schedules, slab bounds."""
import math
import os

import numpy as np
import torch


def test_config_base_inheritance(tmp_path):
    from poseprobe_amd.config import Config
    (tmp_path / 'default.py').write_text(
        "expname = None\nfine_train = dict(N_iters=20000, N_rand=1024, lrate_k0=1e-1, ray_sampler='flatten')\n"
        "fine_model_and_render = dict(num_voxels=160**3, stepsize=1.5)\ndata = dict(inverse_y=False, ndc=False)\n")
    (tmp_path / 'scan1.py').write_text(
        "_base_ = './default.py'\nexpname = 'scan1'\nsurf_train = dict(N_iters=10000, lr_pose=0.)\n"
        "fine_train = dict(N_iters=10000)\nfine_model_and_render = dict(num_voxels=96**3)\ndata = dict(inverse_y=True)\n")
    cfg = Config.fromfile(str(tmp_path / 'scan1.py'))
    assert cfg.expname == 'scan1'
    assert cfg.fine_train.N_iters == 10000 and cfg.fine_train.N_rand == 1024 and cfg.fine_train.ray_sampler == 'flatten'
    assert cfg.fine_model_and_render.num_voxels == 96 ** 3 and cfg.fine_model_and_render.stepsize == 1.5
    assert cfg.data.inverse_y is True and cfg.data.ndc is False
    assert getattr(cfg.data, 'flip_x', 'dflt') == 'dflt'
    assert 'lrate_k0' in cfg.fine_train.keys()


def test_parameter_packing_roundtrip():
    from poseprobe_amd.engine import pack_rgbnet, pack_warp, unpack_rgbnet, unpack_warp
    from poseprobe_amd.ops import RGBNET_PARAMS, WARP_PARAMS
    g = torch.Generator().manual_seed(0)
    rg = [(torch.randn(128, 57, generator=g), torch.randn(128, generator=g)), (torch.randn(128, 128, generator=g), torch.randn(128, generator=g)),
          (torch.randn(128, 128, generator=g), torch.randn(128, generator=g)), (torch.randn(3, 128, generator=g), torch.randn(3, generator=g))]
    flat = pack_rgbnet(rg)
    assert flat.numel() == RGBNET_PARAMS
    for (W, b), (W2, b2) in zip(rg, unpack_rgbnet(flat)):
        assert torch.equal(W, W2) and torch.equal(b, b2)
    assert float(flat[:128 * 64].reshape(128, 64)[:, 57:].abs().sum()) == 0      # zero padding 57 -> 64
    wp = [(torch.randn(128, 3, generator=g), torch.randn(128, generator=g))] + \
         [(torch.randn(128, 128, generator=g), torch.randn(128, generator=g)) for _ in range(3)] + \
         [(torch.randn(4, 128, generator=g), torch.randn(4, generator=g))]
    flat = pack_warp(wp)
    assert flat.numel() == WARP_PARAMS
    for (W, b), (W2, b2) in zip(wp, unpack_warp(flat)):
        assert torch.equal(W, W2) and torch.equal(b, b2)


def test_schedules_match_the_oracle():
    from oracle import voxurf_oracle as O
    from poseprobe_amd import synthetic as syn
    from poseprobe_amd.engine import SceneConfig, dynamic_weight
    cfg = SceneConfig(syn.XYZ_MIN, syn.XYZ_MAX, 96 ** 3)
    scene = O.Scene(syn.XYZ_MIN, syn.XYZ_MAX, 96 ** 3)
    assert cfg.world_size == scene.world_size.tolist() == [96, 96, 96]
    assert cfg.n_samples == scene.n_samples() == 113
    assert np.float32(cfg.voxel_size) == np.float32(float(scene.voxel_size))
    for prog in (0.0, 0.001, 0.6, 0.7, 0.8, 1.0):
        w = cfg.pe_weights(prog)
        assert np.array_equal(w[:5], O.barf_weights(scene, prog, 5).numpy())
        assert np.array_equal(w[5:], O.barf_weights(scene, prog, 1).numpy())
    for gs in (0, 10, 7000):
        assert cfg.s_val(gs) == O.s_val_at(scene, gs)
        assert dynamic_weight(1e-1, 1e-3, gs, 10000) == O.dynamic_weight(1e-1, 1e-3, gs, 10000)
    assert math.isclose(dynamic_weight(1e-1, 1e-3, 10000, 10000), 1e-3)


def test_committed_bench_record_follows_the_contract():
    """profiles/r01_end_bench.json is the line `python bench.py` printed on the MI355X: keys and types of the driver's
    contract (metric, value, unit, n_gpus, steps, warmup, ms_per_step, higher_is_better, scaling, vs_baseline, dtype, data,
    config.workload) plus the `roofline` and `cpu_baseline` objects."""
    import json
    import os
    from tests.conftest import ROOT
    d = json.load(open(os.path.join(ROOT, 'profiles', 'r01_end_bench.json')))
    for k, t in (('metric', str), ('value', float), ('unit', str), ('n_gpus', int), ('steps', int), ('warmup', int),
                 ('ms_per_step', float), ('higher_is_better', bool), ('scaling', str), ('dtype', str), ('data', str)):
        assert isinstance(d[k], t), k
    assert d['vs_baseline'] is None and d['scaling'] == 'weak' and d['higher_is_better'] is True
    assert 'workload' in d['config'] and 'model' not in d['config']
    r = d['roofline']
    assert r['bound'] in ('hbm', 'mfma') and r['unit'] in ('GB/s', 'TFLOP/s')
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9 and 0 < r['frac'] < 1
    assert r['traffic'] is None or r['traffic'] >= 0.9 * r['algorithmic_bytes_per_launch']
    c = d['cpu_baseline']
    assert c['kind'] in ('port', 'reference') and c['cores'] >= 1 and c['value'] > 0 and isinstance(c['sample'], str)
    assert abs(d['value'] - 1024 * d['n_gpus'] / (d['ms_per_step'] * 1e-3)) < 1e-3 * d['value']


def test_joint_loop_schedules():
    """Pure schedule functions of the trainer counterpart (lib/utils.py:300-313, renderer.py:580-584, sparf.py:33)."""
    from poseprobe_amd import trainer as T
    assert abs(T.scene_lr(0, 1e-3, 1e-4, 1000) - 1e-3) < 1e-12
    assert abs(T.scene_lr(1000, 1e-3, 1e-4, 1000) - 1e-4) < 1e-12
    assert abs(T.scene_lr(500, 1e-3, 1e-4, 1000) - 10 ** -3.5) < 1e-12
    assert not T.fine_phase(299, 1000, 0.3) and T.fine_phase(300, 1000, 0.3)
    assert T.fine_phase(0, 1000, None) and not T.fine_phase(900, 1000, 0.3, fine_sampling=False)
    assert T.pose_phase(299, 1000, 0.3) and not T.pose_phase(300, 1000, 0.3)
    assert T.c2f_progress(250, 1000) == 0.25
