"""PSNR parity (the second half of BASELINE.json's metric): the HIP engine and the oracle trainer, started from one
initialisation and fed identical per-step rays and jitter, must reach the same PSNR on held-out pixels within 0.1 dB
(lib/recon_scene.py:654-685 prints -10 log10(mse) every i_print steps; lib/utils.py mse2psnr)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_psnr_of_hip_and_oracle_training_runs_agree():
    sys.path.insert(0, ROOT)
    import bench
    r = bench.cpu_baseline_psnr('cuda:0', steps=150)
    print(r)
    assert r['psnr_oracle'] > 15.0, r                     # the run learned the teacher's views (untrained: ~8 dB)
    assert r['abs_delta_db'] <= 0.1, r
