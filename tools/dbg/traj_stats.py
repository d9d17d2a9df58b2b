"""Deviation statistics of the engine's trajectory against the oracle trainer: teacher-forced (every step from the oracle's state) and
free-running, with the deterministic scatter.  Used to set the thresholds of tests/test_hip_step.py's trajectory tests."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import voxurf_oracle as O
from poseprobe_amd import synthetic as syn
from poseprobe_amd.engine import unpack_rgbnet, unpack_warp
from tests.helpers import load, params_from_npz, scene_for
from tests.test_hip_step import build_engine

tag = sys.argv[1] if len(sys.argv) > 1 else 'g24_s10'
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
d = load(f'forward_{tag}.npz')
V, H, W = d['images'].shape[:3]
c = lambda t: t.detach().cpu().double().numpy()


def make():
    eng, cfg = build_engine(d, pose_iters=1000, deterministic_scatter=True)
    P = params_from_npz(d)
    st = O.TrainState(P, scene_for(d['G']), torch.tensor(d['w2c_init']), torch.tensor(d['Ks']), torch.tensor(d['images']),
                      torch.tensor(d['masks']), se3_refine=torch.tensor(d['se3']), pose_iters=1000)
    eng.zero_grads()
    return eng, st, P


def tensors(eng, st, P):
    out = {'k0': (c(eng.k0_reference_layout()), c(P['k0'])), 'se3': (c(eng.se3), c(st.se3)),
           'sdf_ab': (c(eng.flat.view('sdf_ab')), np.concatenate([c(P['sdf_alpha']), c(P['sdf_beta'])]))}
    for li, (Wt, b) in enumerate(unpack_rgbnet(eng.flat.view('rgbnet'))):
        out[f'rgbnet{li}.W'] = (c(Wt), c(P['rgbnet'][li][0]))
    for li, (Wt, b) in enumerate(unpack_warp(eng.flat.view('warp'))):
        out[f'warp{li}.W'] = (c(Wt), c(P['warp'][li][0]))
    return out


def put(eng, st):
    lr = {g['name']: g['lr'] for g in st.groups}
    eng.load_training_state({g['name']: (g['p'], g['m'], g['v']) for g in st.groups}, st.se3, st.pose_m, st.pose_v, st.n_step,
                            {'k0': lr['k0'], 'rgbnet': lr['rgbnet.0.weight'], 'warp': lr['warp.0.weight'], 'sdf_ab': lr['sdf_alpha']}, st.lr_pose)


for mode in ('teacher-forced', 'free-running'):
    eng, st, P = make()
    print('====', mode, tag)
    for s in range(n_steps):
        idx, jit = syn.step_randomness(V * H * W, int(d['n_rand']), seed=40 + s)
        if mode == 'teacher-forced':
            put(eng, st)
        before = {k: v[1].copy() for k, v in tensors(eng, st, P).items()}
        st.step(torch.tensor(idx), torch.tensor(jit), 10 + s)
        eng.train_step(torch.tensor(idx, dtype=torch.int32, device='cuda'), torch.tensor(jit, device='cuda'), 10 + s)
        torch.cuda.synchronize()
        if s in (0, 2, 9, n_steps - 1):
            for k, (a, b) in tensors(eng, st, P).items():
                dev = np.abs(a - b)
                move = np.abs(b - before[k])
                rel = dev / (move + 1e-12)
                moved = move > 0
                print(f'step {s + 1:2d} {k:10s} n={a.size:8d} max|dev|={dev.max():.2e}  frac(dev>1e-6)={np.mean(dev > 1e-6):.2e} '
                      f'frac(dev>1e-4)={np.mean(dev > 1e-4):.2e}  frac(dev>1e-3*move & moved)={np.mean((dev > 1e-3 * move + 1e-7) & moved):.2e} max move={move.max():.2e}')
