"""Soak of the dual-branch loop at the bench workload: `steps` iterations of DualBranchTrainer (object 160^3 / 1024 rays + scene
3 x 341 rays x 128 samples, fine network from 30 %), checks finiteness and memory, prints one JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from poseprobe_amd import bg_nerf, synthetic as syn
from poseprobe_amd.engine import SceneConfig, TrainEngine
from poseprobe_amd.trainer import DualBranchTrainer
from bench import init_engine_params

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
G, H, W, V, N = 160, 400, 400, 3, 1024
dev = torch.device('cuda:0')
rs = syn.range_shape()
cfg = SceneConfig(syn.XYZ_MIN, syn.XYZ_MAX, G ** 3, out_range=float(rs.max()))
views = syn.make_views(V, H, W)
eng = TrainEngine(cfg, V, H, W, N, device=dev, pose_iters=3000)
eng.set_views(views['images'], views['masks'], views['Ks'], views['w2c'])
init_engine_params(eng, cfg, seed=3)
eng.zero_grads()
opt = bg_nerf.default_options(sample_intvs=128)
opt.nerf.fine_sampling, opt.nerf.sample_intvs_fine = True, 128
tr = DualBranchTrainer(eng, opt, max_iter=steps, depth_range=(0.5, 3.0))
losses, mem = [], []
t0 = time.perf_counter()
t_phase = {}
for s in range(steps):
    if s == int(0.3 * steps):
        torch.cuda.synchronize(); t_phase['coarse_ms'] = (time.perf_counter() - t0) / max(s, 1) * 1e3; t1 = time.perf_counter()
    _, loss_bg = tr.train_step(s)
    if s % 100 == 0 or s == steps - 1:
        losses.append(float(loss_bg)); mem.append(torch.cuda.memory_allocated() / 2 ** 30)
torch.cuda.synchronize()
t_phase['hierarchical_ms'] = (time.perf_counter() - t1) / (steps - int(0.3 * steps)) * 1e3
ok = bool(np.isfinite(losses).all() and torch.isfinite(tr.nerf.flat).all() and torch.isfinite(tr.nerf_fine.flat).all()
          and torch.isfinite(eng.k0_cl).all() and torch.isfinite(eng.se3).all())
print(json.dumps(dict(steps=steps, finite=ok, scene_loss_first=losses[0], scene_loss_last=losses[-1], mem_gib_first=mem[1],
                      mem_gib_last=mem[-1], fine_steps=tr.joint.scene.states[1].steps, **t_phase,
                      progress=float(tr.nerf.progress), lr_scene=float(tr.joint.scene.seg_lr))))
