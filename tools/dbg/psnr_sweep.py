import sys, os, json, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
ev = (10, 25, 50, 100, 200, 400)
for kw in (dict(), dict(gs0=100), dict(pose_std=0.0), dict(gs0=100, pose_std=0.0)):
    for seed in (0, 1, 2):
        r = bench.cpu_baseline_psnr('cuda:0', steps=400, seed=seed, threads=8, eval_at=ev, twin_eps=1e-7, **kw)
        print(kw, seed, ' | '.join(f"{c['step']}: {c['psnr_hip']:.2f} {c['psnr_oracle']:.2f} {c['psnr_oracle_twin']:.2f}" for c in r['curve']), flush=True)
