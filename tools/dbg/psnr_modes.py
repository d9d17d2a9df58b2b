import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch; torch.zeros(1, device="cuda:0")
import bench
from poseprobe_amd import _lib
for seed in (0, 1, 2):
    for mode in (0, 1):
        _lib.set_option('mlp_split', mode)
        r = bench.cpu_baseline_psnr('cuda:0', steps=25, seed=seed, threads=8, eval_at=(5, 10, 25), twin_eps=0.0)
        print('seed', seed, 'mode', mode, [(row['step'], round(row['psnr_hip'], 4), round(row['psnr_oracle'], 4)) for row in r['curve']], flush=True)
