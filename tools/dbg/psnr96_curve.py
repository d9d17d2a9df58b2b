"""PSNR curve of oracle / twin / HIP students at the reference configuration (96^3): where does the trajectory decorrelate?"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
every = int(sys.argv[2]) if len(sys.argv) > 2 else 5
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
pose_std = float(sys.argv[4]) if len(sys.argv) > 4 else 5e-3
r = bench.cpu_baseline_psnr('cuda:0', steps=steps, seed=seed, threads=16, eval_at=tuple(range(every, steps + 1, every)), twin_eps=1e-7,
                            pose_std=pose_std, **bench.REFERENCE_WORKLOAD)
for row in r['curve']:
    print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in row.items()}))
print('oracle s/step', r['oracle_s_per_step'])
