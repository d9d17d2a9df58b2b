import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch; torch.zeros(1, device='cuda')
from poseprobe_amd import _lib
from tests import test_hip_convergence as T
mode, n = int(sys.argv[1]), int(sys.argv[2])
_lib.set_option('mlp_split', mode)
bad = 0
import re
vals = []
for i in range(n):
    try:
        T.test_student_converges_to_teacher_images_and_poses()
        vals.append(T.LAST[1])
    except AssertionError as e:
        bad += 1; print('FAIL', i, str(e)[:300], flush=True); vals.append(T.LAST[1])
import numpy as np
v = np.sort(np.array(vals))
print(f'mode {mode}: {bad}/{n} failures; final loss quantiles 10/50/90/max: {v[len(v)//10]:.4f} {v[len(v)//2]:.4f} {v[(9*len(v))//10]:.4f} {v[-1]:.4f}', flush=True)
