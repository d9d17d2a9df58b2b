"""Full-image inference through the drop-in module (Voxurf.inference, reference-style 4096-ray chunks, lib/nvs_fun.py:39-86):
rays/s for one 400x400 view.   python tools/bench_inference.py [G] [chunk]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from poseprobe_amd import synthetic as syn
from poseprobe_amd import voxurf_coarse as Model

G = int(sys.argv[1]) if len(sys.argv) > 1 else 160
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
H = W = 400
rs = syn.range_shape()
m = Model.Voxurf(syn.XYZ_MIN, syn.XYZ_MAX, num_voxels=G ** 3, num_voxels_base=G ** 3, alpha_init=1e-2, rgbnet_dim=12,
                 rgbnet_direct=True, rgbnet_depth=4, rgbnet_width=128, posbase_pe=5, viewbase_pe=1, geo_rgb_dim=3, s_ratio=50,
                 s_start=0.2, barf_c2f=[0.6, 1], i_train=np.arange(3), N_iters=10000, HW=np.array([[H, W]] * 3), range_shape=rs,
                 rect_size=rs.tolist(), camera_noise=0.).cuda()
with torch.no_grad():
    m.k0.grid.normal_(0, 0.1)
K = torch.tensor(syn.intrinsics(1, H, W)[0]).cuda()
w2c = torch.tensor(syn.cameras(3)[1])
c2w = torch.eye(4); c2w[:3] = w2c; c2w = torch.linalg.inv(c2w)[:3].cuda()
rays_o, rays_d, viewdirs = Model.get_rays_of_a_view(H, W, K, c2w, ndc=False, inverse_y=True, flip_x=False, flip_y=False)
ro, rd, vd = rays_o.reshape(-1, 3), rays_d.reshape(-1, 3), viewdirs.reshape(-1, 3)
kw = dict(near=syn.NEAR, far=syn.FAR, bg=0, stepsize=1.5, inverse_y=True, flip_x=False, flip_y=False)

def render():
    out = []
    with torch.no_grad():
        for b in range(0, H * W, chunk):
            r = m.inference(ro[b:b + chunk], rd[b:b + chunk], vd[b:b + chunk], global_step=None, **kw)
            out.append(r['rgb_marched'])
    return torch.cat(out)

img = render(); torch.cuda.synchronize()
t0 = time.perf_counter()
n = 3
for _ in range(n): img = render()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f'G={G} chunk={chunk}: {dt * 1e3:.1f} ms per 400x400 view = {H * W / dt / 1e6:.2f} M rays/s (mean rgb {float(img.mean()):.4f})')
