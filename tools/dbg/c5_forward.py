import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import voxurf_oracle as O
from poseprobe_amd import synthetic as syn
from poseprobe_amd.engine import SceneConfig, TrainEngine
from poseprobe_amd.params_init import reference_like_params
G = int(sys.argv[1]) if len(sys.argv) > 1 else 320
V, N, H, W, GS = 6, 1024, 400, 400, 10
rs = syn.range_shape()
cfg = SceneConfig(syn.XYZ_MIN, syn.XYZ_MAX, G ** 3, out_range=float(rs.max()))
views = syn.make_views(V, H, W)
P = reference_like_params(cfg, 3)
se3 = syn.se3_perturbation(V)
eng = TrainEngine(cfg, V, H, W, N, pose_iters=3000)
eng.set_views(views['images'], views['masks'], views['Ks'], views['w2c'])
eng.load_reference_params(P['k0'], P['sdf'], P['sdf_alpha'], P['sdf_beta'], P['rgbnet'], P['warp'], se3=torch.tensor(se3))
eng.zero_grads()
idx, jit = syn.step_randomness(V * H * W, N, seed=11)
eng.render_and_grads(torch.tensor(idx, dtype=torch.int32, device='cuda'), torch.tensor(jit, device='cuda'), GS)
torch.cuda.synchronize()
scene = O.Scene(syn.XYZ_MIN, syn.XYZ_MAX, G ** 3, output_range=float(rs.max()), rect_size=rs.tolist())
with torch.no_grad():
    pass
s3 = torch.tensor(se3, requires_grad=True)
c2w = O.pose_invert(O.current_pose_pnp(s3, torch.tensor(views['w2c']), True))
ro, rd, vd, target, mask = O.select_training_rays(torch.tensor(idx), torch.tensor(views['images']), torch.tensor(views['masks']), torch.tensor(views['Ks']), c2w)
out = O.voxurf_forward(P, scene, ro, rd, vd, jitter=torch.tensor(jit), global_step=GS)
ws = eng.ws
M = int(ws.count.item())
print('M', M, 'world', cfg.world_size, 'S', cfg.n_samples)
def cmp(name, a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    e = (a - b).abs()
    i = int(e.reshape(-1).argmax())
    print(f'{name:14s} max err {float(e.max()):.3e} at flat {i} (row {i // max(1, a[0].numel())}) max|ref| {float(b.abs().max()):.3e}  n>1e-3: {int((e > 1e-3).sum())}')
wo = ws.warp_out[:M].view(M, 4, 4)
# oracle warp
pts = out['_ray_pts'].detach().requires_grad_(True)
deform, corr = O.warp_mlp(P, scene, pts)
cmp('deform', wo[:, 0, :3], deform)
cmp('correction', wo[:, 0, 3], corr.squeeze(-1))
cmp('sdf_final', ws.sdf_final[:M], out['_sdf_final'])
cmp('gradient', ws.gradient[:M], out['gradient'])
cmp('alpha', ws.alpha[:M], out['raw_alpha'])
cmp('feat', ws.feat[:M, :57], out['_rgb_feat'])
cmp('feat.k0', ws.feat[:M, :12], out['_rgb_feat'][:, :12])
cmp('rgb', ws.rgb[:M], out['raw_rgb'])
cmp('weights', ws.weights[:M], out['weights'])
cmp('rgb_marched', ws.rgb_marched, out['rgb_marched'])
# where do k0 features differ: sample rows & positions
e = (ws.feat[:M, :12].cpu() - out['_rgb_feat'][:, :12].detach()).abs().amax(1)
bad = torch.nonzero(e > 1e-3)[:, 0]
print('bad k0 rows', len(bad), bad[:10].tolist())
if len(bad):
    p = out['_ray_pts'][bad[:5]].detach()
    print('pts', p)
    lo, hi = torch.tensor(syn.XYZ_MIN), torch.tensor(syn.XYZ_MAX)
    print('u', (p - lo) / (hi - lo) * (torch.tensor(cfg.world_size).float() - 1))
# torch-GPU grid_sample on the same grid as a third opinion
k0g = eng.k0_reference_layout().contiguous()
ind = ((ws.pts[:M] - torch.tensor(syn.XYZ_MIN).cuda()) / (torch.tensor(syn.XYZ_MAX).cuda() - torch.tensor(syn.XYZ_MIN).cuda())).flip(-1) * 2 - 1
third = torch.nn.functional.grid_sample(k0g, ind.view(1, 1, 1, -1, 3), mode='bilinear', align_corners=True).view(12, -1).T
cmp('k0 hip-vs-gpu', ws.feat[:M, :12], third)
cmp('k0 cpu-vs-gpu', out['_rgb_feat'][:, :12], third)
