#!/usr/bin/env python
"""bench.py - rays/s of one object-branch TRAIN STEP (ray select -> render -> loss -> backward -> optimiser step) on
the BASELINE.json workload: DTU-scan1-like 3 views of 400x400, 160^3 grid, 186 samples/ray, N_rand = 1024 rays per
GPU (weak scaling: global batch = 1024 * n_gpus), fp32, synthetic inputs resident in HBM.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (N > 1)

Prints ONE JSON line on rank 0 (see the driver contract).  Extra objects:
  roofline     - the dominant kernel (fused TV+Adam pass over the dense k0 grid, HBM bound): algorithmic bytes per
                 launch (384 B/voxel) / mean launch duration measured with HIP events on the launch stream
  cpu_baseline - the oracle (CPU restatement, "port") timed on this host's cores for a bounded sample (N=1, rank 0)
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from poseprobe_amd import synthetic as syn            # noqa: E402
from poseprobe_amd.engine import SceneConfig, TrainEngine   # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X spec (MI355X_MICROARCH.md: 8 TB/s spec, ~6.3 TB/s achievable)
GRID_BYTES_PER_VOXEL = 384  # fused pass, C=12 fp32: read p,g,m,v (192) + write p',m,v,g=0 (192)


def pmc_traffic(grid, voxels, sparse):
    """HBM bytes per launch of k_grid_tv_adam from the committed rocprofv3 PMC passes (profiles/r01_grid_traffic*.json:
    FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, KB -> bytes), only when it was collected for this grid size and for
    this variant of the pass (dense / sparse-gradient)."""
    try:
        name = 'r01_grid_traffic_sparse.json' if sparse else 'r01_grid_traffic.json'
        rec = json.load(open(os.path.join(ROOT, 'profiles', name)))
        if int(rec['grid']) == int(grid) and int(rec['voxels_per_launch']) == int(voxels):
            return float(rec['hbm_bytes_per_launch'])
    except Exception:
        pass
    return None


def init_engine_params(eng, cfg, seed):
    """Random-init parameters of the reference's architecture (cube-init SDF, k0~N(0,.1), warp last layer N(0,1e-2))."""
    from poseprobe_amd.params_init import reference_like_params
    P = reference_like_params(cfg, seed)
    eng.load_reference_params(P['k0'], P['sdf'], P['sdf_alpha'], P['sdf_beta'], P['rgbnet'], P['warp'],
                              se3=torch.tensor(syn.se3_perturbation(eng.V)))
    return P


def cpu_baseline(G, H, W, V, n_rand, views, budget_s=20.0):
    """Oracle train step on the host cores: bounded sample of the SAME workload.  torch-CPU stops scaling (and then
    degrades: 128 threads are 3x slower than 32 for these op sizes) beyond ~32 threads, so at most 32 are used."""
    from oracle import voxurf_oracle as O
    threads_before = torch.get_num_threads()
    torch.set_num_threads(min(32, threads_before))
    rs = syn.range_shape()
    scene = O.Scene(syn.XYZ_MIN, syn.XYZ_MAX, G ** 3, output_range=float(rs.max()), rect_size=rs.tolist())
    P = O.init_params(scene, seed=3)
    st = O.TrainState(P, scene, torch.tensor(views['w2c']), torch.tensor(views['Ks']), torch.tensor(views['images']),
                      torch.tensor(views['masks']), se3_refine=torch.tensor(syn.se3_perturbation(V)), pose_iters=1000)
    times = []
    t_start = time.time()
    s = 0
    while True:
        idx, jit = syn.step_randomness(V * H * W, n_rand, seed=1000 + s)
        t0 = time.time()
        st.step(torch.tensor(idx), torch.tensor(jit), 10 + s)
        times.append(time.time() - t0)
        s += 1
        if s >= 2 and (time.time() - t_start > budget_s or s >= 12):
            break
    t = float(np.median(times[1:])) if len(times) > 1 else times[0]
    used = torch.get_num_threads()
    scene = cpu_baseline_scene(V, used)
    torch.set_num_threads(threads_before)
    return {'value': n_rand / t, 'unit': 'rays/s', 'cores': used, 'kind': 'port',
            'sample': f'{len(times)} oracle train steps (torch-CPU fp32, {n_rand} rays, {G}^3 grid, first step dropped), '
                      f'median {t:.3f} s/step', 'scene_branch': scene}


def cpu_baseline_scene(V, threads, n_pix=341, S=128, reps=3):
    """The scene branch's coarse pass (forward, loss, backward) of the oracle on the same host cores, beside `dual_branch`."""
    from oracle import scene_nerf as SN
    torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(0)
    R = V * n_pix
    P = {}
    dims = [(256, 63)] + [(256, 256)] * 3 + [(256, 319)] + [(256, 256)] * 2 + [(257, 256)]
    for i, (o, k) in enumerate(dims):
        P[f'mlp_feat.{i}.weight'] = (torch.randn(o, k, generator=g) * (2.0 / k) ** 0.5).requires_grad_(True)
        P[f'mlp_feat.{i}.bias'] = torch.zeros(o, requires_grad=True)
    for i, (o, k) in enumerate([(128, 283), (3, 128)]):
        P[f'mlp_rgb.{i}.weight'] = (torch.randn(o, k, generator=g) * (2.0 / k) ** 0.5).requires_grad_(True)
        P[f'mlp_rgb.{i}.bias'] = torch.zeros(o, requires_grad=True)
    center, ray = torch.randn(R, 3, generator=g) * 0.3, torch.randn(R, 3, generator=g)
    depth = (torch.rand(R, S, generator=g) + torch.arange(S)) / S * 2.5 + 0.5
    image = torch.rand(R, 3, generator=g)
    times = []
    for _ in range(reps):
        t0 = time.time()
        loss = SN.photometric_loss(SN.render(P, center, ray, depth, 0.5, (0.4, 0.7))['rgb'], image)
        loss.backward()
        times.append(time.time() - t0)
    t = float(np.median(times[1:])) if len(times) > 1 else times[0]
    return {'value': R / t, 'unit': 'rays/s', 'cores': threads, 'kind': 'port',
            'sample': f'{reps} oracle coarse passes (torch-CPU fp32, {R} rays x {S} samples, forward + loss + backward, first '
                      f'dropped), median {t:.3f} s'}


def dual_branch_leg(eng, idx_all, jit_all, gs, N, V, H, W, object_ms, dev, steps=20, warmup=3):
    """Informative second measurement (never `value`): the same object-branch step plus the scene branch of the reference's
    joint loop (lib/recon_scene.py:639-649): rand_rays // V pixels per view x 128 stratified samples through the 8 x 256
    NeRF, 2 * huber loss, backward, Adam, poses shared through the object engine's pose Jacobian - first the coarse-only phase
    (the first 30 % of the schedule), then the hierarchical phase (coarse + fine network on 128 + 128 samples)."""
    from poseprobe_amd import bg_nerf
    from poseprobe_amd.joint import DualBranchEngine
    opt = bg_nerf.default_options(sample_intvs=128)
    opt.nerf.fine_sampling, opt.nerf.sample_intvs_fine = True, 128
    torch.manual_seed(0)
    nets = [bg_nerf.NeRF(opt, device=dev), bg_nerf.NeRF(opt, is_fine_network=True, device=dev)]
    for n in nets:
        n.progress.data.fill_(0.5)
    joint = DualBranchEngine(eng, nets[0], lr_scene=1e-3, depth_range=(0.5, 3.0), scene_net_fine=nets[1])
    n_pix, S = opt.nerf.rand_rays // V, 128
    g = torch.Generator().manual_seed(1)
    px = [(torch.rand(n_pix, 2, generator=g) * torch.tensor([W - 1., H - 1.])).to(dev) for _ in range(steps + warmup)]
    img = torch.rand(V, n_pix, 3, generator=g).to(dev)
    n_avail = idx_all.shape[0]
    fl_sample = 2 * (64 * 256 + 6 * 256 * 256 + 320 * 256 + 256 + 288 * 128 + 128 * 3)       # forward FLOP per sample
    out = {'workload': f'object-branch step + scene branch: {V} x {n_pix} rays, 8x256 NeRF (BARF PE), 2*huber loss, backward, '
                       f'Adam; shared poses, loss = 0.1 L_obj + L_bg; coarse phase = {S} stratified samples, hierarchical '
                       f'phase = coarse + fine network on {S}+{S} samples', 'object_rays': N, 'scene_rays': V * n_pix, 'n_gpus': 1,
           'scene_arithmetic': 'fp32 operands and accumulation; forward / data-gradient products as 3 fp16 MFMA products (error vs '
                               'fp64 equal to the fp32 MFMA path), weight-gradient products on fp32 MFMA; PP_NERF_SPLIT=0: fp32 MFMA only'}
    for label, fine, samples in (('coarse_phase', False, V * n_pix * S), ('hierarchical_phase', True, V * n_pix * 3 * S)):
        for s in range(warmup):
            joint.train_step(idx_all[s % n_avail], jit_all[s % n_avail], gs + s, px[s], img, fine=fine)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in range(warmup, warmup + steps):
            joint.train_step(idx_all[s % n_avail], jit_all[s % n_avail], gs + s, px[s], img, fine=fine)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        scene_ms = ms - object_ms
        tf = 3 * fl_sample * samples / (scene_ms * 1e-3) / 1e12                                # fwd + data grad + weight grad
        out[label] = {'ms_per_step': ms, 'rays_per_s': (N + V * n_pix) / (ms * 1e-3), 'scene_samples': samples,
                      'scene_ms': scene_ms, 'scene_algorithmic_tflops': tf,
                      'scene_tflops_over_fp32_mfma_peak': tf / 157.3}      # informative: the products run as 3 fp16 MFMAs each
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--grid', type=int, default=160)
    ap.add_argument('--n-rand', type=int, default=1024)
    ap.add_argument('--hw', type=int, default=400)
    ap.add_argument('--views', type=int, default=3)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-budget', type=float, default=20.0)
    ap.add_argument('--no-dual', action='store_true', help='skip the informative dual-branch (object + scene) leg')
    args = ap.parse_args()

    # stdout carries exactly ONE line (the JSON record).  Libraries that print to fd 1 (RCCL prints a version banner when
    # a communicator is created) are sent to stderr for the duration of the run.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    dctx = None
    use_dist = world > 1 or os.environ.get('PP_FORCE_DIST') == '1'     # PP_FORCE_DIST: rehearse the RCCL path on 1 rank
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        from poseprobe_amd.dist import DistContext
        dctx = DistContext()
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'

    G, H, W, V, N = args.grid, args.hw, args.hw, args.views, args.n_rand
    rs = syn.range_shape()
    cfg = SceneConfig(syn.XYZ_MIN, syn.XYZ_MAX, G ** 3, out_range=float(rs.max()))
    views = syn.make_views(V, H, W)
    eng = TrainEngine(cfg, V, H, W, N, device=dev, pose_iters=3000, dist_ctx=dctx)
    eng.set_views(views['images'], views['masks'], views['Ks'], views['w2c'])
    init_engine_params(eng, cfg, seed=3)
    eng.zero_grads()

    # every step's randomness is generated up front and is resident on the device: rank r takes its own ray shard
    total = args.steps + args.warmup
    extra = min(10, args.steps)          # untimed post-pass that prices the MLP chains (second, informative roofline)
    idx_all, jit_all = [], []
    for s in range(total + extra):
        idx, jit = syn.step_randomness(V * H * W, N * world, seed=2000 + s)
        idx_all.append(idx[rank::world])
        jit_all.append(jit[rank::world])
    idx_all = torch.tensor(np.stack(idx_all), dtype=torch.int32, device=dev)
    jit_all = torch.tensor(np.stack(jit_all), dtype=torch.float32, device=dev)

    def barrier():
        if use_dist:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # time the dominant kernel with HIP events on the launch stream (torch's current stream == our launch stream)
    from poseprobe_amd import ops
    ev = []
    def timed_grid_step(fn):
        def wrapper(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn(*a, **k)
            e1.record()
            ev.append((e0, e1))
        return wrapper

    ops.grid_tv_adam_step = timed_grid_step(ops.grid_tv_adam_step)                   # dense pass (ZeRO-1 slabs)
    ops.grid_tv_adam_step_sparse = timed_grid_step(ops.grid_tv_adam_step_sparse)     # same kernel with the touched-voxel map
    # second roofline: the two MLPs on the matrix cores (fp32 MFMA).  Every event pair drains the launch pipeline, so
    # these four extra pairs per step are taken in a short pass AFTER the timed region, not inside it.
    mlp_ev = []

    def timed(fn):
        def wrapper(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn(*a, **k)
            e1.record()
            mlp_ev.append((e0, e1))
        return wrapper

    gs = 10
    for s in range(args.warmup):
        eng.train_step(idx_all[s], jit_all[s], gs + s)
    barrier()
    ev.clear()
    t0 = time.perf_counter()
    for s in range(args.warmup, total):
        eng.train_step(idx_all[s], jit_all[s], gs + s)
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    M = int(eng.ws.count.item())
    mlp_names = ('warp_fwd', 'warp_bwd', 'rgbnet_fwd', 'rgbnet_bwd')
    mlp_orig = {n: getattr(ops, n) for n in mlp_names}
    for n in mlp_names:
        setattr(ops, n, timed(mlp_orig[n]))
    n_ev = len(ev)
    for s in range(total, total + extra):
        eng.train_step(idx_all[s], jit_all[s], gs + s)
    barrier()
    for n in mlp_names:
        setattr(ops, n, mlp_orig[n])
    del ev[n_ev:]                                    # the grid kernel is priced over the timed region only
    grid_ms = float(np.mean([a.elapsed_time(b) for a, b in ev])) if ev else float('nan')
    xb, xe = eng.x_slab
    X, Y, Z = cfg.world_size
    # algorithmic bytes of the pass: p, m, v read + p', m, v written for every voxel (288 B at C=12), the gradient read
    # + re-zeroed only for the voxels the scatter marked (96 B x marked fraction; fraction = 1 for the dense pass)
    marked = 1.0
    if (xb, xe) == (0, X) and not (dctx is not None and dctx.local_scatter):
        marked = float(eng.k0_touched[1 - eng.touch_par].ne(0).sum().item()) / (X * Y * Z)   # the map the last step consumed
    per_voxel = GRID_BYTES_PER_VOXEL * (0.75 + 0.25 * marked)
    grid_bytes = per_voxel * (xe - xb) * Y * Z
    achieved = grid_bytes / (grid_ms * 1e-3) / 1e9
    # MFMA-shaped work per sample: warp hidden GEMMs 3 layers x 4 rows x 2*128*128 x (fwd + 2 bwd),
    # rgbnet (64*128 + 2*128*128) x 2 x (fwd + 2 bwd)
    flop_per_sample = 3 * (3 * 4 * 2 * 128 * 128) + 3 * (2 * (64 * 128 + 2 * 128 * 128))
    mlp_ms = float(np.sum([a.elapsed_time(b) for a, b in mlp_ev])) / max(extra, 1) if mlp_ev else float('nan')
    mlp_tflops = flop_per_sample * M / (mlp_ms * 1e-3) / 1e12

    dual = None
    if world == 1 and not args.no_dual:
        dual = dual_branch_leg(eng, idx_all, jit_all, gs, N, V, H, W, dt / args.steps * 1e3, dev)

    if rank == 0:
        out = {
            'metric': 'rays_per_sec_train_step', 'value': N * world * args.steps / dt, 'unit': 'rays/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'DTU-scan1-like {V}-view {H}x{W}, object-branch train step (ray select, render, '
                                   f'losses, backward, TV+Adam), {G}^3 grid, {cfg.n_samples} samples/ray, '
                                   f'N_rand={N}/GPU', 'grid': G, 'n_rand_per_gpu': N, 'samples_in_bbox_last_step': M,
                       'parallelism': (f'ray-sharded dp{world}, k0 gradient exchanged per sample (all-gather), replicated grid optimiser' if (dctx is None or dctx.mode == 'samples') else f'ray-sharded dp{world}, dense reduce-scatter + ZeRO-1 grid optimiser') if world > 1 else 'single GPU'},
            'roofline': {'bound': 'hbm', 'kernel': 'k_grid_tv_adam (fused TV-grad + Adam + zero-grad over k0, gradient touched only where the scatter marked)',
                         'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                         'traffic': pmc_traffic(G, (xe - xb) * Y * Z, marked < 1.0), 'ms_per_launch': grid_ms, 'algorithmic_bytes_per_launch': grid_bytes,
                         'grad_voxels_marked': marked},
            'roofline_mfma': {'bound': 'mfma', 'kernel': 'warp + rgbnet MLP chains (layer-fused fwd / bwd-data / weight-gradient kernels, fp32 MFMA 32x32x2)',
                              'achieved': mlp_tflops, 'peak': 157.3, 'unit': 'TFLOP/s', 'frac': mlp_tflops / 157.3,
                              'ms_per_step': mlp_ms, 'flop_per_sample': flop_per_sample},
        }
        out['dual_branch'] = dual
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(G, H, W, V, N, views, args.cpu_budget)
        else:
            out['cpu_baseline'] = None
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + '\n').encode())
    if use_dist:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
