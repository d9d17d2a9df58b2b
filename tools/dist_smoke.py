"""Two ranks on ONE GPU (gloo backend, CUDA tensors): runs the real multi-GPU choreography of poseprobe_amd.dist with the HIP
kernels - pack -> all-gather -> replayed scatter -> replicated optimiser ("samples") or reduce / slab Adam / gather ("zero1")
- and checks that both replicas end up with the same parameters and that they track a single-process run over the union of
the rays.   python tools/dist_smoke.py [samples|zero1] [det]"""
import os, sys, socket
import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


DET = len(sys.argv) > 2 and sys.argv[2] == 'det'      # deterministic (sorted) k0 scatter: replicas bit-identical in "samples" mode


def build(N, dctx=None):
    from poseprobe_amd import synthetic as syn
    from poseprobe_amd.engine import SceneConfig, TrainEngine
    from poseprobe_amd.params_init import reference_like_params
    G, H, W, V = 32, 64, 64, 3
    rs = syn.range_shape()
    cfg = SceneConfig(syn.XYZ_MIN, syn.XYZ_MAX, G ** 3, out_range=float(rs.max()))
    views = syn.make_views(V, H, W)
    eng = TrainEngine(cfg, V, H, W, N, pose_iters=1000, dist_ctx=dctx, deterministic_scatter=DET)
    eng.set_views(views['images'], views['masks'], views['Ks'], views['w2c'])
    P = reference_like_params(cfg, 3)
    eng.load_reference_params(P['k0'], P['sdf'], P['sdf_alpha'], P['sdf_beta'], P['rgbnet'], P['warp'],
                              se3=torch.tensor(syn.se3_perturbation(V)))
    eng.zero_grads()
    return eng, V * H * W


def steps(eng, total, N, world, rank, n=3):
    from poseprobe_amd import synthetic as syn
    for s in range(n):
        idx, jit = syn.step_randomness(total, N * world, seed=300 + s)
        eng.train_step(torch.tensor(idx[rank::world], dtype=torch.int32, device='cuda'),
                       torch.tensor(jit[rank::world], device='cuda'), 10 + s)
    torch.cuda.synchronize()


def first_step_gradients(eng, total, N, world, rank):
    """Gradients of the shared parameters for step 0's batch, averaged over the ranks as the optimiser will see them
    (no optimiser step): k0 [X,Y,Z,C], small-parameter block, se3."""
    from poseprobe_amd import synthetic as syn
    idx, jit = syn.step_randomness(total, N * world, seed=300)
    eng.render_and_grads(torch.tensor(idx[rank::world], dtype=torch.int32, device='cuda'),
                         torch.tensor(jit[rank::world], device='cuda'), 10)
    scale = 1.0
    if eng.dist is not None:
        eng.dist.reduce_gradients(eng)
        eng.dist.wait_small()
        scale = eng.grad_scale
    torch.cuda.synchronize()
    out = [(t * scale).detach().cpu() for t in (eng.k0_grad, eng.flat.grad, eng.se3_grad)]
    eng.zero_grads()
    return out


def worker(rank, world, port, mode, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from poseprobe_amd.dist import DistContext
    eng, total = build(256, DistContext(mode=mode, resync_every=0))
    g = first_step_gradients(eng, total, 256, world, rank)
    if mode == 'zero1':                     # every rank holds its own x-slab of the reduced grid gradient: assemble
        parts = [None] * world
        dist.all_gather_object(parts, g[0])
        g[0] = sum(parts)
    eng, total = build(256, DistContext(mode=mode, resync_every=0))
    steps(eng, total, 256, world, rank)
    k0 = eng.k0_cl.detach().cpu()
    flat = eng.flat.data.detach().cpu()
    se3 = eng.se3.detach().cpu()
    outs = [None] * world
    dist.all_gather_object(outs, (k0, flat, se3))
    if rank == 0:
        # numpy: pickled by value (torch tensors travel as file descriptors that die with this process)
        q.put(([[t.numpy() for t in o] for o in outs], [t.numpy() for t in g]))
    dist.destroy_process_group()


if __name__ == '__main__':
    mode = sys.argv[1] if len(sys.argv) > 1 else 'samples'
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, 2, port, mode, q)) for r in range(2)]      # (spawned children re-read sys.argv: DET carries over)
    for p in procs: p.start()
    outs, g_sharded = q.get(timeout=240)
    for p in procs: p.join(60)
    (k0a, fa, sa), (k0b, fb, sb) = [[torch.from_numpy(t) for t in o] for o in outs]
    g_sharded = [torch.from_numpy(t) for t in g_sharded]
    print(f'[{mode}] replicas: max|k0 diff| {float((k0a - k0b).abs().max()):.3e}  max|mlp diff| {float((fa - fb).abs().max()):.3e}  '
          f'max|se3 diff| {float((sa - sb).abs().max()):.3e}')
    # single process over the union of the rays (global batch 2 x 256).  The sharded step normalises its losses over the union
    # batch (DistContext.start_batch_stats), so step 0's averaged gradients must EQUAL the union step's up to fp32 summation
    # order (float atomics): rtol 1e-3 + 5e-5 of the largest entry, the tolerance of the single-GPU gradient tests
    eng, total = build(512)
    g_union = first_step_gradients(eng, total, 512, 1, 0)
    grads_ok = True
    for name, a, b in zip(('k0', 'mlp/alpha/beta', 'se3'), g_sharded, g_union):
        err = (a - b).abs()
        bad = err > 1e-3 * b.abs() + 5e-5 * b.abs().max()
        print(f'[{mode}] step-0 gradient {name}: max abs err {float(err.max()):.3e} (max |ref| {float(b.abs().max()):.3e}), '
              f'{int(bad.sum())} of {bad.numel()} outside tolerance')
        grads_ok = grads_ok and not bool(bad.any())
    eng, total = build(512)
    steps(eng, total, 512, 1, 0)
    k0s = eng.k0_cl.detach().cpu()
    frac = float(((k0a - k0s).abs() > 1e-3).float().mean())
    print(f'[{mode}] vs single process on the union batch: {frac:.2%} of k0 entries differ by more than 1e-3, '
          f'se3 max diff {float((sa - eng.se3.detach().cpu()).abs().max()):.3e}')
    if DET and mode == 'samples':
        print(f'[{mode}] deterministic scatter: replicas bit-identical: {bool(torch.equal(k0a, k0b))}')
        grads_ok = grads_ok and bool(torch.equal(k0a, k0b))
    ok = grads_ok and float((k0a - k0b).abs().max()) < 1e-4 and float((fa - fb).abs().max()) < 1e-4 and frac < 0.02
    print('OK' if ok else 'MISMATCH')
    sys.exit(0 if ok else 1)
