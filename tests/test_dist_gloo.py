"""N>1 path on CPU: world_size-2 and world_size-4 `gloo` rehearsal of the ray-sharded data-parallel step (poseprobe_amd.dist):
reduce-scatter of the dense grid gradient along X, ZeRO-1 sharded TV+Adam, all-gather of the updated slabs and the
single small all-reduce bucket.  The optimiser arithmetic here is the ORACLE's (no GPU in this container); what is
under test is the sharding/collective choreography that bench.py --gpus N runs with RCCL."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _tv_grad(k0, scale):
    p = k0.clone().requires_grad_(True)
    from oracle import voxurf_oracle as O
    (O.total_variation(p.permute(3, 0, 1, 2)[None]) * scale).backward()
    return p.grad


def _worker(rank, world, port, X, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from oracle import voxurf_oracle as O
    from poseprobe_amd.dist import DistContext, slab_bounds
    ctx = DistContext()
    g = torch.Generator().manual_seed(7)
    Y, Z, C = 5, 4, 12
    k0 = torch.randn(X, Y, Z, C, generator=g) * 0.1            # replicated parameters (channels-last)
    m, v = torch.zeros_like(k0), torch.zeros_like(k0)
    gr = torch.Generator().manual_seed(100 + rank)
    grad_local = torch.randn(X, Y, Z, C, generator=gr) * 1e-3  # this rank's ray shard contributes a different gradient
    small = [torch.randn(1000, generator=gr), torch.randn(3, 6, generator=gr)]
    small_local = [t.clone() for t in small]
    # ---- the choreography of TrainEngine.train_step
    grad = grad_local.clone()
    xb, xe = ctx.reduce_scatter_grid(grad)
    assert (xb, xe) == slab_bounds(X, world, rank)
    ctx.all_reduce_small(small)
    scale = 1.0 / world
    tv = _tv_grad(k0, 0.1 * 0.01)                               # rank invariant: never reduced
    gtot = grad[xb:xe] * scale + tv[xb:xe]
    O.adam_update(k0[xb:xe], gtot, m[xb:xe], v[xb:xe], 1, 0.1)
    ctx.all_gather_grid(k0)
    # ---- gather every rank's inputs on rank 0 for the single-process reference
    gl = [torch.empty_like(grad_local) for _ in range(world)]
    dist.all_gather(gl, grad_local)
    sl = [torch.empty_like(small_local[0]) for _ in range(world)]
    dist.all_gather(sl, small_local[0])
    if rank == 0:
        g0 = torch.Generator().manual_seed(7)
        k_ref = torch.randn(X, Y, Z, C, generator=g0) * 0.1
        m_ref, v_ref = torch.zeros_like(k_ref), torch.zeros_like(k_ref)
        gsum = sum(gl) * scale + _tv_grad(k_ref, 0.1 * 0.01)
        O.adam_update(k_ref, gsum, m_ref, v_ref, 1, 0.1)
        ok = torch.allclose(k0, k_ref, rtol=1e-6, atol=1e-7) and torch.allclose(small[0], sum(sl), rtol=1e-6, atol=1e-6)
        q.put(bool(ok))
    # every rank must hold identical parameters after the gather
    ks = [torch.empty_like(k0) for _ in range(world)]
    dist.all_gather(ks, k0)
    assert all(torch.equal(ks[0], k) for k in ks)
    # outside its slab a rank's grad buffer is zero (ready for the next step's accumulation)
    assert float(grad[:xb].abs().sum()) == 0 and float(grad[xe:].abs().sum()) == 0
    dist.destroy_process_group()


@pytest.mark.parametrize('X,world', [(8, 2), (6, 2), (8, 4)])
def test_sharded_step_equals_single_process(X, world):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, X, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def _sample_count(rank, world):
    """Samples a rank contributes to the exchange: ragged, and at world size 4 rank 2 has NO sample at all (all its rays miss the
    box) - its buffer still travels (row 0 carries the count 0) and contributes nothing to the replay."""
    return [20, 27, 0, 41][rank] if world == 4 else 20 + 7 * rank


def _worker_samples(rank, world, port, q):
    """mode "samples": every rank contributes a packed [cap,16] buffer (count in row 0, slot 15); after ONE all-gather
    each rank replays all shards into a dense gradient.  The replay here is a CPU stand-in (nearest-voxel deposit) for
    pp_k0_scatter_packed; what is under test is the collective and the count-in-band convention."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from poseprobe_amd.dist import DistContext
    ctx = DistContext(mode='samples')
    assert not ctx.local_scatter
    cap, G, C = 64, 6, 12
    g = torch.Generator().manual_seed(11 + rank)
    M = _sample_count(rank, world)                               # ragged: a different sample count per rank, one rank EMPTY at W = 4
    packed = torch.zeros(cap, 16)
    packed[:M, :C] = torch.randn(M, C, generator=g)
    packed[:M, 12:15] = torch.rand(M, 3, generator=g)            # positions in [0,1)^3
    packed[M:, :15] = 99.0                                        # stale rows past the count must be ignored
    packed[0, 15] = torch.tensor([M], dtype=torch.int32).view(torch.float32)[0]

    def replay(shards):
        grad = torch.zeros(G * G * G, C)
        for sh in shards:
            m = int(sh[0, 15:16].view(torch.int32)[0])
            ijk = (sh[:m, 12:15] * G).long().clamp_(0, G - 1)
            lin = (ijk[:, 0] * G + ijk[:, 1]) * G + ijk[:, 2]
            grad.index_add_(0, lin, sh[:m, :C])
        return grad

    out, _ = ctx.all_gather_rows(packed)
    assert out.shape == (world, cap, 16)
    mine = replay(out)
    allp = [torch.empty_like(packed) for _ in range(world)]
    dist.all_gather(allp, packed)
    ref = replay(allp)
    gs = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gs, mine)
    ok = torch.equal(mine, ref) and all(torch.equal(gs[0], x) for x in gs)
    counts = [int(out[r, 0, 15:16].view(torch.int32)[0]) for r in range(world)]
    ok = ok and counts == [_sample_count(r, world) for r in range(world)]
    # periodic resync primitive
    t = torch.full((4,), float(rank))
    ctx.broadcast_state([t])
    ok = ok and bool((t == 0).all())
    # the small bucket in its asynchronous form: values are only valid after wait_small()
    a, b = torch.full((5,), float(rank + 1)), torch.full((2, 3), 10.0 * (rank + 1))
    ctx.all_reduce_small([a, b], async_op=True)
    ctx.wait_small()
    tot = sum(range(1, world + 1))
    ok = ok and bool((a == tot).all()) and bool((b == 10.0 * tot).all())
    ctx.wait_small()                                             # idempotent
    # exact exchange size per step: every rank learns every rank's sample count right after its sampler and derives the same
    # row count from the largest one - also when a rank suddenly produces far more samples than in any earlier step
    for step, count_of in enumerate((lambda r: 3000 + 900 * r, lambda r: 15000 if r == world - 1 else 100, lambda r: 16384)):
        mask = torch.zeros(64)
        mask[:10 + 5 * rank + step] = 1.0
        ctx.start_batch_stats(torch.tensor([count_of(rank)], dtype=torch.int32), mask)
        counts = ctx.step_counts()
        ok = ok and [int(c) for c in counts] == [count_of(r) for r in range(world)]
        rows = ctx.exchange_rows(counts.max(), cap=16384)
        ok = ok and rows % 1024 == 0 and rows >= max(count_of(r) for r in range(world)) and rows <= 16384
        ok = ok and rows - max(count_of(r) for r in range(world)) < 1024
        bn = ctx.wait_batch_stats()
        ok = ok and abs(float(bn[1]) - sum(count_of(r) for r in range(world)) / world) < 1e-3
        ok = ok and abs(float(bn[0]) - sum(10 + 5 * r + step for r in range(world)) / world) < 1e-6
    ok = ok and ctx.exchange_rows(0, cap=16384) == 1024 and ctx.exchange_rows(5, cap=512) == 512
    if rank == 0:
        q.put(bool(ok))
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 4])
def test_sample_exchange_equals_single_process(world):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_samples, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_default_exchange_mode_per_world_size():
    from poseprobe_amd.dist import default_mode
    assert [default_mode(w) for w in (1, 2, 4, 8, 16)] == ['samples'] * 4 + ['zero1']


def test_slab_bounds_cover_grid():
    from poseprobe_amd.dist import slab_bounds
    for X, W in ((160, 8), (96, 4), (8, 2)):
        cover = []
        for r in range(W):
            b, e = slab_bounds(X, W, r)
            cover += list(range(b, e))
        assert cover == list(range(X))


def _worker_scene(rank, world, port, q):
    """Scene-branch data parallelism (joint.DualBranchEngine.train_step): each rank's packed gradient block is summed by
    DistContext.all_reduce_tensor and the Adam step divides by the world size; the replicas stay identical and equal one
    process that saw the mean gradient."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from poseprobe_amd.dist import DistContext
    ctx = DistContext()
    g = torch.Generator().manual_seed(3)
    p = torch.randn(1000, generator=g)
    grads = [torch.randn(1000, generator=torch.Generator().manual_seed(50 + r)) for r in range(world)]
    mine = grads[rank].clone()
    ctx.all_reduce_tensor(mine)
    opt_p = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([opt_p], lr=1e-3)
    opt_p.grad = mine / world                                   # == adam_flat(..., grad_scale = 1 / world)
    opt.step()
    q.put((rank, opt_p.detach().numpy().copy(), torch.stack(grads).mean(0).numpy().copy()))     # numpy: no shared-memory handles
    dist.barrier()
    dist.destroy_process_group()


def test_scene_gradient_all_reduce_keeps_replicas_identical():
    world, port = 2, _free_port()
    ctxm = mp.get_context('spawn')
    q = ctxm.Queue()
    procs = [ctxm.Process(target=_worker_scene, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert (res[0][1] == res[1][1]).all()
    ref_p = torch.randn(1000, generator=torch.Generator().manual_seed(3)).requires_grad_(True)
    ref = torch.optim.Adam([ref_p], lr=1e-3)
    ref_p.grad = torch.tensor(res[0][2])
    ref.step()
    assert torch.allclose(torch.tensor(res[0][1]), ref_p.detach(), rtol=0, atol=1e-7)


def _union_loss(out, target, mask, gs, n_iters, norm_mask, norm_samples, n_rays):
    """object_losses (oracle/voxurf_oracle.py:614-639 = lib/losses.py:34-74, tv weight 0) with explicit normalisers for the
    masked MSE (masked-pixel count) and the sample-level priors (sample count): the quantities pp_loss_rays /
    pp_geometry_bwd_priors take from `batch_norm` in the ray-sharded step."""
    from oracle import voxurf_oracle as O
    mse = (((out['rgb_marched'] - target) * mask) ** 2).sum() / (norm_mask * 3)
    pout = out['alphainv_cum'].clamp(1e-6, 1 - 1e-6)
    ent = -(pout * torch.log(pout) + (1 - pout) * torch.log(1 - pout)).sum() / n_rays
    eik = torch.abs(out['gradient'].norm(dim=-1) - 1).sum() / norm_samples
    w = O.dynamic_weight(1e-1, 1e-3, gs, n_iters)
    gd = out['grad_deform'].norm(dim=-1).sum() / (3 * norm_samples)
    sc = torch.abs(out['sdf_correct']).sum() / norm_samples
    sd = torch.abs(out['sdf_deform']).sum() / norm_samples
    bce = torch.nn.functional.binary_cross_entropy(out['cum_weights'].clip(1e-3, 1.0 - 1e-3), mask, reduction='sum') / n_rays
    return mse + 0.01 * ent + eik + w * (gd + sc + sd) + 0.1 * bce


def _worker_union(rank, world, port, q):
    """W ranks x N rays with DistContext's batch statistics == one process with the union batch of W*N rays: the gradients of
    the shared parameters (pose, alpha / beta, both MLPs) averaged over the ranks equal the union step's."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    from oracle import voxurf_oracle as O
    from poseprobe_amd import synthetic as syn
    from poseprobe_amd.dist import DistContext
    from tests.helpers import scene_for
    ctx = DistContext(mode='samples')
    G, V, H, W, N, gs = 12, 3, 16, 16, 48, 10
    scene = scene_for(G)
    views = syn.make_views(V, H, W)
    idx, jit = syn.step_randomness(V * H * W, N * world, seed=5)

    def grads_of(sel, norm=None):
        P = O.params_require_grad(O.init_params(scene, seed=2))
        se3 = torch.tensor(syn.se3_perturbation(V), requires_grad=True)
        c2w = O.pose_invert(O.current_pose_pnp(se3, torch.tensor(views['w2c'])))
        ro, rd, vd, target, mask = O.select_training_rays(torch.tensor(idx[sel]), torch.tensor(views['images']),
                                                          torch.tensor(views['masks']), torch.tensor(views['Ks']), c2w)
        out = O.voxurf_forward(P, scene, ro, rd, vd, jitter=torch.tensor(jit[sel]), global_step=gs)
        M = out['weights'].shape[0]
        if norm is None:
            nm, ns = mask.sum(), float(M)
        else:
            bn = norm(torch.tensor([M], dtype=torch.int32), mask)
            nm, ns = bn[0], bn[1]
        _union_loss(out, target, mask, gs, scene.N_iters, nm, ns, len(ro)).backward()
        flat = [se3.grad.reshape(-1), P['sdf_alpha'].grad, P['sdf_beta'].grad]
        flat += [t.grad.reshape(-1) for Wb in P['rgbnet'] + P['warp'] for t in Wb]
        return torch.cat(flat), M

    def norm(count, mask):
        ctx.start_batch_stats(count, mask)
        return ctx.wait_batch_stats()

    g_mine, M_mine = grads_of(slice(rank, None, world), norm)
    ctx.all_reduce_tensor(g_mine)
    g_mine /= world
    if rank == 0:
        g_union, M_union = grads_of(slice(None))
        # and the oracle's own object_losses on the union batch is the same function
        err = float((g_mine - g_union).abs().max() / g_union.abs().max())
        q.put((err, int(ctx.step_counts().sum()), M_union))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_losses_equal_the_union_batch():
    world, port = 2, _free_port()
    ctxm = mp.get_context('spawn')
    q = ctxm.Queue()
    procs = [ctxm.Process(target=_worker_union, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    err, m_sum, m_union = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert m_sum == m_union
    assert err < 2e-5, err


def test_union_loss_helper_is_the_oracle_loss():
    """_union_loss with the batch's own counts == oracle object_losses (which is pinned to lib/losses.py by the fixtures)."""
    from oracle import voxurf_oracle as O
    from poseprobe_amd import synthetic as syn
    from tests.helpers import scene_for
    scene = scene_for(12)
    views = syn.make_views(3, 16, 16)
    idx, jit = syn.step_randomness(3 * 16 * 16, 64, seed=5)
    P = O.params_require_grad(O.init_params(scene, seed=2))
    se3 = torch.tensor(syn.se3_perturbation(3), requires_grad=True)
    c2w = O.pose_invert(O.current_pose_pnp(se3, torch.tensor(views['w2c'])))
    ro, rd, vd, target, mask = O.select_training_rays(torch.tensor(idx), torch.tensor(views['images']),
                                                      torch.tensor(views['masks']), torch.tensor(views['Ks']), c2w)
    out = O.voxurf_forward(P, scene, ro, rd, vd, jitter=torch.tensor(jit), global_step=10)
    ref = O.object_losses(out, target, mask, 10, scene.N_iters, weight_tv_k0=0.0)[2]
    mine = _union_loss(out, target, mask, 10, scene.N_iters, mask.sum(), float(out['weights'].shape[0]), len(ro))
    assert abs(float(ref) - float(mine)) < 1e-6 * abs(float(ref))
