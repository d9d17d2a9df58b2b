"""GPU parity tests, kernel level: every test calls the HIP path through the C ABI (poseprobe_amd.ops) and checks
it against the oracle (CPU restatement) or the golden vectors produced by the reference."""
import numpy as np
import pytest
import torch

from tests.helpers import assert_close, load, scene_for

pytestmark = pytest.mark.gpu


def _cfg(G, **kw):
    from poseprobe_amd import synthetic as syn
    from poseprobe_amd.engine import SceneConfig
    rs = syn.range_shape()
    return SceneConfig(syn.XYZ_MIN, syn.XYZ_MAX, int(G) ** 3, out_range=float(rs.max()), **kw)


def cu(x, dtype=torch.float32):
    return torch.as_tensor(np.asarray(x)).to('cuda', dtype).contiguous()


def test_library_loaded_and_fails_loudly_without_gpu_tensors():
    from poseprobe_amd import _lib, ops
    assert _lib.lib().pp_abi_version() == 3
    a = torch.zeros(4)
    with pytest.raises(RuntimeError):
        ops.alpha2weight_fwd(a, a, 1, a, a, a, a)


def test_pose_fwd_bwd_matches_reference_golden():
    from poseprobe_amd import ops
    d = load('pose.npz')
    V = d['wu'].shape[0]
    se3, init = cu(d['wu']), cu(d['init'])
    w2c, c2w = torch.empty(V, 3, 4, device='cuda'), torch.empty(V, 3, 4, device='cuda')
    jac = torch.empty(V, 12, 6, device='cuda')
    ops.pose_fwd(se3, init, None, w2c, c2w, jac)
    assert_close(w2c.cpu(), d['composed'], rtol=1e-6, atol=1e-6, name='w2c')
    assert_close(c2w.cpu(), d['inverted'], rtol=1e-6, atol=1e-6, name='c2w')
    g = torch.empty(V, 6, device='cuda')
    ops.pose_bwd(jac, cu(d['wsum']), g)
    assert_close(g.cpu(), d['grad_wu'], rtol=1e-4, atol=1e-6, name='grad_se3')


def test_raygen_selected_pixels_bit_exact():
    from poseprobe_amd import ops
    d = load('rays.npz')
    H, W, V = 8, 12, 3
    cfg = _cfg(8)
    Ks = d['Ks']
    intr = cu(np.stack([Ks[:, 0, 0], Ks[:, 1, 1], Ks[:, 0, 2], Ks[:, 1, 2]], -1))
    c2w = cu(d['c2w'][:, :3, :4])
    idx = torch.arange(V * H * W, dtype=torch.int32, device='cuda')
    N = idx.numel()
    for inv_y in (True, False):
        for normalize, pre in ((True, 'vox'), (False, 'dvgo')):
            o, dd, vd = (torch.empty(N, 3, device='cuda') for _ in range(3))
            ops.raygen_select_fwd(cfg.pp, idx, c2w, intr, H, W, inv_y, normalize, None, None, o, dd, vd, None, None)
            for v in range(V):
                tag = f'v{v}_invy{int(inv_y)}'
                sl = slice(v * H * W, (v + 1) * H * W)
                assert np.array_equal(dd[sl].cpu().numpy().reshape(H, W, 3), d[f'{pre}_d_{tag}']), (pre, tag, 'd')
                assert np.array_equal(vd[sl].cpu().numpy().reshape(H, W, 3), d[f'{pre}_v_{tag}']), (pre, tag, 'v')
                assert np.array_equal(o[sl].cpu().numpy().reshape(H, W, 3), d[f'{pre}_o_{tag}']), (pre, tag, 'o')


@pytest.mark.parametrize('tag', ['g8', 'g24'])
def test_dense_sampler_indices_bit_exact(tag):
    """ray/sample indices, steps, t_min/t_max AND sample positions bit-exact against the reference's sample_ray_ori."""
    from poseprobe_amd import ops
    d = load(f'sampler_{tag}.npz')
    cfg = _cfg(d['G'])
    ro, rd = cu(d['rays_o']), cu(d['rays_d'])
    N, S = ro.shape[0], cfg.n_samples
    assert S == d['mask_out_train'].shape[1]
    for sfx, jit in (('train', cu(d['jitter'])), ('eval', None)):
        cap = N * S
        f, i = dict(device='cuda'), dict(device='cuda', dtype=torch.int32)
        t_min, t_max = torch.empty(N, **f), torch.empty(N, **f)
        ray_start, count = torch.empty(N + 1, **i), torch.empty(1, **i)
        pts, ray_id, step_k, step = torch.empty(cap, 3, **f), torch.empty(cap, **i), torch.empty(cap, **i), torch.empty(cap, **f)
        keep = torch.empty(N * S, device='cuda', dtype=torch.uint8)
        ops.sample_dense(cfg.pp, ro, rd, jit, cap, t_min, t_max, ray_start, count, pts, ray_id, step_k, step, keep)
        M = int(count.item())
        ref_keep = ~d[f'mask_out_{sfx}']
        assert np.array_equal(keep.cpu().numpy().reshape(N, S).astype(bool), ref_keep)
        rid, sk = np.nonzero(ref_keep)
        assert M == len(rid)
        assert np.array_equal(ray_id[:M].cpu().numpy(), rid)
        assert np.array_equal(step_k[:M].cpu().numpy(), sk)
        assert np.array_equal(step[:M].cpu().numpy(), d[f'step_{sfx}'][ref_keep])
        assert np.array_equal(pts[:M].cpu().numpy(), d[f'pts_{sfx}'][ref_keep])
        assert np.array_equal(t_min.cpu().numpy(), d[f't_min_{sfx}'])
        assert np.array_equal(t_max.cpu().numpy(), d[f't_max_{sfx}'])
        assert int(ray_start[-1].item()) == M


def test_dense_sampler_with_capacity_below_the_total_stays_inside_the_allocation():
    """capacity < number of in-bbox samples: count and every ray_start entry are clamped, nothing past `capacity` rows is
    written (guard rows keep their sentinel), rays in front of the cut are unchanged, and the per-ray kernels that walk
    [ray_start[r], ray_start[r+1]) (transmittance scan forward / backward) stay inside the allocation."""
    from poseprobe_amd import ops
    d = load('sampler_g24.npz')
    cfg = _cfg(d['G'])
    ro, rd, jit = cu(d['rays_o']), cu(d['rays_d']), cu(d['jitter'])
    N, S = ro.shape[0], cfg.n_samples
    f, i = dict(device='cuda'), dict(device='cuda', dtype=torch.int32)

    def run(cap, rows):
        t_min, t_max = torch.empty(N, **f), torch.empty(N, **f)
        ray_start, count = torch.empty(N + 1, **i), torch.empty(1, **i)
        pts, step = torch.full((rows, 3), -777., **f), torch.full((rows,), -777., **f)
        ray_id, step_k = torch.full((rows,), -7, **i), torch.full((rows,), -7, **i)
        ops.sample_dense(cfg.pp, ro, rd, jit, cap, t_min, t_max, ray_start, count, pts, ray_id, step_k, step)
        return ray_start.cpu().numpy(), int(count.item()), pts.cpu().numpy(), ray_id.cpu().numpy(), step_k.cpu().numpy(), step.cpu().numpy()

    rs_full, M, pts_f, rid_f, sk_f, st_f = run(N * S, N * S)
    cap = max(N, M * 2 // 3)
    assert N <= cap < M
    rs, cnt, pts, rid, sk, st = run(cap, cap + 4096)
    assert cnt == cap and rs[-1] == cap
    assert np.array_equal(rs, np.minimum(rs_full, cap))
    assert (pts[cap:] == -777.).all() and (rid[cap:] == -7).all() and (sk[cap:] == -7).all() and (st[cap:] == -777.).all()
    assert np.array_equal(pts[:cap], pts_f[:cap]) and np.array_equal(rid[:cap], rid_f[:cap])
    assert np.array_equal(sk[:cap], sk_f[:cap]) and np.array_equal(st[:cap], st_f[:cap])
    # consumers of ray_start: buffers have exactly `cap` rows + guard rows that must keep their sentinel
    g = torch.Generator().manual_seed(0)
    alpha = torch.full((cap + 4096,), -777., **f)
    alpha[:cap] = (torch.rand(cap, generator=g) * 0.3).cuda()
    w, T = torch.full((cap + 4096,), -777., **f), torch.full((cap + 4096,), -777., **f)
    last, i_end = torch.empty(N, **f), torch.empty(N, **i)
    rs_d = torch.tensor(rs, **i)
    ops.alpha2weight_fwd(alpha, rs_d, N, w, T, last, i_end)
    gw, gl, ga = torch.ones(cap + 4096, **f), torch.zeros(N, **f), torch.full((cap + 4096,), -777., **f)
    ops.alpha2weight_bwd(alpha, w, T, last, rs_d, i_end, N, gw, gl, ga)
    torch.cuda.synchronize()
    assert float(w[cap:].max()) == -777. and float(T[cap:].max()) == -777. and float(ga[cap:].max()) == -777.
    assert bool((i_end <= cap).all())
    full_ray = np.nonzero(rs_full[1:] <= cap)[0]                 # rays in front of the cut: same result as an untruncated run
    w2, T2 = torch.zeros(M, **f), torch.zeros(M, **f)
    alpha2 = torch.zeros(M, **f)
    alpha2[:cap] = alpha[:cap]
    last2, i_end2 = torch.empty(N, **f), torch.empty(N, **i)
    ops.alpha2weight_fwd(alpha2, torch.tensor(rs_full, **i), N, w2, T2, last2, i_end2)
    assert torch.equal(last[full_ray], last2[full_ray])
    e = int(rs_full[full_ray[-1] + 1])
    assert torch.equal(w[:e], w2[:e])


def _random_segments(n_rays, max_len, seed, hot=False):
    rng = np.random.RandomState(seed)
    lens = rng.randint(0, max_len, size=n_rays)
    lens[rng.rand(n_rays) < 0.1] = 0            # empty rays
    lens[:3] = [1, 64, 65]                      # single sample, exact wave, wave+1
    ray_id = np.repeat(np.arange(n_rays), lens)
    M = int(lens.sum())
    alpha = rng.rand(M).astype(np.float32) * (0.6 if hot else 0.05)
    alpha[rng.rand(M) < 0.02] = 0.0
    if hot:
        alpha[rng.rand(M) < 0.01] = 1.0
    start = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    return alpha, ray_id.astype(np.int64), start, M


@pytest.mark.parametrize('hot', [False, True])
def test_alpha2weight_fwd_bwd_exact(hot):
    """Sequential-order wave scan reproduces the oracle's (and the .cu's) arithmetic exactly, incl. the 1e-3 early stop."""
    from oracle import native_ops
    from poseprobe_amd import ops
    N = 300
    alpha, ray_id, start, M = _random_segments(N, 200, 5, hot)
    w0, T0, last0, is0, ie0 = native_ops.alpha2weight(torch.tensor(alpha), torch.tensor(ray_id), N)
    a, rs = cu(alpha), cu(start, torch.int32)
    w, T, last = torch.empty(M, device='cuda'), torch.empty(M, device='cuda'), torch.empty(N, device='cuda')
    i_end = torch.empty(N, device='cuda', dtype=torch.int32)
    ops.alpha2weight_fwd(a, rs, N, w, T, last, i_end)
    assert np.array_equal(w.cpu().numpy(), w0.numpy())
    assert np.array_equal(T.cpu().numpy(), T0.numpy())
    assert np.array_equal(last.cpu().numpy(), last0.numpy())
    nonempty = start[1:] > start[:-1]
    assert np.array_equal(i_end.cpu().numpy()[nonempty], ie0.numpy()[nonempty])
    if hot:
        assert (ie0.numpy()[nonempty] < start[1:][nonempty]).any(), 'test must exercise the early stop'
    rng = np.random.RandomState(1)
    gw, gl = rng.randn(M).astype(np.float32), rng.randn(N).astype(np.float32)
    g0 = native_ops.alpha2weight_backward(torch.tensor(alpha), w0, T0, last0, is0, ie0, N, torch.tensor(gw), torch.tensor(gl))
    g = torch.empty(M, device='cuda')
    ops.alpha2weight_bwd(a, w, T, last, rs, i_end, N, cu(gw), cu(gl), g)
    assert np.array_equal(g.cpu().numpy(), g0.numpy())


def test_flat_adam_matches_reference_trajectory():
    from poseprobe_amd import ops
    d = load('adam.npz')
    p = cu(d['p0'].reshape(-1))
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    seg_end = torch.tensor([p.numel()], dtype=torch.int32, device='cuda')
    lr = 0.1
    for s in range(3):
        lr *= 0.1 ** (1 / 10000)
        g = cu(d['grads'][s].reshape(-1))
        ops.adam_flat(p, g, m, v, seg_end, torch.tensor([lr], device='cuda'), 1.0, 0.9, 0.99, 1e-8, s + 1, 1)
        assert_close(p.cpu().numpy().reshape(4, 5), d['traj'][s], rtol=2e-6, atol=1e-7, name=f'adam step {s}')
        assert float(g.abs().max()) == 0.0
    assert_close(m.cpu().numpy().reshape(4, 5), d['exp_avg'], rtol=2e-6, atol=1e-8)
    assert_close(v.cpu().numpy().reshape(4, 5), d['exp_avg_sq'], rtol=2e-6, atol=1e-10)


@pytest.mark.parametrize('shape', [(5, 6, 7), (16, 12, 9)])
def test_grid_tv_adam_step_matches_oracle(shape):
    """Fused TV-gradient + Adam + zero-fill on the channels-last grid == oracle total_variation autograd + adam_update."""
    from oracle import voxurf_oracle as O
    from poseprobe_amd import ops
    X, Y, Z = shape
    C = 12
    g = torch.Generator().manual_seed(3)
    k0 = (torch.randn(1, C, X, Y, Z, generator=g) * 0.1)
    k0[0, :, 1, 1, 1] = k0[0, :, 1, 1, 2]             # exact ties -> sign(0) = 0
    grad_render = torch.randn(1, C, X, Y, Z, generator=g) * 1e-3
    m0 = torch.randn(1, C, X, Y, Z, generator=g) * 1e-3
    v0 = torch.rand(1, C, X, Y, Z, generator=g) * 1e-6
    ls, w_tv, lr, step = 0.1, 0.01, 0.0977, 7
    # oracle
    p = k0.clone().requires_grad_(True)
    tv = O.total_variation(p)
    (tv * w_tv * ls).backward()
    gtot = p.grad + grad_render
    p_ref, m_ref, v_ref = k0.clone(), m0.clone(), v0.clone()
    O.adam_update(p_ref, gtot, m_ref, v_ref, step, lr)
    # HIP
    cl = lambda t: t[0].permute(1, 2, 3, 0).contiguous().cuda()
    p_in, p_out = cl(k0), torch.empty(X, Y, Z, C, device='cuda')
    gr, m, v = cl(grad_render), cl(m0), cl(v0)
    tv_out = torch.zeros(1, device='cuda')
    for xb, xe in ((0, X // 2), (X // 2, X)):      # two x-slabs, as two ranks would do
        ops.grid_tv_adam_step(p_in, p_out, gr, m, v, (X, Y, Z), C, xb, xe, ls * w_tv / (3 * k0.numel()), 1.0, lr, 0.9,
                              0.99, 1e-8, step, tv_out)
    back = lambda t: t.permute(3, 0, 1, 2)[None].cpu()
    assert_close(back(p_out), p_ref, rtol=1e-5, atol=5e-6, name='p')   # one Adam step moves by <= lr ~ 0.1
    assert_close(back(m), m_ref, rtol=1e-5, atol=1e-9, name='m')
    assert_close(back(v), v_ref, rtol=1e-5, atol=1e-12, name='v')
    assert float(gr.abs().max()) == 0.0
    assert_close(tv_out.cpu()[0] / (3 * k0.numel()), tv.detach(), rtol=1e-5, name='tv value')
    tv2 = torch.zeros(1, device='cuda')
    ops.grid_tv_value(p_in, (X, Y, Z), C, tv2)
    assert_close(tv2.cpu()[0] / (3 * k0.numel()), tv.detach(), rtol=1e-5, name='tv value (standalone)')


@pytest.mark.parametrize('shape', [(5, 6, 7), (16, 12, 9), (40, 32, 32)])
def test_sparse_grid_step_is_bit_identical_to_dense(shape):
    """pp_grid_tv_adam_step_sparse with the scatter's touched-voxel map == the dense pass, bit for bit: parameters,
    both moments and the zero-filled gradient (the TV value up to summation order); the previous step's map is cleared."""
    from poseprobe_amd import ops
    X, Y, Z = shape
    C = 12
    g = torch.Generator().manual_seed(5)
    p = (torch.randn(X, Y, Z, C, generator=g) * 0.1).cuda()
    m0 = (torch.randn(X, Y, Z, C, generator=g) * 1e-3).cuda()
    v0 = (torch.rand(X, Y, Z, C, generator=g) * 1e-5).cuda()
    hit = torch.rand(X, Y, Z, generator=g) < 0.07                      # ~7 % of the voxels receive data gradient
    grad = torch.zeros(X, Y, Z, C)
    grad[hit] = torch.randn(int(hit.sum()), C, generator=g) * 1e-2
    grad[hit.nonzero()[0][0], hit.nonzero()[0][1], hit.nonzero()[0][2]] = 0.0   # a marked voxel whose gradient is exactly 0
    touched = hit.reshape(-1).to(torch.uint8).cuda()
    stale = torch.full((X * Y * Z,), 1, dtype=torch.uint8, device='cuda')   # last step's map: must come back all zero
    outs = []
    for sparse in (False, True):
        gr, m, v = grad.cuda().clone(), m0.clone(), v0.clone()
        po = torch.empty_like(p)
        tv = torch.zeros(1, device='cuda')
        a = (p, po, gr, m, v, (X, Y, Z), C, 0, X, 1e-4, 0.5, 0.1, 0.9, 0.99, 1e-8, 3, tv)
        if sparse:
            ops.grid_tv_adam_step_sparse(*a, touched, stale)
        else:
            ops.grid_tv_adam_step(*a)
        torch.cuda.synchronize()
        outs.append((po.cpu(), m.cpu(), v.cpu(), gr.cpu(), tv.cpu()))
    for a, b, name in zip(outs[0][:4], outs[1][:4], ('p', 'exp_avg', 'exp_avg_sq', 'grad')):
        assert torch.equal(a, b), name
    # the TV value is a sum of per-block partials combined with float atomics: same terms, unordered
    assert abs(float(outs[0][4]) - float(outs[1][4])) <= 1e-5 * abs(float(outs[0][4]))
    assert float(outs[1][3].abs().max()) == 0
    assert int(stale.sum()) == 0


def test_packed_sample_exchange_replays_the_local_scatter():
    """pp_k0_pack_samples + pp_k0_scatter_packed (the multi-GPU exchange format, two shards with ragged counts and stale rows
    past the count) == pp_k0_scatter_samples applied shard by shard; marks included."""
    from poseprobe_amd import ops
    cfg = _cfg(12)
    sc = cfg.pp
    X, Y, Z = cfg.world_size
    cap, C = 300, 12
    g = torch.Generator().manual_seed(9)
    lo, hi = torch.tensor(cfg.xyz_min), torch.tensor(cfg.xyz_max)
    shards = []
    for M in (257, 120):
        pts = (lo + (hi - lo) * (torch.rand(cap, 3, generator=g) * 1.1 - 0.05)).float().cuda()    # some land outside the box
        gf = torch.zeros(cap, 64); gf[:, :57] = torch.randn(cap, 57, generator=g)
        shards.append((pts, gf.cuda(), torch.tensor([M], dtype=torch.int32, device='cuda')))
    ref = torch.zeros(X, Y, Z, C, device='cuda'); ref_t = torch.zeros(X * Y * Z, dtype=torch.uint8, device='cuda')
    packed = torch.full((2, cap, 16), 77.0, device='cuda')                                         # stale content everywhere
    for r, (pts, gf, cnt) in enumerate(shards):
        ops.k0_scatter_samples(sc, pts, cnt, cap, gf, ref, ref_t)
        ops.k0_pack_samples(pts, gf, cnt, cap, C, packed[r])
    got = torch.zeros_like(ref); got_t = torch.zeros_like(ref_t)
    ops.k0_scatter_packed(sc, packed, 2, cap, got, got_t)
    torch.cuda.synchronize()
    assert int(packed[0, 0, 15:16].view(torch.int32)[0]) == 257 and int(packed[1, 0, 15:16].view(torch.int32)[0]) == 120
    assert torch.equal(got_t, ref_t) and int(ref_t.sum()) > 0
    assert_close(got.cpu().numpy(), ref.cpu().numpy(), rtol=1e-5, atol=1e-6, name='replayed scatter')   # unordered atomics


def test_sorted_scatter_is_reproducible_and_equals_the_atomic_one():
    """pp_k0_scatter_samples_sorted / _packed_sorted: (i) the same sums as the atomic kernels (to fp32 summation order) and the
    same touched map, (ii) bit-identical from run to run, (iii) the packed form over two shards bit-identical to the plain form
    over the concatenated samples (both add in (shard, sample, corner) order), (iv) equal to a sequential fp32 accumulation in
    sample order on the host for a voxel that many samples hit."""
    from poseprobe_amd import ops
    cfg = _cfg(12)
    sc = cfg.pp
    X, Y, Z = cfg.world_size
    cap, C = 4096, 12
    g = torch.Generator().manual_seed(21)
    lo, hi = torch.tensor(cfg.xyz_min), torch.tensor(cfg.xyz_max)
    counts = (4096, 1500)
    shards = []
    for M in counts:
        pts = (lo + (hi - lo) * (torch.rand(cap, 3, generator=g) * 1.1 - 0.05)).float()
        pts[:64] = (lo + (hi - lo) * 0.503).float()                     # 64 samples in one cell: a long run for one voxel
        gf = torch.zeros(cap, 64); gf[:, :57] = torch.randn(cap, 57, generator=g) * torch.exp(torch.randn(cap, 1, generator=g) * 3)
        shards.append((pts.cuda(), gf.cuda(), torch.tensor([M], dtype=torch.int32, device='cuda')))
    zeros = lambda: (torch.zeros(X, Y, Z, C, device='cuda'), torch.zeros(X * Y * Z, dtype=torch.uint8, device='cuda'))
    work = torch.empty(ops.k0_scatter_sorted_workspace(2 * cap), dtype=torch.uint8, device='cuda')
    # (i) + (ii): first shard, atomic vs sorted, sorted twice
    pts, gf, cnt = shards[0]
    ref, ref_t = zeros(); ops.k0_scatter_samples(sc, pts, cnt, cap, gf, ref, ref_t)
    a, a_t = zeros(); ops.k0_scatter_samples_sorted(sc, pts, cnt, cap, gf, a, work, a_t)
    b, b_t = zeros(); ops.k0_scatter_samples_sorted(sc, pts, cnt, cap, gf, b, work, b_t)
    torch.cuda.synchronize()
    assert torch.equal(a, b) and torch.equal(a_t, b_t)
    assert torch.equal(a_t, ref_t) and int(ref_t.sum()) > 0
    assert_close(a.cpu().numpy(), ref.cpu().numpy(), rtol=1e-5, atol=1e-6, scaled=1e-6, name='sorted vs atomic scatter')
    # (iii): packed form over both shards == plain form over the concatenation (valid samples only, in shard order)
    packed = torch.full((2, cap, 16), 77.0, device='cuda')
    for r, (p_, g_, c_) in enumerate(shards):
        ops.k0_pack_samples(p_, g_, c_, cap, C, packed[r])
    got, got_t = zeros(); ops.k0_scatter_packed_sorted(sc, packed, 2, cap, got, work, got_t)
    cat_pts = torch.cat([shards[0][0][:counts[0]], shards[1][0][:counts[1]], torch.zeros(2 * cap - sum(counts), 3, device='cuda')])
    cat_gf = torch.cat([shards[0][1][:counts[0]], shards[1][1][:counts[1]], torch.zeros(2 * cap - sum(counts), 64, device='cuda')])
    cat, cat_t = zeros()
    ops.k0_scatter_samples_sorted(sc, cat_pts, torch.tensor([sum(counts)], dtype=torch.int32, device='cuda'), 2 * cap, cat_gf, cat, work, cat_t)
    torch.cuda.synchronize()
    assert torch.equal(got, cat) and torch.equal(got_t, cat_t)
    # (iv): the voxel the 64 coincident samples hit, channel 0: sequential fp32 sum in sample order (all samples of shard 0 that
    # reach it, with the kernel's own weights = its result divided out is not available, so compare against the atomic result
    # within summation-order tolerance and against the exact float64 sum)
    u = ((shards[0][0][0].cpu() - lo) / (hi - lo) * (torch.tensor([X, Y, Z]) - 1)).double()
    i0 = u.floor().long()
    vox = a[i0[0], i0[1], i0[2], 0]
    assert float(vox.abs()) > 0 and abs(float(vox) - float(ref[i0[0], i0[1], i0[2], 0])) <= 1e-5 * float(vox.abs()) + 1e-6
