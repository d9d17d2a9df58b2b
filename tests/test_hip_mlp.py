"""The two MLP chains (pp_warp_fwd/bwd, pp_rgbnet_fwd/bwd) against a plain PyTorch fp32 restatement (autograd for the
backward), at sizes that exercise every tile-boundary case of the layer-fused kernels: empty input, one sample, one short
of / exactly / one past a 16-sample (warp) and 64-row (rgbnet) tile, a capacity that is not a multiple of the tile, and
enough tiles for several persistent iterations per work-group.

Tolerances (fp32, different summation order than torch): values rtol 1e-4 / atol 1e-5; gradients 1e-3 of the tensor's max.
"""
import numpy as np
import pytest
import torch

from tests.helpers import assert_close

pytestmark = pytest.mark.gpu
OUT_RANGE = 0.7619


def _warp_params(seed):
    g = torch.Generator().manual_seed(seed)
    shapes = [(128, 3), (128, 128), (128, 128), (128, 128), (4, 128)]
    layers = []
    for i, (o, k) in enumerate(shapes):
        W = torch.randn(o, k, generator=g) * (1.0 / np.sqrt(k)) * (0.3 if i == 4 else 1.4)
        b = torch.randn(o, generator=g) * 0.1
        layers.append((W, b))
    return layers


def _warp_ref(layers, pts):
    """4-row form in torch: row 0 = primal, rows 1..3 = d/dp_i; ReLU mask of row 0 applied to all rows (tests/analytic_model.py
    derives it from the reference's autograd.grad passes)."""
    (W0, b0) = layers[0]
    y0 = pts @ W0.T + b0
    m = (y0 > 0).float()
    X = torch.stack([y0 * m, m * W0[:, 0], m * W0[:, 1], m * W0[:, 2]], dim=1)          # [M,4,128]
    for W, b in layers[1:4]:
        Y = X @ W.T
        Y = torch.cat([Y[:, :1] + b, Y[:, 1:]], dim=1)
        X = Y * (Y[:, :1] > 0).float()
    W4, b4 = layers[4]
    out = X @ W4.T
    out = torch.cat([out[:, :1] + b4, out[:, 1:]], dim=1)
    return out * OUT_RANGE                                                                # [M,4,4]


def _close_but_flipped_rows(a, b, rtol, atol, name, max_rows, loose=0.2):
    err = np.abs(a - b)
    assert err.max() <= loose * np.abs(b).max(), f'{name}: max abs err {err.max():.3e}'
    bad = (err > atol + rtol * np.abs(b)).reshape(a.shape[0], -1).any(axis=1)
    assert bad.sum() <= max_rows, f'{name}: {bad.sum()} rows outside rtol {rtol} / atol {atol} (max abs err {err.max():.3e})'


def _pack(layers):
    return torch.cat([t.reshape(-1) for W, b in layers for t in (W, b)])


@pytest.mark.parametrize('M,cap', [(0, 40), (1, 1), (15, 15), (16, 16), (17, 40), (63, 100), (65, 65), (1000, 1003),
                                   (9001, 9100)])
def test_warp_chain_matches_torch(M, cap):
    from poseprobe_amd import ops
    dev = 'cuda'
    layers = _warp_params(3)
    g = torch.Generator().manual_seed(M + 1)
    pts_h = torch.randn(cap, 3, generator=g) * 0.5
    og_h = torch.randn(cap, 16, generator=g)
    P = _pack(layers)
    params = torch.zeros(P.numel() + 60, device=dev); params[:P.numel()] = P.to(dev)
    pts, og = pts_h.to(dev), og_h.to(dev)
    count = torch.tensor([M], dtype=torch.int32, device=dev)
    acts = torch.full((4 * cap * 4 * 128,), float('nan'), device=dev)
    out = torch.full((cap, 16), 7.0, device=dev)
    ops.warp_fwd(params, pts, count, cap, OUT_RANGE, acts, out)
    scratch = torch.zeros(3 * cap * 4 * 128 + 49152, device=dev)
    pgrad = torch.zeros_like(params)
    ptsg = torch.full((cap, 3), 0.25, device=dev)                      # pts_grad accumulates (+=)
    ops.warp_bwd(params, pts, acts, og, count, cap, OUT_RANGE, scratch, pgrad, ptsg)
    torch.cuda.synchronize()
    assert float((out[M:] - 7.0).abs().max()) == 0 if M < cap else True          # rows past the count are not written
    assert float((ptsg[M:] - 0.25).abs().max()) == 0 if M < cap else True
    if M == 0:
        assert float(pgrad.abs().max()) == 0
        return
    lay = [(W.clone().requires_grad_(True), b.clone().requires_grad_(True)) for W, b in layers]
    p = pts_h[:M].clone().requires_grad_(True)
    ref = _warp_ref(lay, p)
    (ref.reshape(M, 16) * og_h[:M]).sum().backward()
    c = lambda t: t.detach().cpu().numpy()
    # ~3.5e6 pre-activations at the largest size: between two fp32 implementations about one of them lies within rounding
    # distance of zero and flips its ReLU state, which moves that sample's 16 outputs by ~1e-2; everything else is tight
    _close_but_flipped_rows(c(out[:M]), c(ref.reshape(M, 16)), rtol=1e-4, atol=1e-5, name='warp out', max_rows=3)
    _close_but_flipped_rows(c(ptsg[:M]) - 0.25, c(p.grad), rtol=1e-3, atol=2e-5 + 1e-3 * float(p.grad.abs().max()),
                            name='warp pts_grad', max_rows=3)
    gref = torch.cat([t.grad.reshape(-1) for W, b in lay for t in (W, b)])
    # a flipped sample changes the few gradient entries it feeds by its own O(1) contribution
    ga, gb = c(pgrad[:gref.numel()]), c(gref)
    bad = np.abs(ga - gb) > 2e-5 + 1e-3 * np.abs(gb) + 1e-3 * np.abs(gb).max()
    assert bad.sum() <= 8 and np.abs(ga - gb).max() <= 2e-2 * np.abs(gb).max(), f'warp param grads: {bad.sum()} entries, max {np.abs(ga - gb).max():.3e}'


def _rgb_params(seed):
    g = torch.Generator().manual_seed(seed)
    shapes = [(128, 57), (128, 128), (128, 128), (3, 128)]
    return [(torch.randn(o, k, generator=g) * (1.4 / np.sqrt(k)), torch.randn(o, generator=g) * 0.1) for o, k in shapes]


@pytest.mark.parametrize('M,cap', [(0, 70), (1, 1), (63, 63), (64, 64), (65, 130), (1000, 1003), (20000, 20011)])
def test_rgbnet_chain_matches_torch(M, cap):
    from poseprobe_amd import ops
    from poseprobe_amd.engine import pack_rgbnet
    dev = 'cuda'
    layers = _rgb_params(5)
    g = torch.Generator().manual_seed(M + 2)
    feat_h = torch.zeros(cap, 64); feat_h[:, :57] = torch.randn(cap, 57, generator=g)
    gr_h = torch.randn(cap, 3, generator=g)
    P = pack_rgbnet(layers)
    params = torch.zeros(P.numel() + 60, device=dev); params[:P.numel()] = P.to(dev)
    feat, gr = feat_h.to(dev), gr_h.to(dev)
    count = torch.tensor([M], dtype=torch.int32, device=dev)
    acts = torch.full((3 * cap * 128,), float('nan'), device=dev)
    rgb = torch.full((cap, 3), 7.0, device=dev)
    ops.rgbnet_fwd(params, feat, count, cap, acts, rgb)
    scratch = torch.zeros(3 * cap * 128 + 49152, device=dev)
    pgrad = torch.zeros_like(params)
    fgrad = torch.full((cap, 64), 5.0, device=dev)                     # feat_grad is assigned (=) for rows < M
    ops.rgbnet_bwd(params, feat, acts, rgb, gr, count, cap, scratch, pgrad, fgrad)
    torch.cuda.synchronize()
    if M < cap:
        assert float((rgb[M:] - 7.0).abs().max()) == 0 and float((fgrad[M:] - 5.0).abs().max()) == 0
    if M == 0:
        assert float(pgrad.abs().max()) == 0
        return
    lay = [(W.clone().requires_grad_(True), b.clone().requires_grad_(True)) for W, b in layers]
    x = feat_h[:M, :57].clone().requires_grad_(True)
    h = x
    for W, b in lay[:3]:
        h = torch.relu(h @ W.T + b)
    ref = torch.sigmoid(h @ lay[3][0].T + lay[3][1])
    (ref * gr_h[:M]).sum().backward()
    c = lambda t: t.detach().cpu().numpy()
    assert_close(c(rgb[:M]), c(ref), rtol=1e-4, atol=1e-5, name='rgb')
    assert_close(c(fgrad[:M, :57]), c(x.grad), rtol=1e-3, atol=2e-6, name='rgb feat_grad', scaled=1e-3)
    from poseprobe_amd.engine import unpack_rgbnet
    for i, ((gW, gb), (W, b)) in enumerate(zip(unpack_rgbnet(pgrad), lay)):
        assert_close(c(gW), c(W.grad), rtol=1e-3, atol=2e-5, name=f'rgb W{i} grad', scaled=1e-3)
        assert_close(c(gb), c(b.grad), rtol=1e-3, atol=2e-5, name=f'rgb b{i} grad', scaled=1e-3)
    # the zero padding of the 57 -> 64 input columns receives no gradient from real data (features there are zero)
    assert float(pgrad[:128 * 64].view(128, 64)[:, 57:].abs().max()) == 0


def test_split_precision_kernels_are_as_accurate_as_the_fp32_instructions():
    """The default MLP kernels evaluate every 128 x 128 product as three fp16 MFMA products per fp32 product (option mlp_split).
    Against a float64 evaluation of the same network on the same inputs their error must not exceed the fp32-instruction
    kernels' (mlp_split = 0) by more than a quarter - for the warp outputs (values and Jacobians), the rgb outputs and the
    weight gradients, with O(1) weights and with weights / gradients scaled down by 1e-3 / 1e-6 (small magnitudes are where a
    fixed-range 16-bit format would lose)."""
    from poseprobe_amd import ops, _lib
    from poseprobe_amd.engine import pack_rgbnet
    dev = 'cuda'
    M = cap = 20000
    count = torch.tensor([M], dtype=torch.int32, device=dev)
    default = _lib.get_option('mlp_split')
    ctxs = {0: ops.Context(mlp_split=0), default: ops.Context(mlp_split=default)}     # two contexts, two arithmetics, one process
    try:
        for wscale, gscale in ((1.0, 1.0), (1e-3, 1e-6)):
            g = torch.Generator().manual_seed(17)
            layers = [(W * (wscale if 0 < i < 4 else 1.0), b * wscale) for i, (W, b) in enumerate(_warp_params(3))]
            pts_h = torch.randn(cap, 3, generator=g) * 0.5
            ref = _warp_ref([(W.double(), b.double()) for W, b in layers], pts_h.double()).reshape(M, 16)
            P = _pack(layers)
            params = torch.zeros(P.numel() + 60, device=dev); params[:P.numel()] = P.to(dev)
            og = (torch.randn(cap, 16, generator=g) * gscale * torch.exp(torch.randn(cap, 1, generator=g) * 2)).to(dev)
            rl = [(W * (wscale if i < 3 else 1.0), b * wscale) for i, (W, b) in enumerate(_rgb_params(5))]
            feat_h = torch.zeros(cap, 64); feat_h[:, :57] = torch.randn(cap, 57, generator=g)
            h = feat_h[:, :57].double()
            for li, (W, b) in enumerate(rl):
                h = h @ W.double().T + b.double()
                h = torch.relu(h) if li < 3 else torch.sigmoid(h)
            rp = pack_rgbnet(rl)
            rparams = torch.zeros(rp.numel() + 60, device=dev); rparams[:rp.numel()] = rp.to(dev)
            err = {}
            for mode in (0, default):
                cx = ctxs[mode]
                acts = torch.zeros(4 * cap * 4 * 128, device=dev); out = torch.zeros(cap, 16, device=dev)
                ops.warp_fwd(params, pts_h.to(dev), count, cap, OUT_RANGE, acts, out, cx)
                e = (out.cpu().double() - ref).abs()
                keep = ~(e > 1e-3 * ref.abs().max()).any(1)                     # samples with a flipped ReLU are not rounding error
                racts = torch.zeros(3 * cap * 128, device=dev); rgb = torch.zeros(cap, 3, device=dev)
                ops.rgbnet_fwd(rparams, feat_h.to(dev), count, cap, racts, rgb, cx)
                # weight gradients against float64 products of THIS mode's own Ybar and X (isolates the weight-gradient kernel)
                scratch = torch.zeros(3 * cap * 4 * 128 + 49152, device=dev)
                wg = torch.zeros_like(params); pg = torch.zeros(cap, 3, device=dev)
                stage2 = ops.warp_bwd_data(params, pts_h.to(dev), acts, og, count, cap, OUT_RANGE, scratch, wg, pg, cx)
                assert stage2 == (1 if mode & 2 else 0)
                wg.zero_()
                ops.warp_bwd_weights(acts, scratch, count, cap, wg, stage2, cx)
                torch.cuda.synchronize()
                X = acts.view(4, cap * 4, 128)[:, :4 * M].double().cpu()
                Y = scratch[:3 * cap * 4 * 128].view(3, cap * 4, 128)[:, :4 * M].double().cpu()
                o3 = 128 * 3 + 128 + 2 * (128 * 128 + 128)
                w3 = wg[o3:o3 + 128 * 128].view(128, 128).double().cpu()
                r3 = Y[0].T @ X[2]
                err[mode] = (float((e[keep] ** 2).mean().sqrt()), float(((rgb.cpu().double() - h) ** 2).mean().sqrt()),
                             float(((w3 - r3) ** 2).mean().sqrt() / (r3 ** 2).mean().sqrt()))
            for name, a, b in zip(('warp out', 'rgb', 'W3 gradient'), err[0], err[default]):
                assert b <= 1.25 * a + 1e-12, f'{name} (weights x {wscale}, gradients x {gscale}): fp32 {a:.3e}, split {b:.3e}'
    finally:
        assert _lib.get_option('mlp_split') == default             # the host's default context was never touched


def test_weight_gradient_kernels_follow_growing_magnitudes():
    """k_wgrad_chain_s keeps ONE running power-of-two scale per operand and wavefront over its whole row range and rescales its
    accumulators whenever a 16-row group exceeds it.  Synthetic operands whose magnitude grows by 2^40 along the rows (in the
    order the kernel walks them: from the END of the range) force dozens of rescales per wavefront; small rows in between must
    not be lost either.  Against float64 products the split kernel may not be worse than the fp32-instruction chain kernel by
    more than a quarter; the bias column sums (primal rows) ride along."""
    from poseprobe_amd import ops, _lib
    dev = 'cuda'
    M = cap = 6000
    R = 4 * M
    count = torch.tensor([M], dtype=torch.int32, device=dev)
    g = torch.Generator().manual_seed(31)
    ramp = torch.exp2(torch.linspace(20, -20, R)).unsqueeze(1)                  # rows near the end are the small ones ...
    ramp[torch.randperm(R, generator=g)[:R // 7]] *= 2.0 ** -12                   # ... and every seventh row is tiny where it sits
    acts = torch.zeros(4, cap * 4, 128); scratch = torch.zeros(3 * cap * 4 * 128 + 49152)
    Y = torch.randn(3, R, 128, generator=g) * ramp
    X = torch.relu(torch.randn(3, R, 128, generator=g)) * torch.exp2(torch.randn(1, R, 1, generator=g) * 4)
    acts[:3, :R] = X
    scratch[:3 * cap * 4 * 128].view(3, cap * 4, 128)[:, :R] = Y
    acts_d, scratch_d = acts.reshape(-1).to(dev), scratch.to(dev)
    # layers: (Ybar3 = Y[0], X2) -> W3, (Y[1], X1) -> W2, (Y[2], X0) -> W1 ; bias b_l = column sums of Ybar_l over the primal rows
    base = 128 * 3 + 128
    off = {1: base, 2: base + 128 * 128 + 128, 3: base + 2 * (128 * 128 + 128)}
    ref = {3: Y[0].double().T @ X[2].double(), 2: Y[1].double().T @ X[1].double(), 1: Y[2].double().T @ X[0].double()}
    refb = {3: Y[0][::4].double().sum(0), 2: Y[1][::4].double().sum(0), 1: Y[2][::4].double().sum(0)}
    default = _lib.get_option('mlp_split')
    err = {}
    try:
        for mode in (default & ~16, default | 16):
            cx = ops.Context(mlp_split=mode)
            wg = torch.zeros(50564 + 60, device=dev)
            ops.warp_bwd_weights(acts_d, scratch_d, count, cap, wg, 1, cx)      # stage2 = 1: b1..b3 are this stage's (handed over explicitly)
            torch.cuda.synchronize()
            e = []
            for l in (3, 2, 1):
                got = wg[off[l]:off[l] + 128 * 128].view(128, 128).double().cpu()
                e.append(float(((got - ref[l]) ** 2).mean().sqrt() / (ref[l] ** 2).mean().sqrt()))
                gb = wg[off[l] + 128 * 128:off[l] + 128 * 128 + 128].double().cpu()
                assert float((gb - refb[l]).abs().max()) <= 1e-5 * float(refb[l].abs().max()), f'bias gradient of layer {l}, mode {mode}'
            err[mode & 16] = e
    finally:
        pass
    for l, a, b in zip((3, 2, 1), err[0], err[16]):
        assert b <= 1.25 * a + 1e-9 and b < 5e-6, f'W{l} gradient: fp32 chain {a:.3e}, split chain {b:.3e}'


def test_forward_only_mode_gives_the_same_outputs_without_keeping_activations():
    """pp_warp_fwd / pp_rgbnet_fwd with acts = NULL (inference): bit-identical outputs, nothing written; refused by the kernels that
    cannot run without the buffer."""
    from poseprobe_amd import ops, _lib
    g = torch.Generator().manual_seed(7)
    M, cap = 4099, 4608
    count = torch.tensor([M], dtype=torch.int32, device='cuda')
    warp_p = torch.zeros(ops.WARP_PARAMS + 60, device='cuda'); warp_p[:ops.WARP_PARAMS] = (torch.randn(ops.WARP_PARAMS, generator=g) * 0.09).cuda()
    rgb_p = torch.zeros(ops.RGBNET_PARAMS + 60, device='cuda'); rgb_p[:ops.RGBNET_PARAMS] = (torch.randn(ops.RGBNET_PARAMS, generator=g) * 0.09).cuda()
    pts = (torch.randn(cap, 3, generator=g) * 0.5).cuda()
    feat = torch.randn(cap, 64, generator=g).cuda(); feat[:, 57:] = 0
    acts = torch.empty(4, cap * 4, 128, device='cuda'); racts = torch.empty(3, cap, 128, device='cuda')
    out_a, out_b = torch.zeros(cap, 16, device='cuda'), torch.zeros(cap, 16, device='cuda')
    rgb_a, rgb_b = torch.zeros(cap, 3, device='cuda'), torch.zeros(cap, 3, device='cuda')
    ops.warp_fwd(warp_p, pts, count, cap, 1.5, acts, out_a)
    ops.warp_fwd(warp_p, pts, count, cap, 1.5, None, out_b)
    ops.rgbnet_fwd(rgb_p, feat, count, cap, racts, rgb_a)
    ops.rgbnet_fwd(rgb_p, feat, count, cap, None, rgb_b)
    torch.cuda.synchronize()
    assert torch.equal(out_a[:M], out_b[:M]) and torch.equal(rgb_a[:M], rgb_b[:M])
    fp32 = ops.Context(mlp_split=0)                 # the fp32-instruction kernels cannot run without the buffer
    with pytest.raises(_lib.PoseProbeError):
        ops.warp_fwd(warp_p, pts, count, cap, 1.5, None, out_b, fp32)
    with pytest.raises(_lib.PoseProbeError):
        ops.rgbnet_fwd(rgb_p, feat, count, cap, None, rgb_b, fp32)
