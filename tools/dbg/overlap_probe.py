"""Can the HBM-bound grid pass run BESIDE the matrix-bound warp-MLP kernels?  (GPU box)

A: k_grid_tv_adam on a stream restricted to P compute units (hipExtStreamCreateWithCUMask), alone.
B: the same, while the main stream runs warp fwd + bwd + weight gradients on 256 - P persistent work-groups (option mlp_wgs);
   wall time of the pair against the serial sum at full width.

    python tools/dbg/overlap_probe.py [P ...]
"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from poseprobe_amd import ops, _lib

hip = ctypes.CDLL('libamdhip64.so')
dev = torch.device('cuda:0')
torch.cuda.init(); torch.zeros(1, device=dev)


def masked_stream(n_cus):
    """Bit i of the mask = compute unit i in the driver's enumeration (it deals the bits round-robin over the XCDs)."""
    words = (ctypes.c_uint32 * 8)()
    for i in range(n_cus):
        words[i // 32] |= 1 << (i % 32)
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
    assert rc == 0, f'hipExtStreamCreateWithCUMask -> {rc}'
    return torch.cuda.ExternalStream(st.value, device=dev)


G, C = 160, 12
p = torch.randn(G, G, G, C, device=dev) * 0.1
po = torch.empty_like(p)
m, v, g = torch.zeros_like(p), torch.zeros_like(p), torch.zeros_like(p)
nvox = G ** 3
hit = (torch.rand(nvox // 4, device=dev) < 0.084).repeat_interleave(4)
touched, other = hit.to(torch.uint8), torch.zeros(nvox, dtype=torch.uint8, device=dev)
tv = torch.zeros(1, device=dev)
grid_args = (p, po, g, m, v, (G, G, G), C, 0, G, 1e-4, 1.0, 0.1, 0.9, 0.99, 1e-8, 3, tv, touched, other)
grid = lambda: ops.grid_tv_adam_step_sparse(*grid_args)

gen = torch.Generator(device='cpu').manual_seed(1)
rnd = lambda *s, scale=1.0: (torch.randn(*s, generator=gen) * scale).to(dev)
M, cap = 55000, 1024 * 186
count = torch.tensor([M], dtype=torch.int32, device=dev)
warp_p = torch.zeros(50564 + 60, device=dev); warp_p[:50564] = rnd(50564, scale=0.09)
pts = rnd(cap, 3, scale=0.5)
acts = torch.zeros(4 * cap * 4 * 128, device=dev); out = torch.zeros(cap, 16, device=dev)
g_out = rnd(cap, 16); scratch = torch.zeros(3 * cap * 4 * 128 + 49152, device=dev)
wgrad = torch.zeros_like(warp_p); pgrad = torch.zeros(cap, 3, device=dev)


def mlp():
    ops.warp_bwd(warp_p, pts, acts, g_out, count, cap, 1.5, scratch, wgrad, pgrad)
    ops.warp_fwd(warp_p, pts, count, cap, 1.5, acts, out)


def wall(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


# the null stream synchronises implicitly with every blocking stream (the masked one has no non-blocking variant): the main work
# runs on a stream of its own
torch.cuda.synchronize()
torch.cuda.set_stream(torch.cuda.Stream())
ops.warp_fwd(warp_p, pts, count, cap, 1.5, acts, out)
main = torch.cuda.current_stream()
t_grid, t_mlp = wall(grid), wall(mlp)
print(f'full width: grid {t_grid:6.1f} us   warp bwd + wgrad + fwd {t_mlp:6.1f} us   serial {t_grid + t_mlp:6.1f} us', flush=True)
for P in [int(x) for x in sys.argv[1:]] or [32, 64, 80, 96, 112, 128]:
    side = masked_stream(P)
    def grid_side():
        side.wait_stream(main)
        with torch.cuda.stream(side): grid()
        main.wait_stream(side)
    tg = wall(grid_side)
    _lib.set_option('mlp_wgs', 256 - P)
    tm = wall(mlp)
    def pair():
        # the MLP kernels first: their work-groups take whole CUs, the grid kernel's fit only where the mask allows
        ev = torch.cuda.Event(); ev.record(main)
        ops.warp_bwd(warp_p, pts, acts, g_out, count, cap, 1.5, scratch, wgrad, pgrad)
        side.wait_event(ev)
        with torch.cuda.stream(side): grid()
        ops.warp_fwd(warp_p, pts, count, cap, 1.5, acts, out)
        main.wait_stream(side)
    tp = wall(pair)
    _lib.set_option('mlp_wgs', 0)
    print(f'P = {P:3d}: grid alone on P CUs {tg:6.1f} us ({(288 + 96 * 0.084) * nvox / tg / 1e6:5.2f} TB/s)   MLP on {256 - P} WGs {tm:6.1f} us   '
          f'both {tp:6.1f} us   vs serial full width {t_grid + t_mlp:6.1f}', flush=True)
