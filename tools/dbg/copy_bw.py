import torch, time
M = 130944
a = torch.randn(M, 256, device='cuda'); b = torch.empty_like(a)
def t(fn, n=20):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
us = t(lambda: b.copy_(a)); print(f'copy 134 MB -> 134 MB: {us:.1f} us = {2 * a.numel() * 4 / us / 1e6:.2f} TB/s (read + write)')
us = t(lambda: torch.relu_(a)); print(f'in-place relu (read + write same lines): {us:.1f} us = {2 * a.numel() * 4 / us / 1e6:.2f} TB/s')
us = t(lambda: a.sum()); print(f'sum (read only): {us:.1f} us = {a.numel() * 4 / us / 1e6:.2f} TB/s')
us = t(lambda: b.zero_()); print(f'zero (write only): {us:.1f} us = {a.numel() * 4 / us / 1e6:.2f} TB/s')
c = torch.randn(M, 320, device='cuda')
us = t(lambda: b.copy_(c[:, :256])); print(f'strided rows (ld 320) copy: {us:.1f} us = {2 * b.numel() * 4 / us / 1e6:.2f} TB/s')
