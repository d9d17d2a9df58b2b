"""CPU study for DESIGN.md 10.2: accuracy of split-precision products against fp64, next to a plain fp32 product, for
the MLP shapes of the path (K = 128).  fp16 hi/lo pairs, 3 products (hi*hi, hi*lo, lo*hi), fp32 accumulation; optional
power-of-two tile scaling so that small operands (backward: gradients of 1e-6) stay in fp16's normal range.
Prints max / rms error relative to the rms of the exact result.   python tools/split_precision_study.py"""
import numpy as np

rng = np.random.default_rng(0)


def split16(x, scale):
    xs = (x * scale).astype(np.float32)
    hi = xs.astype(np.float16).astype(np.float32)
    lo = (xs - hi).astype(np.float16).astype(np.float32)
    return hi, lo


def split_bf16(x, n):
    parts, r = [], x.astype(np.float32)
    for _ in range(n):
        b = (r.view(np.uint32) & 0xFFFF0000).view(np.float32)       # truncation to bf16
        parts.append(b)
        r = (r - b).astype(np.float32)
    return parts


def report(name, got, ref):
    e = got.astype(np.float64) - ref
    s = np.sqrt((ref ** 2).mean())
    print(f'  {name:34s} max {np.abs(e).max() / s:.2e}   rms {np.sqrt((e ** 2).mean()) / s:.2e}')


for label, a_scale in (('forward (activations O(1))', 1.0), ('backward (gradients O(1e-6))', 1e-6)):
    A = (np.maximum(rng.standard_normal((4096, 128)), 0) * a_scale).astype(np.float32)       # rows x K
    W = (rng.standard_normal((128, 128)) * np.sqrt(2.0 / 128)).astype(np.float32)
    ref = A.astype(np.float64) @ W.astype(np.float64).T
    print(label)
    report('fp32 product', A @ W.T, ref)
    for sa_name, sa in (('no scaling', 1.0), ('tile scaled to 2^10', 2.0 ** np.round(10 - np.log2(np.abs(A).max())))):
        ah, al = split16(A, sa)
        wh, wl = split16(W, 2.0 ** 8)
        got = (ah @ wh.T + ah @ wl.T + al @ wh.T) / np.float32(sa * 2.0 ** 8)
        report(f'fp16 hi/lo, 3 products, {sa_name}', got, ref)
    a3, w3 = split_bf16(A, 3), split_bf16(W, 3)
    got3 = a3[0] @ w3[0].T + a3[0] @ w3[1].T + a3[1] @ w3[0].T
    report('bf16 x3 (3 products)', got3, ref)
    got6 = got3 + a3[0] @ w3[2].T + a3[2] @ w3[0].T + a3[1] @ w3[1].T
    report('bf16 x3 (6 products)', got6, ref)
