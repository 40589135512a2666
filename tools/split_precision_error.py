#!/usr/bin/env python3
"""How accurate is an fp32 GEMM emulated with split low-precision MFMA operands?  (DESIGN.md §4 "Precision", §7.)

x = hi + lo with hi = f16(x), lo = f16((x - hi) * 2^11); C = hi.hi + (hi.lo + lo.hi) / 2^11 accumulated in fp32 (the MFMA
accumulates exact f16 products in fp32).  Compared against float64 on a 256x256x256 product whose A operand has x30 outlier
columns (LayerNorm outputs with a few large features), next to a plain fp32 GEMM and the bf16 splits.  CPU only."""
import numpy as np

rng = np.random.default_rng(0)
M = K = N = 256
A = rng.standard_normal((M, K)).astype(np.float32)
A[:, :8] *= 30
W = (rng.standard_normal((N, K)) * 1.5 / 16).astype(np.float32)
ref = A.astype(np.float64) @ W.astype(np.float64).T


def mm(a, b):
    return a.astype(np.float32) @ b.astype(np.float32).T


def split_f16(x, scale):
    hi = x.astype(np.float16)
    lo = ((x - hi.astype(np.float32)) * scale).astype(np.float16)
    return hi, lo


def bf16(x):
    u = x.astype(np.float32).view(np.uint32)
    r = ((u >> 16) & 1) + 0x7FFF
    return ((u + r) & 0xFFFF0000).view(np.float32)


print(f"|C| max = {np.abs(ref).max():.1f}")
print(f"fp32 GEMM                max err {np.abs(mm(A, W) - ref).max():.3e}")
for scale in (2048.0, 1.0):
    ah, al = split_f16(A, scale)
    wh, wl = split_f16(W, scale)
    print(f"f16 x1 (hi.hi)           max err {np.abs(mm(ah, wh) - ref).max():.3e}")
    c3 = mm(ah, wh) + (mm(ah, wl) + mm(al, wh)) / scale
    print(f"f16 x3, lo scale {scale:6.0f}  max err {np.abs(c3 - ref).max():.3e}")
ah = bf16(A); al = bf16(A - ah); al2 = bf16(A - ah - al)
wh = bf16(W); wl = bf16(W - wh); wl2 = bf16(W - wh - wl)
c3 = mm(ah, wh) + mm(ah, wl) + mm(al, wh)
c6 = c3 + mm(al, wl) + mm(ah, wl2) + mm(al2, wh)
print(f"bf16 x3                  max err {np.abs(c3 - ref).max():.3e}")
print(f"bf16 x6                  max err {np.abs(c6 - ref).max():.3e}")
