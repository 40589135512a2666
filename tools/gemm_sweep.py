#!/usr/bin/env python3
"""On-GPU micro-benchmark of vapx_gemm over the shapes the VAP step uses: TFLOP/s per
(M, N, K, epilogue, tile_rows).  Usage: python tools/gemm_sweep.py [S ...]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vap_realtime_amd import engine

lib = engine.load_library()
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
EPI = {0: "store", 1: "gelu", 2: "resid", 3: "resid_ln", 4: "cn_relu", 5: "bias_ln_gelu"}


def bench(M, N, K, epi, tile, iters=20):
    A = torch.randn(M, K, device="cuda")
    W = torch.randn(N, K, device="cuda") / K ** 0.5
    Cm = torch.empty(M, N, device="cuda")
    C2 = torch.empty(M, N, device="cuda") if epi == 3 else None
    bias = torch.randn(N, device="cuda")
    g = torch.randn(N, device="cuda")
    R = torch.randn(M, N, device="cuda") if epi in (2, 3) else None
    args = (None, M, N, K, p(A), p(W), p(Cm), epi, p(bias), p(g), p(g), p(R), p(C2), tile)
    for _ in range(3):
        assert lib.vapx_gemm(*args) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        lib.vapx_gemm(*args)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return ms, 2.0 * M * N * K / ms / 1e9


if __name__ == "__main__":
    Ss = [int(a) for a in sys.argv[1:]] or [256, 4096]
    for S in Ss:
        rows = S * 2 * 50
        shapes = [(rows, 768, 256, 0), (rows, 768, 256, 1), (rows, 256, 256, 3), (rows, 256, 768, 3), (rows, 256, 256, 0),
                  (rows, 512, 256, 0), (S * 2 * 56, 256, 2048, 4), (S * 2 * 28, 256, 1024, 4), (S * 2 * 14, 256, 1024, 4),
                  (S * 2 * 5, 256, 1024, 4), (S * 2, 256, 1280, 5)]
        for (M, N, K, epi) in shapes:
            row = []
            for tile in (32, 64, 128):
                ms, tf = bench(M, N, K, epi, tile)
                row.append(f"t{tile}: {ms*1e3:8.1f}us {tf:6.1f}TF")
            print(f"S={S:5d} M={M:7d} N={N:4d} K={K:5d} {EPI[epi]:13s} " + " | ".join(row), flush=True)
