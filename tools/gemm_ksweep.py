#!/usr/bin/env python3
"""Fixed-overhead probe: time of vapx_gemm vs K at fixed M,N (intercept = per-launch overhead)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gemm_sweep import bench, EPI
M = int(sys.argv[1]) if len(sys.argv) > 1 else 25600
for N in (256, 768):
    for epi in (0, 3, 1):
        if epi == 3 and N != 256: continue
        for tile in (64, 128):
            row = []
            for K in (32, 64, 128, 256, 512, 1024):
                ms, tf = bench(M, N, K, epi, tile, iters=30)
                row.append(f"K{K}:{ms*1e3:6.1f}us")
            print(f"M={M} N={N} {EPI[epi]:9s} t{tile:3d} " + " ".join(row), flush=True)
