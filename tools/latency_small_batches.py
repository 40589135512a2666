#!/usr/bin/env python3
"""Host-inclusive vapx_step latency for small batches (1 / 8 / 64 streams) on the GPU."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vap_realtime_amd import engine, synth, weights as W
cpc, vap = W.synthetic_weights(0, 20)
for S in (1, 8, 64):
    eng = engine.Engine(W.pack_blob(cpc, vap), 20, 2.5, max_streams=S)
    a = synth.noise_batch(S, 800 * 8)
    for i in range(60): eng.step(a[:, :, (i % 8) * 800:(i % 8 + 1) * 800])
    ts = []
    for i in range(200):
        t = time.perf_counter(); eng.step(a[:, :, (i % 8) * 800:(i % 8 + 1) * 800]); ts.append((time.perf_counter() - t) * 1e3)
    ts = np.array(ts); print(f"S={S}: host-inclusive step p50 {np.percentile(ts,50):.3f} ms p99 {np.percentile(ts,99):.3f} ms")
    eng.close()
