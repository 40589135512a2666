#!/usr/bin/env python3
"""Timeline of attn_block_kernel workgroups from the s_memtime phase stamps of the DEBUG build (make -C vap-realtime_amd/csrc trace;
engine env VAPX_ATTN_TRACE=<file>: layer-1 self-attention block of the last step — attention + projection + LN_src + cross-q projection).
Usage: VAPX_LIBRARY=vap-realtime_amd/libvapx_trace.so VAPX_ATTN_TRACE=/tmp/a.bin python bench.py --workload c2 --configs= --no-latency
       --no-cpu-baseline ; tools/attn_trace.py /tmp/a.bin
Long windows (attention_long2_kernel, layer-1 self-attention, first 16384 workgroups): same recipe with --workload c3, then
       tools/attn_trace.py --long /tmp/a.bin"""
import sys

import numpy as np

LONG = "--long" in sys.argv
if LONG:
    sys.argv.remove("--long")
SPLIT = "--split" in sys.argv
if SPLIT:
    sys.argv.remove("--split")
NAMES = ["item start", "Q loads issued", "barrier A (previous item done)", "convert + LDS stores", "barrier B", "Q converted",
         "next loads issued, accumulators zeroed", "key tiles (wave 0: 1)", "next rows arrived, maxima", "l reduce + stores"] if (LONG and SPLIT) else ["entry", "metadata (n, ring rotation) loaded", "Q + K tile 0 (+ the V loads ahead of them) arrived", "first score tile + softmax", "V -> LDS (this wave)", "barrier (all 4 waves)",
         "query tile w (wave 0: 1 key tile)", "query tile 7-w (wave 0: 8 key tiles)"] if LONG else ["entry", "V in LDS (1st latency)", "scores+softmax+PV", "resid/ring loads issued", "barrier (all heads)", "O -> LDS + barrier",
         "proj mm", "LN + xmid stores", "cross-q mm", "qx stores"]
t = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 32).astype(np.int64)
nst = int(np.median(t[:, 30]))
valid = (t[:, :nst] > 0).all(axis=1) & (t[:, 29] > 0)
n = t.shape[0]
t = t[valid]
rt0 = t[:, 28].min()
start, end = (t[:, 28] - rt0) * 0.01, (t[:, 29] - rt0) * 0.01          # us (s_memrealtime: constant 100 MHz, chip-wide)
dur = end - start
ticks = (t[:, nst - 1] - t[:, 0]).astype(np.float64)
us_per_tick = dur / np.maximum(ticks, 1)
print(f"s_memtime rate: median {1.0 / np.median(us_per_tick):.1f} ticks/us")
tick_us = float(np.median(us_per_tick))
print(f"{n} workgroups ({valid.sum()} complete, {nst} stamps); kernel span {end.max():.1f} us; per-WG duration median {np.median(dur):.1f} us "
      f"(min {dur.min():.1f}, max {dur.max():.1f})")
q = [0, 0.25, 0.5, 0.75, 1.0]
print("start time quantiles (us):", [round(float(np.quantile(start, x)), 1) for x in q])
print("end   time quantiles (us):", [round(float(np.quantile(end, x)), 1) for x in q])
d = np.diff(t[:, :nst], axis=1) * tick_us
print("phase durations, median over WGs (us) [p10 .. p90]:")
mm = other = 0.0
for k in range(nst - 1):
    med = float(np.median(d[:, k]))
    name = NAMES[k + 1] if k + 1 < len(NAMES) else f"phase {k + 1}"
    if nst == 9 and k + 1 >= 8:
        name = "stores (no cross-q)"
    if "mm" in name:
        mm += med
    else:
        other += med
    print(f"  {name:26s} {med:7.2f}  [{np.quantile(d[:, k], 0.1):6.2f} .. {np.quantile(d[:, k], 0.9):6.2f}]")
print(f"sum of projection-MFMA phases {mm:.1f} us, everything else {other:.1f} us")
ts = np.linspace(0, end.max(), 12)[1:-1]
print("resident WGs over the kernel:", [int(((start <= x) & (end > x)).sum()) for x in ts])
