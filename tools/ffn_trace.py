#!/usr/bin/env python3
"""Timeline of ffn_block_kernel workgroups from the s_memtime phase stamps (engine env VAPX_FFN_TRACE=<file>, layer-0 FFN
block of the last step).  Usage: VAPX_FFN_TRACE=/tmp/t.bin python bench.py --streams S --configs= ... ; tools/ffn_trace.py /tmp/t.bin
s_memtime ticks at a constant 100 MHz on gfx950 (10 ns)."""
import sys

import numpy as np

NAMES = ["entry", "tile staged+LN", "ffn1.0 mm", "gelu.0", "h->LDS.0", "ffn2.0 mm", "ffn1.1 mm", "gelu.1", "h->LDS.1", "ffn2.1 mm",
         "ffn1.2 mm", "gelu.2", "h->LDS.2", "ffn2.2 mm", "resid add", "x_out store", "x->LDS", "kvx0 mm", "kvx0 store", "kvx1 mm",
         "kvx1 store", "LN_self->LDS", "q mm", "q store", "k mm", "k store", "v mm", "v store"]
SPLIT = "--split" in sys.argv
if SPLIT:       # ffn_block_f16x3_kernel<0>: coarser stamps (VAPX_FLAG_SPLIT_F16 engine, bench.py --split-f16)
    sys.argv.remove("--split")
    NAMES = ["entry", "tile staged+LN+split", "ffn1.0 mm", "gelu.0 -> sH under ffn1.1 mm", "ffn2.0 mm", "gelu.1 -> sH under ffn1.2 mm", "ffn2.1 mm",
             "gelu.2 -> sH", "ffn2.2 mm", "resid + x_out store", "row stats", "kvx0 mm+store", "kvx1 mm", "kvx1 scale + stores issued",
             "LN->LDS + q mm+store", "k mm+store", "v mm+store"]
NST = len(NAMES)
t = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 32).astype(np.int64)
if SPLIT and (t[:, 30] > 0).any() and int(np.median(t[t[:, 30] > 0, 30])) != NST:
    # the kernel records how many stamps it wrote: mode 1 without Q|K|V chunks (round 5: the next layer's attention kernel projects them)
    NST = int(np.median(t[t[:, 30] > 0, 30]))
    head = ["entry", "attention rows staged (scaled, split)", "proj mm", "residual add", "barrier (rows read)", "park + barrier", "row stats (LDS reads, reductions)", "LN + split -> sX + barrier"]
    woven = ["ffn1.0 mm", "gelu.0 -> sH under ffn1.1 mm", "ffn2.0 mm", "gelu.1 -> sH under ffn1.2 mm", "ffn2.1 mm", "gelu.2 -> sH", "ffn2.2 mm"]
    apart = ["ffn1.0 mm", "gelu.0 -> sH", "ffn2.0 mm", "ffn1.1 mm", "gelu.1 -> sH", "ffn2.1 mm", "ffn1.2 mm", "gelu.2 -> sH", "ffn2.2 mm"]      # the un-woven variant (tools/microbench/patches/ffn_block_f16x3_knobs.patch)
    tail = ["resid (+ x_out store)", "row stats + x_out / xn_out rows", "kvx0 mm+store", "kvx1 mm", "kvx1 scale + stores issued"]
    m1 = head + (apart if NST >= len(head) + len(apart) + len(tail) else woven) + tail
    NAMES = (m1 + [f"stamp {k}" for k in range(len(m1), NST)])[:NST]
tick_ns = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
n = t.shape[0]
valid = (t[:, :NST] > 0).all(axis=1)
t = t[valid]
# absolute times from s_memrealtime (constant 100 MHz, chip-wide); the phase stamps are s_memtime (per-XCD counter whose rate
# is calibrated per workgroup against the realtime pair)
rt0 = t[:, 28].min()
start, end = (t[:, 28] - rt0) * 0.01, (t[:, 29] - rt0) * 0.01          # us
dur = end - start
ticks = (t[:, NST - 1] - t[:, 0]).astype(np.float64)
us_per_tick = dur / np.maximum(ticks, 1)
print(f"s_memtime rate: median {1.0 / np.median(us_per_tick):.1f} ticks/us")
tick_ns = float(np.median(us_per_tick)) * 1e3
print(f"{n} workgroups ({valid.sum()} complete); kernel span {end.max():.1f} us; per-WG duration median {np.median(dur):.1f} us "
      f"(min {dur.min():.1f}, max {dur.max():.1f})")
order = np.argsort(start)
q = [0, 0.25, 0.5, 0.75, 1.0]
print("start time quantiles (us):", [round(float(np.quantile(start, x)), 1) for x in q])
print("end   time quantiles (us):", [round(float(np.quantile(end, x)), 1) for x in q])
d = np.diff(t[:, :NST], axis=1) * tick_ns / 1e3
print("phase durations, median over WGs (us) [p10 .. p90]:")
mm_tot = other_tot = 0.0
for k in range(NST - 1):
    med = float(np.median(d[:, k]))
    if "mm" in NAMES[k + 1]:
        mm_tot += med
    else:
        other_tot += med
    print(f"  {NAMES[k + 1]:16s} {med:7.2f}  [{np.quantile(d[:, k], 0.1):6.2f} .. {np.quantile(d[:, k], 0.9):6.2f}]")
print(f"sum of MFMA phases {mm_tot:.1f} us, everything else {other_tot:.1f} us")
# first-round vs later-round workgroups
first = start < np.median(dur) * 0.5
if (~first).any():
    print(f"first-wave WGs: {first.sum()}, median duration {np.median(dur[first]):.1f} us; later WGs: {(~first).sum()}, median {np.median(dur[~first]):.1f} us")
# concurrency over time: how many WGs are resident at 10 evenly spaced instants
ts = np.linspace(0, end.max(), 12)[1:-1]
print("resident WGs over the kernel:", [int(((start <= x) & (end > x)).sum()) for x in ts])
