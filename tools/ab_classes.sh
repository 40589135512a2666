#!/bin/bash
# Same-box A/B of library builds with the per-class kernel times of a tick: tools/ab_classes.sh "<bench args>" lib1.so lib2.so ...  (two rounds)
# prints frames/s, ms/tick (min / median of the timed ticks), parity, and the ms per tick of the biggest kernel classes
args="$1"; shift
for rep in 1 2; do
for lib in "$@"; do
  VAPX_LIBRARY=$lib timeout 600 python bench.py $args --configs= --no-latency --no-cpu-baseline --front-end-streams 0 --full-record /tmp/ab_full.json >/dev/null 2>&1
  python - "$lib" <<'PY'
import json, sys
d = json.load(open('/tmp/ab_full.json'))
sp = d.get('ms_per_step_spread') or {}
k = d['kernel_ms_per_step']
print(sys.argv[1], round(d['value'], 1), 'ms/tick', round(d['ms_per_step'], 3), 'min', round(sp.get('min', 0), 3), 'med', round(sp.get('median', 0), 3), 'parity', d['parity_gate']['worst_abs'],
      ' '.join(f"{c}={v:.3f}" for c, v in list(k.items())[:5]), 'sclk', (d.get('board') or {}).get('sclk_mhz_median'))
PY
done; done
