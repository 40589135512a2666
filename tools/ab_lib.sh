#!/bin/bash
# Same-box A/B of library builds: tools/ab_lib.sh "<bench args>" lib1.so lib2.so ...  (two rounds; prints frames/s, ms/tick, parity, dominant-class frac and mean launch us)
args="$1"; shift
for rep in 1 2; do
for lib in "$@"; do
  VAPX_LIBRARY=$lib timeout 300 python bench.py $args --configs= --no-latency --no-cpu-baseline --front-end-streams 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['value'], d['ms_per_step'], d['parity_gate']['worst_abs'], d['roofline']['frac'], d['roofline']['avg_launch_us'])"
done; done
