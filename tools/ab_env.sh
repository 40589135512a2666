#!/bin/bash
# Same-box A/B of environment settings: tools/ab_env.sh "<bench args>" "ENV=.. ENV=.." "ENV=.." ...  (two rounds)
args="$1"; shift
for rep in 1 2; do
for e in "$@"; do
  env $e timeout 300 python bench.py $args --configs= --no-latency --no-cpu-baseline --front-end-streams 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$e', d['value'], d['ms_per_step'], d['parity_gate']['worst_abs'], d['roofline']['frac'], d['roofline']['avg_launch_us'])"
done; done
