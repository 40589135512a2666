#!/bin/bash
# Drive one resource flat out (tools/microbench/power_probe) and sample watts / shader clock beside it.  Usage: tools/power_probe.sh [seconds]
sec=${1:-4}
for mode in idle f16 f32 lds l2 valu "mix 0" "mix 2" "mix 4" "mix 8"; do
  ( while true; do rocm-smi --showclocks --showpower --csv 2>/dev/null | tail -n +2 | head -1; sleep 0.2; done ) > /tmp/pp.txt &
  W=$!
  if [ "$mode" = idle ]; then sleep 2; res="(idle)"; else res=$(tools/microbench/power_probe $mode $sec | tail -1); [ -z "${mode##mix*}" ] && res=$(tools/microbench/power_probe mix $sec ${mode#mix } | tail -1); fi
  kill $W; wait $W 2>/dev/null
  python3 - "$mode" "$res" <<'PY'
import sys,re,statistics
rows=[l for l in open('/tmp/pp.txt') if 'card' in l]
rows=rows[len(rows)//3:]          # steady part
sclk=[int(re.findall(r'\((\d+)Mhz\)', l)[2]) for l in rows if len(re.findall(r'\((\d+)Mhz\)', l))>=3]
pw=[float(l.strip().split(',')[-1]) for l in rows]
print(f"{sys.argv[1]:8s} sclk {statistics.median(sclk):6.0f} MHz  power {statistics.median(pw):6.0f} W   {sys.argv[2]}")
PY
done
