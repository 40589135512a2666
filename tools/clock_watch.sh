#!/bin/bash
# Sample the GPU's shader clock and socket power while a command runs: tools/clock_watch.sh <label> <command...>
label="$1"; shift
( while true; do rocm-smi --showclocks --showpower --csv 2>/dev/null | tail -n +2 | head -2 | tr '\n' ' '; echo; sleep 0.25; done ) > /tmp/clk_$label.txt &
W=$!
"$@" > /tmp/out_$label.txt 2>/dev/null
kill $W
python3 - "$label" <<'PY'
import sys,re
lab=sys.argv[1]
rows=[l for l in open(f'/tmp/clk_{lab}.txt') if l.strip()]
print(lab, 'samples', len(rows))
for l in rows[:2]: print('  ', l.strip()[:300])
import statistics
sclk=[];pw=[]
for l in rows:
    m=re.findall(r'\((\d+)Mhz\)', l)
    if len(m) >= 3: sclk.append(int(m[2]))            # fclk, mclk, sclk, socclk
    p=re.findall(r',(\d+\.\d+)', l)
    if p: pw.append(float(p[-1]))
if sclk: print('   sclk MHz: median', statistics.median(sclk), 'min', min(sclk), 'max', max(sclk))
if pw: print('   power W: median', statistics.median(pw), 'max', max(pw))
PY
tail -1 /tmp/out_$label.txt | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('   ', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
