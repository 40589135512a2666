#!/bin/bash
# The HOST half of BASELINE config 4 (32768 dialogues = 8 x 4096 behind ONE port pair), no GPU needed: a front-door process + 8 worker processes
# over the native stand-in for the GPU tick, 8 load-generator processes playing real-time 20 Hz dialogues (10 ms packets, reference framing).
# Usage: tools/c4_host_sweep.sh <out dir> [stream counts ...]      one JSON record per stream count (tools/server_load.py)
OUT=${1:-gpurun_out/r06_frontend}; shift
COUNTS=${@:-"8192 16384 32768"}
mkdir -p $OUT
(nproc; cat /proc/loadavg; ulimit -Hn) > $OUT/host.txt 2>&1
for S in $COUNTS; do
  timeout 600 python tools/server_load.py --standin --shards 8 --worker-procs --streams $S --seconds ${SECONDS_MEASURED:-20} --warm 8 \
      --loadgen-procs 8 --client-threads 8 --src-ips 8 --rx-threads 4 --tx-threads 4 ${EXTRA:-} > $OUT/c4_standin_${S}${TAG:-}.json 2> $OUT/c4_standin_${S}${TAG:-}.err
  python - $OUT/c4_standin_${S}${TAG:-}.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    ss = d["server_stats"]
    print(d["streams"], "sent", d["frames_sent"], "answered", d["frames_answered"], "client p50/p99/max", d["lat_p50_ms"], d["lat_p99_ms"], d["lat_max_ms"],
          "late>10ms", [v for k, v in d.items() if k.startswith("late_over_")], "slips", d["schedule_slips"],
          "| server p99", round(ss["lat_p99_ms"], 2), "late", ss["late_over_10ms"], "of", ss["answered"], "mean batch", round(ss["mean_batch"], 1),
          "worker cpu_s", round(ss.get("cpu_s", 0), 1), "cores", d["cpu_cores_used"], "load", d["host_limits"].get("loadavg"))
except Exception as e:
    print("no record:", e)
PY
done
