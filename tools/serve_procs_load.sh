#!/bin/bash
# `serve --worker-procs on` under load with REAL engines: N worker processes (all on this box's one GPU: --share-gpu), a front-door process, and
# the native load generator playing S real-time 20 Hz dialogues against the reference's port pair.  Usage: tools/serve_procs_load.sh <out.json> [S] [N] [seconds]
OUT=${1:-gpurun_out/serve_procs_load.json}; S=${2:-4096}; N=${3:-2}; SEC=${4:-20}
PIN=$((20000 + RANDOM % 20000)); POUT=$((PIN + 1))
python -u -m vap_realtime_amd.serve --synthetic-weights 0 --streams $((S / N)) --gpus $N --share-gpu --worker-procs on --precision fp32 \
    --port_num_in $PIN --port_num_out $POUT --stats_sec 0 > ${OUT%.json}.serve.log 2>&1 &
SERVE=$!
for i in $(seq 1 600); do grep -q "input :" ${OUT%.json}.serve.log 2>/dev/null && break; sleep 0.5; done
grep "input :" ${OUT%.json}.serve.log || { echo "serve did not come up"; cat ${OUT%.json}.serve.log | tail -5; kill $SERVE; exit 1; }
tools/loadgen --port-in $PIN --port-out $POUT --streams $S --hz 20 --seconds $SEC --warm 6 --threads 8 --inband 1 > $OUT 2> ${OUT%.json}.loadgen.err
kill -TERM $SERVE; wait $SERVE; echo "serve exit code $?"
python - $OUT <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d[k] for k in ("streams", "frames_sent", "frames_answered", "lat_p50_ms", "lat_p99_ms", "lat_max_ms", "route_changes", "inband_unreadable", "schedule_slips")})
PY
