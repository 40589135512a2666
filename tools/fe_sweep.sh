#!/bin/bash
# A/B of front-end options under whatever the host is doing right now: interleaved runs of tools/server_load.py (4096 real-time clients, 10 s)
#   tools/fe_sweep.sh <runs> <variant> [<variant> ...]     variants: lowat (default build), nolowat, each, set, clients
mkdir -p gpurun_out/fe_sweep
R=$1; shift
cat /proc/loadavg
for i in $(seq 1 $R); do for v in "$@"; do
  a="--no-pin"; e=""
  case $v in each) a="--pin-mode each";; set) a="--pin-mode set";; clients) a="--pin-mode clients-only";; nolowat) e="VAPX_INGEST_NO_LOWAT=1";; esac
  env $e VAPX_INGEST_DEBUG=1 timeout 120 python tools/server_load.py --streams 4096 --seconds 10 --warm 4 $a > gpurun_out/fe_sweep/fe_${v}_$i.json 2> gpurun_out/fe_sweep/fe_${v}_$i.err
  python - <<PY
import json
d=json.load(open("gpurun_out/fe_sweep/fe_${v}_$i.json")); s=d["server_stats"]; n=d["net_counters_delta"]
print("$v $i client p50 %.2f p99 %.2f max %.1f late %d | server p99 %.2f max %.1f late %d | cpu %.1f cores, throttled %d, loadavg %s" % (d["lat_p50_ms"], d["lat_p99_ms"], d["lat_max_ms"], d["late_over_10ms"], s["lat_p99_ms"], s["lat_max_ms"], s["late_over_10ms"], n.get("cgroup_usage_usec",0)/1e6/15.0, n.get("cgroup_nr_throttled",0), d["host_limits"].get("loadavg")))
PY
done; done
