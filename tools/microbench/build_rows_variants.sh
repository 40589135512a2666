#!/bin/bash
# builds tools/microbench/ffn_rows_bench_<name> for each "name:defines" argument (run from the repo root)
for spec in "$@"; do
  n=${spec%%:*}; d=${spec#*:}; [ "$d" == "$spec" ] && d=""
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Ivap-realtime_amd/csrc -Itools/microbench -Wno-unused-result -Wno-unused-value $d -o tools/microbench/ffn_rows_bench_$n tools/microbench/ffn_rows_bench.hip 2>&1 | grep -i -A5 "error" &
done
wait
