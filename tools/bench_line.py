#!/usr/bin/env python3
"""Condense bench.py's JSON line (stdin) to one short line: value, ms/step, per-kernel ms.  Usage: bench.py ... | tools/bench_line.py [label]"""
import json
import sys

r = json.loads(sys.stdin.readlines()[-1])
print(" ".join(sys.argv[1:]), round(r["value"]), "frames/s", round(r["ms_per_step"], 3), "ms/step", r.get("kernel_ms_per_step"))
