#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel name, mean of each counter per dispatch.
Usage: tools/pmc_summary.py <counter_collection.csv> [name-substring ...]"""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
filters = sys.argv[2:]
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
with open(path) as f:
    for row in csv.DictReader(f):
        name = row["Kernel_Name"]
        if filters and not any(s in name for s in filters):
            continue
        short = name.replace("(anonymous namespace)::", "").replace("void ", "")[:60]
        acc[short][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[short][row["Counter_Name"]] += 1
for k in sorted(acc):
    parts = [f"{c}={acc[k][c] / cnt[k][c]:.4g}" for c in sorted(acc[k])]
    n = max(cnt[k].values())
    print(f"{k:60s} n={n:4d} " + " ".join(parts))
