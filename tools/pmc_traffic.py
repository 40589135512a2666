#!/usr/bin/env python3
"""HBM traffic per launch of a kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).
Usage: tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <kernel substr> [min_grid]
FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE reports half of the bytes of wide
(16 B/lane) coalesced reads (MI355X_MICROARCH.md §HBM): the corrected figure doubles it."""
import csv
import json
import sys


def mean(path, counter, name, min_grid):
    vals = []
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and name in r["Kernel_Name"] and int(r["Grid_Size"]) >= min_grid:
            vals.append(float(r["Counter_Value"]))
    return sum(vals) / len(vals), len(vals)


f, w, name = sys.argv[1], sys.argv[2], sys.argv[3]
min_grid = int(sys.argv[4]) if len(sys.argv) > 4 else 0
fk, n1 = mean(f, "FETCH_SIZE", name, min_grid)
wk, n2 = mean(w, "WRITE_SIZE", name, min_grid)
print(json.dumps({"kernel": name, "launches": [n1, n2], "fetch_kib_raw": fk, "write_kib": wk,
                  "bytes_per_launch_raw": (fk + wk) * 1024, "bytes_per_launch_corrected": (2 * fk + wk) * 1024}))
