#!/usr/bin/env python3
"""Stamp every workload of profiles/pmc_traffic.json with the tree its PMC passes were taken at: ``_source = {"git": commit, "csrc_sha": hash of
vap-realtime_amd/csrc at that commit}``.  For entries that have no stamp yet the commit is the LAST one that changed the entry's per-kernel
numbers (found by replaying the file's git history) — the passes were committed with the tree they ran on.  New passes are stamped when they are taken:
tools/profile_configs.sh writes the content hash of the sources ON the GPU box (source.json), tools/summarize_profiles.py stores it with the commit.
bench.py compares csrc_sha with the sources it runs (roofline.traffic_source.stale)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vap_realtime_amd import provenance  # noqa: E402

PATH = os.path.join(ROOT, "profiles", "pmc_traffic.json")


def kernels_of(entry):
    return {k: v for k, v in entry.items() if isinstance(v, dict) and "(" in k}


def main():
    d = json.load(open(PATH))
    commits = subprocess.run(["git", "-C", ROOT, "log", "--format=%h", "--", "profiles/pmc_traffic.json"], check=True, capture_output=True, text=True).stdout.split()
    history = []
    for c in commits:                                   # newest first
        try:
            history.append((c, json.loads(subprocess.run(["git", "-C", ROOT, "show", f"{c}:profiles/pmc_traffic.json"], check=True, capture_output=True).stdout)))
        except Exception:                               # noqa: BLE001
            continue
    for key, entry in d.items():
        if key.startswith("_") or not isinstance(entry, dict) or "_source" in entry:
            continue
        want = kernels_of(entry)
        origin = None
        for c, old in history:                          # walk back while the entry's numbers are the same: the oldest such commit took them
            if isinstance(old.get(key), dict) and kernels_of(old[key]) == want:
                origin = c
            else:
                break
        if origin is None:
            continue
        entry["_source"] = {"git": origin, "csrc_sha": provenance.kernel_source_hash_at(origin, ROOT)}
        print(f"{key:48s} <- {origin} {entry['_source']['csrc_sha']}")
    json.dump(d, open(PATH, "w"), indent=1)


if __name__ == "__main__":
    main()
