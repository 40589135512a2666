#!/usr/bin/env python3
"""Per-kernel resource table (VGPR/AGPR/occupancy/spill/LDS) for a .hip file, via
hipcc -Rpass-analysis=kernel-resource-usage.  Usage: tools/kres.py file.hip [extra hipcc flags]"""
import re
import subprocess
import sys

src = sys.argv[1]
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", "/dev/null",
       "-Rpass-analysis=kernel-resource-usage"] + sys.argv[2:]
p = subprocess.run(cmd, capture_output=True, text=True)
rows, cur = [], None
keys = {"VGPRs": "V", "AGPRs": "A", "Occupancy [waves/SIMD]": "occ", "VGPRs Spill": "spill",
        "LDS Size [bytes/block]": "lds", "ScratchSize [bytes/lane]": "scr", "SGPRs": "S"}
for l in p.stderr.splitlines():
    m = re.search(r"Function Name: (\S+)", l)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    hit = False
    for k, short in keys.items():
        m = re.search(r"    " + re.escape(k) + r": (\d+)", l)
        if m and cur is not None:
            cur[short] = m.group(1)
            hit = True
    if not hit and ("error" in l or "warning:" in l):
        print(l)
for r in rows:
    n = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    n = re.sub(r"\(anonymous namespace\)::", "", n)[:80]
    print(f"{n:80s} V={str(r.get('V')):>3} A={str(r.get('A')):>3} S={str(r.get('S')):>3} occ={r.get('occ')} spill={r.get('spill')} scr={r.get('scr')} lds={r.get('lds')}")
sys.exit(p.returncode)
