#!/usr/bin/env python3
"""Where does a kernel wait for memory with a CONSERVATIVE count?  hipcc's s_waitcnt insertion knows one in-order counter (vmcnt: loads AND stores):
behind a branch that may or may not have issued memory operations it must assume it has not, and the next wait for an OLDER load becomes
vmcnt(0) - it then also waits for every prefetch issued since (round 6: attention_long2_kernel waited for the NEXT key tile at the top of every
score tile; ffn_block_kernel<1,1> looked 16 ring slots up one dependent load at a time).  Prints, per kernel of a HIP source, the instruction
stream as tokens - Mn = n MFMAs, Ln / Sn = n global loads / stores, rn / wn = LDS reads / writes, B = barrier, Wk = s_waitcnt vmcnt(k), J = branch,
labels on their own lines - and the number of vmcnt(0) waits.
Usage: tools/isa_waits.py vap-realtime_amd/csrc/<file>.hip [kernel-name-substring] [--summary]"""
import os
import re
import subprocess
import sys
import tempfile
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def assembly(src):
    out = os.path.join(tempfile.mkdtemp(), "k.s")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only", "-o", out, src],
                   check=True, capture_output=True)
    return open(out).read()


def tokens(lines):
    def cls(l):
        l = l.strip()
        if l.startswith("v_mfma"): return "M"
        if l.startswith(("global_load", "buffer_load")): return "L"
        if l.startswith(("global_store", "buffer_store")): return "S"
        if l.startswith("ds_read"): return "r"
        if l.startswith("ds_write"): return "w"
        if l.startswith("s_barrier"): return "B "
        if "s_waitcnt" in l and "vmcnt" in l: return "W" + re.search(r"vmcnt\((\d+)\)", l).group(1) + " "
        if l.startswith(".LBB"): return "\n" + l.split(":")[0] + ": "
        if l.startswith(("s_cbranch", "s_branch")): return "J(" + l.split()[-1] + ") "
        return ""
    out = "".join(cls(l) for l in lines)
    for ch in "MLSrw":
        out = re.sub("(%s+)" % ch, lambda m: "%s%d " % (ch, len(m.group(1))), out)
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    summary = "--summary" in sys.argv
    txt = assembly(args[0])
    want = args[1] if len(args) > 1 else ""
    for m in re.finditer(r"^(_Z\S+):\s*; @.*?\n(.*?)^\.Lfunc_end", txt, re.S | re.M):
        name, lines = m.group(1), m.group(2).split("\n")
        if want not in name:
            continue
        waits = Counter(re.search(r"vmcnt\((\d+)\)", l).group(1) for l in lines if "s_waitcnt" in l and "vmcnt" in l)
        print(f"== {name}: {sum('v_mfma' in l for l in lines)} MFMAs, {sum(waits.values())} vmcnt waits, vmcnt(0): {waits.get('0', 0)}")
        if not summary:
            print(tokens(lines))


if __name__ == "__main__":
    main()
