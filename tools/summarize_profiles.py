#!/usr/bin/env python3
"""Turn the on-box summaries of tools/profile_configs.sh (gpurun_out/prof_<tag>_<workload>/) into the committed artefacts:
  profiles/<tag>_<workload>_kernel_stats.csv   rocprofv3 --kernel-trace --stats (our kernels)
  profiles/<tag>_<workload>_pmc.txt            per-kernel means per dispatch of the four --pmc passes
  profiles/<tag>_<workload>_bench.json         the bench line of the same command
and refresh profiles/pmc_traffic.json (HBM bytes per launch = WRITE_SIZE + 2 x FETCH_SIZE KiB, the gfx950 FETCH correction of
MI355X_MICROARCH.md) for every kernel of every workload.
Usage: tools/summarize_profiles.py <tag> [workload ...]      (workload "c3_split" = gpurun_out/prof_<tag>_c3_split, the --split-f16 run)"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
wls = sys.argv[2:] or ["c2", "s4096_20hz", "c3", "c5"]
KEY = {"c2": "256x20hz_T50", "s4096_20hz": "4096x20hz_T50", "c3": "4096x50hz_T250", "c5": "4096x20hz_T50_bc+nod"}
CLASS = [("ffn_block_kernel", "ffn_block"), ("ffn_block_f16x3_kernel", "ffn_block"), ("attn_block_kernel", "attention"),
         ("attention_long2_kernel", "attention"), ("attention_long_f16x3_kernel", "attention"), ("attention_proj_f16x3_kernel", "attention_proj"), ("conv_tail_kernel", "conv_tail"), ("lstm_kernel", "lstm"), ("last_block_kernel", "last_row"),
         ("head_kernel", "head"), ("conv0_kernel", "conv0"), ("gather_ln_kernel", "gather_ln"), ("gemm_f32_kernel<2, 2, 4", "gemm_cn_relu"),
         ("gemm_f32_kernel<4, 1, 4", "gemm_cn_relu")]
ours = lambda name: ("kernel" in name and "at::" not in name and "rocclr" not in name)


def parse_summary(path):
    out = {}
    if not os.path.exists(path):
        return out
    for line in open(path):
        m = re.match(r"(.{60}) n=\s*(\d+) (.*)", line.rstrip("\n"))
        if not m:
            continue
        name = m.group(1).strip()
        vals = dict(kv.split("=") for kv in m.group(3).split())
        out[name] = (int(m.group(2)), {k: float(v) for k, v in vals.items()})
    return out


traffic_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
traffic = json.load(open(traffic_path)) if os.path.exists(traffic_path) else {}
traffic["_how_r02"] = ("r02 entries: tools/profile_configs.sh (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over "
                       "`bench.py --workload <w> --configs= --steps 4`), means per dispatch by tools/pmc_summary.py; KiB units; "
                       "bytes_per_launch_corrected = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 per MI355X_MICROARCH.md (gfx950 tallies 128-B "
                       "read requests at 64 B).")
traffic["_how_r04"] = ("r04 entries: same recipe on the round-4 kernels (tools/profile_configs.sh <workload> r04 4; c3 passes restricted to steady-state dispatches).")
traffic["_how_r03"] = ("r03 entries (c2, s4096_20hz, c3, c5 re-measured on the round-3 kernels; C3 at its full 4096 streams, not scaled from 512): same recipe as "
                       "_how_r02; the class entry of a workload (\"ffn_block\", \"attention\", ...) is the first kernel of that class in name order, i.e. "
                       "ffn_block_kernel<1, 1> for C3.")
for wl in wls:
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}_{wl}")
    if not os.path.isdir(src):
        print("missing", src)
        continue
    ks = os.path.join(src, "kernel_stats.csv")
    if os.path.exists(ks):
        lines = open(ks).read().splitlines()
        keep = [lines[0]] + [l for l in lines[1:] if ours(l.split('","')[0])]
        open(os.path.join(ROOT, "profiles", f"{tag}_{wl}_kernel_stats.csv"), "w").write("\n".join(keep) + "\n")
    with open(os.path.join(ROOT, "profiles", f"{tag}_{wl}_pmc.txt"), "w") as f:
        for part, title in (("pmc_sq", "pass 1: SQ cycles (quad-cycle units except SQ_VALU_MFMA_BUSY_CYCLES / GRBM)"), ("pmc_mops", "pass 2: instruction mix / LDS"),
                            ("pmc_fetch", "pass 3: FETCH_SIZE (KiB, raw: double for wide coalesced reads on gfx950)"), ("pmc_write", "pass 4: WRITE_SIZE (KiB)")):
            f.write(f"# {title}\n")
            p = os.path.join(src, part + ".txt")
            if os.path.exists(p):
                for line in open(p):
                    if ours(line[:60]):
                        f.write(line)
    bj = os.path.join(src, "bench.json")
    if os.path.exists(bj) and os.path.getsize(bj) > 10:
        open(os.path.join(ROOT, "profiles", f"{tag}_{wl}_bench.json"), "w").write(open(bj).read().strip().splitlines()[-1] + "\n")
    fe, wr = parse_summary(os.path.join(src, "pmc_fetch.txt")), parse_summary(os.path.join(src, "pmc_write.txt"))
    entry = {}
    for name in fe:
        if not ours(name) or name not in wr:
            continue
        f_kib, w_kib = fe[name][1].get("FETCH_SIZE", 0.0), wr[name][1].get("WRITE_SIZE", 0.0)
        rec = {"kernel": name, "dispatches": fe[name][0], "fetch_kib_raw": f_kib, "write_kib": w_kib,
               "bytes_per_launch_raw": (f_kib + w_kib) * 1024, "bytes_per_launch_corrected": (2 * f_kib + w_kib) * 1024}
        entry[name] = rec
        for sub, cls in CLASS:
            if sub in name and cls not in entry:
                entry[cls] = rec
    sj = os.path.join(src, "source.json")
    if entry and os.path.exists(sj):       # the tree the passes ran on (hash taken ON the box) + the commit it belongs to, if HEAD still has those sources
        sys.path.insert(0, ROOT)
        from vap_realtime_amd import provenance
        sha = json.load(open(sj))["csrc_sha"]
        head = provenance.git_head(ROOT)
        entry["_source"] = {"csrc_sha": sha, "git": (head if provenance.kernel_source_hash_at("HEAD", ROOT) == sha else f"{head}+uncommitted"), "tag": tag}
    if entry:
        traffic[KEY[wl.replace("_split", "")] + ("_split_f16" if wl.endswith("_split") else "")] = entry
json.dump(traffic, open(traffic_path, "w"), indent=1)
print("profiles/ updated for", wls)
