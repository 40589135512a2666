#!/usr/bin/env python3
"""Per-tick HBM traffic of every workload in profiles/pmc_traffic.json: per-launch bytes (tools/pmc_traffic.py) x launches per tick, summed per
bench.py kernel class and over the whole tick.  Writes the result back as "<workload>"["_tick"]; bench.py reads it for `roofline.traffic`
(mean bytes per launch of the dominant CLASS: the same launches its HIP-event time covers) and `roofline.traffic_ratio` (tick bytes / algorithmic
bytes).  Launches per tick: dispatches / ticks where a whole run was profiled; the launch chain of DESIGN.md section 4 for the C3 passes, which
profile steady-state dispatches only."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, "profiles", "pmc_traffic.json")


def kernel_class(name: str) -> str:
    if name.startswith("ffn_block_f16x3_kernel<"):      # <MODE, TAILQ> (round 5; <MODE> before)
        return "ffn_proj" if name[len("ffn_block_f16x3_kernel<")] == "2" else "ffn_block"
    if name.startswith("ffn_block"):                    # ffn_block_kernel<.., MODE>
        return "ffn_proj" if (name.endswith("2>(FfnArgs)")) else "ffn_block"
    if name.startswith(("attn_block", "attention_long", "attention_proj")):
        return "attention"
    for pre, cls in (("conv0", "conv0"), ("conv_tail", "conv_tail"), ("last_block", "last_row"), ("lstm", "lstm"), ("head", "head"),
                     ("ring_append", "gather_ln"), ("gather_ln", "gather_ln"), ("gemm_f32", "gemm"), ("add_kernel", "other"), ("ln_rows", "other"),
                     ("pbc_rows", "other")):
        if name.startswith(pre):
            return cls
    return "other"


C3_CHAIN = {"ffn_block": 3, "ffn_proj": 2, "attention": 5}    # launches per tick of the long-window chain; everything else once
C3_SPLIT_R05 = {"attention_long_f16x3_kernel(AttnArgs)": 3, "attention_proj_f16x3_kernel(AttnProjArgs)": 2}   # round 5: Q|K|V projected inside the self-attention of layers 1-2


def main():
    d = json.load(open(PATH))
    for key, entry in d.items():
        if key.startswith("_") or not isinstance(entry, dict):
            continue
        kernels = {k: v for k, v in entry.items() if isinstance(v, dict) and "(" in k and "bytes_per_launch_corrected" in v}
        if not kernels:
            continue
        disp = [v.get("dispatches") for v in kernels.values()]
        c3 = "50hz_T250" in key
        ticks = None if c3 or None in disp else min(disp)          # every kernel runs at least once per tick
        launches, by_class, raw, cor = {}, {}, 0.0, 0.0
        for k, v in kernels.items():
            cls = kernel_class(k)
            n = C3_CHAIN.get(cls, 1) if ticks is None else v["dispatches"] / ticks
            if ticks is None and any("attention_proj" in kk for kk in kernels) and k in C3_SPLIT_R05:
                n = C3_SPLIT_R05[k]
            launches[k] = n
            c = by_class.setdefault(cls, {"launches": 0.0, "bytes_corrected": 0.0, "bytes_raw": 0.0})
            c["launches"] += n
            c["bytes_corrected"] += n * v["bytes_per_launch_corrected"]
            c["bytes_raw"] += n * v["bytes_per_launch_raw"]
            raw += n * v["bytes_per_launch_raw"]
            cor += n * v["bytes_per_launch_corrected"]
        entry["_tick"] = {"ticks_profiled": ticks, "launches_per_tick": launches, "by_class": by_class, "bytes_corrected": cor, "bytes_raw": raw,
                          "note": "corrected = WRITE_SIZE + 2 x FETCH_SIZE (MI355X_MICROARCH.md: gfx950 reports half of wide coalesced reads; an upper bound)"}
        print(f"{key:45s} tick {cor / 1e9:8.2f} GB corrected / {raw / 1e9:8.2f} raw; " + ", ".join(f"{c} {v['bytes_corrected'] / 1e9:.2f}" for c, v in sorted(by_class.items(), key=lambda kv: -kv[1]['bytes_corrected'])[:4]))
    json.dump(d, open(PATH, "w"), indent=1)


if __name__ == "__main__":
    sys.exit(main())
