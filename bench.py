#!/usr/bin/env python3
"""bench.py — VAP frames/s of the MI355X-native streaming forward pass.

One "step" = one VAP frame (one tick) for every stream of this rank: frame assembly -> CPC CNN ->
LSTM -> downsample -> context ring -> 1+3 transformer layers -> heads, through the C ABI
(vapx_step) with audio and outputs resident in HBM.  Workload at N=1 = BASELINE.json configs[1]:
256 concurrent synthetic stereo streams, 20 Hz frames, 2.5 s context (T=50).  With N GPUs every
rank runs its own 256 streams (weak scaling, no data-path collective: streams are independent).

Prints ONE JSON line on rank 0 (contract in the task prompt): metric/value/unit/... plus
  "roofline":     dominant kernel (by summed time), HIP-event timed inside the timed region
  "cpu_baseline": the oracle (CPU restatement of the reference step) timed on this host, 1 thread
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TF = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, exact f32
HBM_PEAK_GBS = 8000.0
GFLOP_PER_STREAM_FRAME = {(20, 50): 0.920, (50, 250): 4.505, (10, 50): 1.054}   # SURVEY.md §8d / BASELINE.md §3


def macs_per_stream_frame(hz: int, T: int) -> dict:
    """Useful multiply-accumulates per stream-frame (both channels) by kernel class."""
    hop = 16000 // hz
    L = hop + 320
    P0 = L // 5; P1 = P0 // 4; P2 = P1 // 2; P3 = P2 // 2; P4 = P3 // 2; ncpc = P4 - 2
    D = 256
    rows = 2 * T
    m = {
        "conv0": 2 * P0 * D * 10,
        "gemm_cn_relu": 2 * (P1 * 8 + P2 * 4 + P3 * 4 + ncpc * 4) * D * D,
        "lstm": 2 * ncpc * D * 4 * D + 2 * ncpc * D * D,                # recurrence (K=256) + fused downsample
        "gemm_bias_ln_gelu": 0,
        # executed work with exact last-layer pruning (only the newest row of layer 3 is consumed):
        "gemm_store": 2 * ncpc * D * 4 * D + 2 * D * 768,               # LSTM input projection + layer-0 QKV of the NEW row (others cached)
        "gemm_resid_ln": 0,
        "ffn_block": rows * D * (3 * 2 * 768 + 2 * 768 + 2 * 512),      # FFN x3 + QKV and cross-KV of layers 1, 2 (layer 3: absorbed)
        # layer 3 on one row per channel: 14 contractions (q, Wk^T q, Wv, proj, their cross twins, FFN) + two
        # 4-head single-query attentions over T rows of 256 (score + weighted sum)
        "last_row": 2 * (14 * D * D + 2 * 4 * T * D * 2),
        "gemm_gelu": 0, "gemm_resid": 0,
        # fused attention block: dense T x T attention (as SURVEY counts it) of layers 0-2 + output
        # projections (x5) + cross-attention query projections (x2)
        "attention": 5 * 2 * 4 * (T * T * 64 * 2) + rows * 7 * D * D,
        "head": 3 * D * D + 2 * D,
        "gather_ln": 0,
    }
    return m


def _cpu_worker(job):
    """One single-threaded oracle process of the multi-core CPU leg: (seed, hz, ctx_sec, seconds) -> (frames, elapsed)."""
    import torch
    from oracle.vap_oracle import ServerFramer, VapOracle
    from vap_realtime_amd import synth, weights as W
    widx, hz, ctx_sec, seconds = job
    torch.set_num_threads(1)
    cpc, vap = W.synthetic_weights(0, hz, "vap")
    hop = 16000 // hz
    NF = 32
    a = synth.dialogue_batch([widx], hop * NF).reshape(1, 2, NF, hop).transpose(2, 0, 1, 3)
    o = VapOracle(cpc, vap, hz, ctx_sec)
    st, fr = o.new_state(1), ServerFramer(1, hop)
    for i in range(int(ctx_sec * hz)):
        o.step(fr.frame(a[i % NF]), st)
    n, t1 = 0, time.perf_counter()
    while time.perf_counter() - t1 < seconds:
        o.step(fr.frame(a[n % NF]), st)
        n += 1
    return n, time.perf_counter() - t1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--streams", type=int, default=256, help="concurrent streams per GPU")
    ap.add_argument("--frame-hz", type=int, default=20)
    ap.add_argument("--ctx-sec", type=float, default=2.5)
    ap.add_argument("--cpu-baseline-sec", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-procs", type=int, default=-1,
                    help="processes of the multi-core CPU leg (-1 = min(16, logical cores / 2), 0 = skip)")
    ap.add_argument("--no-latency", action="store_true")
    ap.add_argument("--groups", type=int, default=0, help="intra-tick overlap groups (0 = engine default)")
    ap.add_argument("--mode", default="vap", choices=["vap", "bc", "nod", "bc+nod", "vap+bc+nod"],
                    help="model variant (config 5: bc / nod); a+b = weight sets served on one shared CPC trunk "
                         "(one stream-frame = one audio frame through the shared encoder and every listed model)")
    ap.add_argument("--split-f16", action="store_true",
                    help="opt-in: FFN-block contractions as fp32-accurate 3-term f16 split products (VAPX_FLAG_SPLIT_F16)")
    ap.add_argument("--defer-join", action="store_true", help="with --groups > 1: let overlap groups free-run across ticks")
    ap.add_argument("--subtick-streams", type=int, default=1024,
                    help="sub-tick size for the <=10 ms latency leg (0 = skip)")
    args = ap.parse_args()

    import torch
    from vap_realtime_amd import dist_util, engine, synth, weights as W
    from vap_realtime_amd.sharding import shard_streams
    rank, local_rank, world = dist_util.env_rank()
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dist = dist_util.init("nccl", torch.device("cuda", local_rank))

    hz, S = args.frame_hz, args.streams
    T = int(args.ctx_sec * hz)
    hop = 16000 // hz
    my_streams = shard_streams(S * world, world, rank)           # global stream ids of this rank
    modes = args.mode.split("+")
    cpc, vap = W.synthetic_weights(0, hz, modes[0])
    eng = engine.Engine(W.pack_blob(cpc, vap, modes[0]), hz, args.ctx_sec, max_streams=S, device_id=local_rank,
                        groups=args.groups, mode=modes[0], split_f16=args.split_f16)
    followers = []
    for k, m in enumerate(modes[1:]):                            # same cpc_model "file", own VAP state dict
        f = engine.Engine(W.pack_blob(cpc, W.synthetic_weights(1 + k, hz, m)[1], m), hz, args.ctx_sec, max_streams=S,
                          device_id=local_rank, groups=args.groups, mode=m)
        f.attach_trunk(eng)
        followers.append(f)

    NF = 32                                                      # distinct audio frames, cycled
    base = synth.dialogue_batch(my_streams[:min(S, 64)], hop * NF)   # [<=64,2,hop*NF]
    reps = (S + base.shape[0] - 1) // base.shape[0]
    audio = np.concatenate([np.roll(base, 97 * r, axis=2) * (1.0 - 0.01 * r) for r in range(reps)], 0)[:S]
    audio = np.ascontiguousarray(audio.reshape(S, 2, NF, hop).transpose(2, 0, 1, 3))   # [NF,S,2,hop]
    d_audio = torch.from_numpy(audio).cuda()
    d_out = torch.zeros(S, engine.OUT_STRIDE, device="cuda")
    d_out_f = [torch.zeros(S, engine.OUT_STRIDE, device="cuda") for _ in followers]
    stream = torch.cuda.current_stream().cuda_stream

    def step(i):
        eng.step_device(S, d_audio[i % NF].data_ptr(), hop, d_out.data_ptr(), stream=stream, defer_join=args.defer_join)
        for f, o in zip(followers, d_out_f):
            f.step_follow_device(S, o.data_ptr(), stream=stream)

    def profile_enable(classes):
        for e in [eng] + followers:
            e.profile_enable(classes)

    def profile_read():
        tot = {}
        for e in [eng] + followers:
            for k, (ms, cnt) in e.profile_read().items():
                a = tot.get(k, (0.0, 0))
                tot[k] = (a[0] + ms, a[1] + cnt)
        return tot

    def barrier():
        dist_util.barrier(dist, torch.cuda.synchronize)

    # prime the context window (so the timed region is the steady state), then a profiled pass to
    # find the dominant kernel class
    for i in range(T):
        step(i)
    torch.cuda.synchronize()
    profile_enable(range(13))
    profile_read()
    NP = 5
    for i in range(NP):
        step(i)
    prof_all = profile_read()
    breakdown = {k: v[0] / NP for k, v in prof_all.items()}
    dominant = max(breakdown, key=breakdown.get)
    dom_id = [k for k, v in engine.PROF_CLASSES.items() if v == dominant][0]
    profile_enable([dom_id])
    profile_read()

    for i in range(args.warmup):
        step(i)
    barrier()
    profile_read()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    barrier()
    dt = time.perf_counter() - t0
    dom_ms, dom_launches = profile_read()[dominant]
    profile_enable([])
    dt = dist_util.max_over_ranks(dist, dt, "cuda")
    assert torch.isfinite(d_out[:, :6]).all(), "non-finite outputs"

    frames = S * world * args.steps
    value = frames / dt
    macs = macs_per_stream_frame(hz, T)
    if len(modes) > 1:   # every model runs its own downsample + transformer; the encoder classes run once
        shared = ("conv0", "gemm_cn_relu", "conv_tail", "lstm")
        macs = {k: v * (1 if k in shared else len(modes)) for k, v in macs.items()}
    launches_per_step = dom_launches / args.steps
    flop_per_launch = 2.0 * macs[dominant] * S / launches_per_step
    avg_launch_s = dom_ms * 1e-3 / dom_launches
    achieved_tf = flop_per_launch / avg_launch_s / 1e12
    gflop_sf = GFLOP_PER_STREAM_FRAME.get((hz, T), 2.0 * sum(macs.values()) / 1e9)

    traffic = None
    try:   # HBM bytes/launch of the dominant kernel from the committed PMC passes (cannot be collected live)
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            pt = json.load(f)
        if S == 256 and hz == 20 and T == 50:
            traffic = pt["256x20hz_T50"].get(dominant, {}).get("bytes_per_launch_corrected")
    except Exception:
        traffic = None

    result = {
        "metric": "VAP frames/sec (concurrent 16 kHz stereo streams, one frame per stream per step)",
        "value": value,
        "unit": "frames/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic (seeded two-speaker dialogue audio, seeded random weights)",
        "config": {"workload": f"{S} concurrent synthetic stereo streams per GPU, {args.ctx_sec} s / {hz} Hz (T={T}), 1 MI355X per rank",
                   "streams_per_gpu": S, "frame_hz": hz, "ctx_frames": T, "mode": args.mode, "gemm_arithmetic": ("fp32 MFMA; FFN block: f16x3 split products, fp32 accumulate" if args.split_f16 else "fp32 MFMA"), "parallelism": f"stream-sharded x{world}, no collective"},
        "realtime_streams_sustained": value / hz,
        "step_tflops": value * gflop_sf / 1e3,
        "step_frac_of_fp32_mfma_peak": value * gflop_sf / 1e3 / (FP32_MFMA_PEAK_TF * world),
        "roofline": {"bound": "mfma", "kernel": {"ffn_block": "ffn_block_kernel", "attention": "attn_block_kernel"}.get(dominant, f"gemm_f32_kernel ({dominant})"), "achieved": achieved_tf,
                     "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": achieved_tf / FP32_MFMA_PEAK_TF,
                     "traffic": traffic, "avg_launch_us": avg_launch_s * 1e6, "launches_per_step": launches_per_step,
                     "gflop_per_launch": flop_per_launch / 1e9},
        "kernel_ms_per_step": {k: round(v, 4) for k, v in sorted(breakdown.items(), key=lambda kv: -kv[1])},
    }

    if rank == 0 and world == 1 and not args.no_latency and not followers and args.groups <= 1:
        # the same workload with the tick split into two overlap groups that free-run across ticks (VAPX_DEFER_JOIN):
        # reported next to `value`, not as `value`, because co-running kernels stretch each other's launch time and the
        # per-kernel roofline above would stop meaning anything
        eng_g = engine.Engine(W.pack_blob(cpc, vap, modes[0]), hz, args.ctx_sec, max_streams=S, device_id=local_rank,
                              groups=2, mode=modes[0])
        for i in range(T + 5):
            eng_g.step_device(S, d_audio[i % NF].data_ptr(), hop, d_out.data_ptr(), stream=stream, defer_join=True)
        eng_g.join(stream)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(args.steps):
            eng_g.step_device(S, d_audio[i % NF].data_ptr(), hop, d_out.data_ptr(), stream=stream, defer_join=True)
        eng_g.join(stream)
        torch.cuda.synchronize()
        dtg = time.perf_counter() - t1
        result["overlap_groups"] = {"groups": 2, "defer_join": True, "value": S * args.steps / dtg, "unit": "frames/s",
                                    "ms_per_step": dtg / args.steps * 1e3}
        eng_g.close()

    if rank == 0 and world == 1 and not args.no_latency and not followers and not args.split_f16:
        # the same workload on the opt-in split-precision path (VAPX_FLAG_SPLIT_F16): every GEMM-shaped contraction as
        # three f16 MFMA products with fp32 accumulation — same deviation from the reference as the fp32-MFMA path
        # (tests/test_split_precision_gpu.py).  Reported next to `value`, never as `value`.
        eng_s = engine.Engine(W.pack_blob(cpc, vap, modes[0]), hz, args.ctx_sec, max_streams=S, device_id=local_rank,
                              mode=modes[0], split_f16=True)
        for i in range(T + 5):
            eng_s.step_device(S, d_audio[i % NF].data_ptr(), hop, d_out.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(args.steps):
            eng_s.step_device(S, d_audio[i % NF].data_ptr(), hop, d_out.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        dts = time.perf_counter() - t1
        assert torch.isfinite(d_out[:, :6]).all(), "non-finite outputs on the split-precision path"
        result["split_f16"] = {"value": S * args.steps / dts, "unit": "frames/s", "ms_per_step": dts / args.steps * 1e3,
                               "arithmetic": "x = hi + lo (f16); hi.hi + lo.hi + hi.lo on v_mfma_f32_32x32x16_f16, fp32 accumulate; "
                                             "FFN block, attention projections, conv / projection GEMMs; opt-in, not the default"}
        eng_s.close()

    if rank == 0 and not args.no_latency and not followers:
        # host-inclusive tick latency: host audio -> results on host (H2D + kernels + D2H + sync)
        lat = []
        for i in range(210):
            a = audio[i % NF]
            t1 = time.perf_counter()
            eng.step(a)
            lat.append((time.perf_counter() - t1) * 1e3)
        lat = np.array(lat[10:])
        result["latency_ms_host_inclusive"] = {"p50": float(np.percentile(lat, 50)), "p99": float(np.percentile(lat, 99)),
                                               "max": float(lat.max())}

    if rank == 0 and not args.no_latency and args.subtick_streams > 0 and hz == 20 and args.mode == "vap":
        # "concurrent streams at <= 10 ms/frame p99": a frame period (50 ms at 20 Hz) is filled with
        # phase-staggered sub-ticks; each sub-tick is host audio -> results on host.  Streams one GPU
        # sustains = sub-tick size x (sub-ticks that fit into one frame period at the p99 latency).
        Ssub = args.subtick_streams
        eng2 = engine.Engine(W.pack_blob(cpc, vap), hz, args.ctx_sec, max_streams=Ssub, device_id=local_rank)
        a2 = np.ascontiguousarray(np.concatenate([audio] * ((Ssub + S - 1) // S), axis=1)[:, :Ssub])
        for i in range(T):
            eng2.step(a2[i % NF])
        lat2 = []
        for i in range(210):                                      # 200 timed sub-ticks: p99 is a real percentile
            t1 = time.perf_counter()
            eng2.step(a2[i % NF])
            lat2.append((time.perf_counter() - t1) * 1e3)
        lat2 = np.array(lat2[10:])
        p99 = float(np.percentile(lat2, 99))
        period_ms = 1000.0 / hz
        result["concurrent_streams_at_10ms"] = {
            "sub_tick_streams": Ssub, "samples": int(lat2.size), "max_ms": float(lat2.max()), "p50_ms": float(np.percentile(lat2, 50)), "p99_ms": p99,
            "sub_ticks_per_frame_period": int(period_ms // p99),
            "sustained_streams": int(period_ms // p99) * Ssub if p99 <= 10.0 else 0,
            "note": "host-inclusive (pageable H2D + kernels + D2H + sync); streams = sub-tick size x floor(50 ms / p99)"}
        eng2.close()

    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.mode == "vap":
        from oracle.vap_oracle import ServerFramer, VapOracle
        torch.set_num_threads(1)
        o = VapOracle(cpc, vap, hz, args.ctx_sec)
        st, fr = o.new_state(1), ServerFramer(1, hop)
        one = audio[:, :1]                                        # [NF,1,2,hop]
        for i in range(T):                                        # fill the window (not timed)
            o.step(fr.frame(one[i % NF]), st)
        n, t1 = 0, time.perf_counter()
        while time.perf_counter() - t1 < args.cpu_baseline_sec:
            o.step(fr.frame(one[n % NF]), st)
            n += 1
        cdt = time.perf_counter() - t1
        result["cpu_baseline"] = {"value": n / cdt, "unit": "frames/s", "cores": 1, "kind": "port",
                                  "sample": f"{n} frames of 1 stream (batch 1, window full, torch-CPU fp32, 1 thread) in {cdt:.1f} s; host has {os.cpu_count()} logical cores",
                                  "ms_per_frame": cdt / n * 1e3}
        P = args.cpu_procs if args.cpu_procs >= 0 else max(1, min(16, (os.cpu_count() or 2) // 2))
        if P > 1:
            # the reference deployed on every core: P independent single-threaded processes (one stream each), as
            # SURVEY.md §8d asks; aggregate = sum of the per-process rates over the common window
            import multiprocessing as mp
            with mp.get_context("spawn").Pool(P) as pool:
                res = pool.map(_cpu_worker, [(i, hz, args.ctx_sec, 8.0) for i in range(P)])
            agg = sum(n_ / dt_ for n_, dt_ in res)
            result["cpu_baseline_multiprocess"] = {
                "value": agg, "unit": "frames/s", "cores": P, "kind": "port",
                "sample": f"{P} single-threaded oracle processes x 8 s, one stream each ({sum(r[0] for r in res)} frames); "
                          f"host has {os.cpu_count()} logical cores", "per_core": agg / P}
    for f in followers:
        f.close()
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
