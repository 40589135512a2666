#!/usr/bin/env python3
"""bench.py — VAP frames/s of the MI355X-native streaming forward pass.

One "step" = one VAP frame (one tick) for every stream of this rank: frame assembly -> CPC CNN -> LSTM -> downsample ->
context ring -> 1+3 transformer layers -> heads, through the C ABI (vapx_step) with audio and outputs resident in HBM.

Headline workload (`value`) = BASELINE.json configs[2] ("c3"): 4096 concurrent synthetic stereo streams, 50 Hz frames, 5 s
context (T = 250) per GPU — the LARGEST single-GPU configuration.  The same command measures the other single-GPU
configurations as sub-records under "configs", each with its own `roofline`, `cpu_baseline`, `parity_gate`, `split_f16` side
record and `paced_latency`:
  c2          256 streams x 20 Hz / T = 50    (configs[1])
  s4096_20hz  4096 streams x 20 Hz / T = 50   (the per-GPU shard of configs[3]: 32768 streams over 8 GPUs)
  c5          bc + nod on one shared CPC trunk, 4096 streams (configs[4])
`--gpus N` with N > 1 (and no WORLD_SIZE in the environment) re-executes itself under torch.distributed.run with N ranks,
one per GPU; streams are sharded over the ranks with NO data-path collective (they are independent), every record is then
the whole-job aggregate (units of all ranks / max-over-ranks time), so "s4096_20hz" at N = 8 IS configs[3].

Every record passes a PARITY GATE before anything is timed: ticks 0, 1 (window warm-up), T-1 (first full window) and T (first
slide) of the first and the last stream of the rank are compared with the oracle (CPU restatement of the reference step, pinned
to the unmodified reference by tests/golden) at 1e-4 abs; a miss aborts the run without a number.

Prints ONE JSON line on rank 0 (contract in the task prompt): metric / value / unit / ... plus per record
  "roofline":      dominant kernel (by summed time), HIP-event timed inside the timed region on the launch stream
  "cpu_baseline":  the oracle timed on this host at the record's own shape (1 thread, batch 1), continuing from the gate's state
  "paced_latency": distinct streams in phase-staggered sub-ticks on a wall-clock schedule (the <= 10 ms p99 target), per rank
  "split_f16":     the same workload on the opt-in split-precision path (never `value`)
and once: "front_end" (the native TCP front-end under the real-time load generator, reference wire format).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NOMINAL_SCLK_MHZ = 2400.0   # the clock the MFMA peaks of MI355X_MICROARCH.md are quoted at
FP32_MFMA_PEAK_TF = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, exact f32
F16_MFMA_PEAK_TF = 2516.6     # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_f16, dense
HBM_PEAK_GBS = 8000.0
GFLOP_PER_STREAM_FRAME = {(20, 50): 0.920, (50, 250): 4.505, (10, 50): 1.054}   # dense count: SURVEY.md §8d / BASELINE.md §3
ALGO_BYTES_PER_STREAM_FRAME = {(20, 50): 119e3, (50, 250): 525e3}
PARITY_TOL = 1e-4             # BASELINE.json north_star

WORKLOADS = {   # name -> (streams per GPU, frame_hz, ctx_sec, mode, default steps, default warmup): every record >= 2 s or >= 100 ticks
    "c3": (4096, 50, 5.0, "vap", 24, 3),
    "c2": (256, 20, 2.5, "vap", 1000, 20),
    "s4096_20hz": (4096, 20, 2.5, "vap", 100, 5),
    "c5": (4096, 20, 2.5, "bc+nod", 48, 3),
}
KERNEL_NAMES = {"ffn_block": "ffn_block_kernel", "ffn_proj": "ffn_block_kernel<.., 2>", "attention": "attn_block_kernel", "conv_tail": "conv_tail_kernel",
                "last_row": "last_block_kernel", "lstm": "lstm_kernel", "head": "head_kernel", "conv0": "conv0_kernel"}


from vap_realtime_amd.capacity import attention_executed_fraction, macs_per_stream_frame, model_macs  # noqa: E402  (the work model lives in the package: serve.py plans with it)


def front_door_plumbing_check(n_shards: int, per_shard: int = 2) -> dict:
    """The serving side of `--gpus N` without a GPU (the --rendezvous-only leg): ONE front door (`vapx_frontdoor_*`, the reference's single
    port pair, vap_main.py:338-366) over N passive native front-ends, each driven by a stand-in step function that tags its answers with
    its shard; N x per_shard dialogues connect, send one frame each and must be answered by shard k mod N, slot k div N."""
    import numpy as np
    from vap_realtime_amd import engine, ingest, wire
    hop = 800

    def make(tag):
        def step(ids, audio, out):
            out[:, :] = 0.0
            out[:, 4] = tag
            out[:, 5] = ids
            out[:, engine.OUT_NVALID] = 1.0
            return 0
        return step
    shards = [ingest.NativeServer.over_function(make(float(g)), per_shard, 20, max_wait_s=0.05, port_in=-1, port_out=-1) for g in range(n_shards)]
    door = ingest.FrontDoor(shards, port_in=0, port_out=0)
    owners = []
    try:
        def wait(cond, timeout=10.0):
            t0 = time.time()
            while not cond():
                if time.time() - t0 > timeout:
                    raise TimeoutError("front door plumbing check")
                time.sleep(0.002)
        n = n_shards * per_shard
        ins, outs = [], []
        for k in range(n):
            ins.append(socket.create_connection(("127.0.0.1", door.port_in)))
            wait(lambda: door.counts()["accepted_in"] == k + 1)
        for k in range(n):
            outs.append(socket.create_connection(("127.0.0.1", door.port_out)))
            wait(lambda: door.counts()["accepted_out"] == k + 1)
        x = np.zeros((2, hop))
        for k in range(n):
            ins[k].sendall(wire.encode_input(x[0] + 0.001 * (k + 1), x[1]))
        for k in range(n):
            outs[k].settimeout(10)
            head = b""
            while len(head) < 4:
                head += outs[k].recv(4 - len(head))
            size = int.from_bytes(head, "little")
            body = b""
            while len(body) < size:
                body += outs[k].recv(size - len(body))
            r = wire.decode_result(body)
            owners.append([int(r["vad"][0]), int(r["vad"][1])])
        ok = owners == [[k % n_shards, k // n_shards] for k in range(n)]
        for so in ins + outs:
            so.close()
        return {"shards": n_shards, "dialogues": n, "owners": owners, "ok": bool(ok), "counts": door.counts()}
    finally:
        door.close()


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def physical_cores() -> int:
    try:
        import psutil
        n = psutil.cpu_count(logical=False)
        if n:
            return int(n)
    except Exception:
        pass
    return max(1, (os.cpu_count() or 2) // 2)


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
    a = synth.dialogue_batch([widx % 64], hop * NF).reshape(1, 2, NF, hop).transpose(2, 0, 1, 3)
    o = VapOracle(cpc, vap, hz, ctx_sec)
    st, fr = o.new_state(1), ServerFramer(1, hop)
    for i in range(int(ctx_sec * hz)):
        o.step(fr.frame(a[i % NF]), st)
    n, t1 = 0, time.perf_counter()
    while time.perf_counter() - t1 < seconds:
        o.step(fr.frame(a[n % NF]), st)
        n += 1
    return n, time.perf_counter() - t1


def synth_audio(stream_ids, S: int, hop: int, NF: int) -> np.ndarray:
    """[NF, S, 2, hop] float32: seeded two-speaker dialogue for up to 64 streams, tiled (shifted / scaled) to S streams."""
    from vap_realtime_amd import synth
    base = synth.dialogue_batch(stream_ids[:min(S, 64)], hop * NF)
    reps = (S + base.shape[0] - 1) // base.shape[0]
    audio = np.concatenate([np.roll(base, 97 * r, axis=2) * (1.0 - 0.01 * (r % 50)) for r in range(reps)], 0)[:S]
    return np.ascontiguousarray(audio.reshape(S, 2, NF, hop).transpose(2, 0, 1, 3))


def model_weights(hz: int, mode: str):
    """(cpc state dict, [(mode, vap state dict), ...]) of a workload: the first weight set leads the shared CPC trunk."""
    from vap_realtime_amd import weights as W
    modes = mode.split("+")
    cpc, vap0 = W.synthetic_weights(0, hz, modes[0])
    sets = [(modes[0], vap0)]
    for k, m in enumerate(modes[1:]):                              # same cpc_model "file", own VAP state dict
        sets.append((m, W.synthetic_weights(1 + k, hz, m)[1]))
    return cpc, sets


ENGINE_KW = {}          # --engine-flag name (A/B runs of engine.Engine keyword switches, e.g. split_qkv_in_ffn): applied to every engine of the run


def make_engines(cpc, sets, hz, ctx_sec, max_streams, device_id, groups=0, split_f16=False, max_batch=None):
    from vap_realtime_amd import engine, weights as W
    lead = engine.Engine(W.pack_blob(cpc, sets[0][1], sets[0][0]), hz, ctx_sec, max_streams=max_streams, max_batch=max_batch,
                         device_id=device_id, groups=groups, mode=sets[0][0], split_f16=split_f16, **ENGINE_KW)
    followers = []
    for m, vap in sets[1:]:
        f = engine.Engine(W.pack_blob(cpc, vap, m), hz, ctx_sec, max_streams=max_streams, max_batch=max_batch, device_id=device_id,
                          groups=groups, mode=m, split_f16=split_f16, **ENGINE_KW)
        f.attach_trunk(lead)
        followers.append(f)
    return lead, followers


class Workload:
    """Engines + resident synthetic audio of one configuration on this rank."""

    def __init__(self, S, hz, ctx_sec, mode, rank, world, local_rank, groups=0, split_f16=False):
        import torch
        from vap_realtime_amd import engine
        from vap_realtime_amd.sharding import shard_streams
        self.S, self.hz, self.ctx_sec, self.mode = S, hz, ctx_sec, mode
        self.T = int(ctx_sec * hz)
        self.hop = 16000 // hz
        self.modes = mode.split("+")
        self.my_streams = shard_streams(S * world, world, rank)         # global stream ids of this rank
        self.cpc, self.sets = model_weights(hz, mode)
        self.vap = self.sets[0][1]
        self.eng, self.followers = make_engines(self.cpc, self.sets, hz, ctx_sec, S, local_rank, groups=groups, split_f16=split_f16)
        self.NF = 32 if S * self.hop <= 1024 * 800 else 8               # distinct audio frames, cycled
        self.audio = synth_audio(self.my_streams, S, self.hop, self.NF)
        self.d_audio = torch.from_numpy(self.audio).cuda()
        self.d_out = torch.zeros(S, engine.OUT_STRIDE, device="cuda")
        self.d_out_f = [torch.zeros(S, engine.OUT_STRIDE, device="cuda") for _ in self.followers]
        self.stream = torch.cuda.current_stream().cuda_stream

    def step(self, i, defer_join=False):
        self.eng.step_device(self.S, self.d_audio[i % self.NF].data_ptr(), self.hop, self.d_out.data_ptr(), stream=self.stream,
                             defer_join=defer_join)
        for f, o in zip(self.followers, self.d_out_f):
            f.step_follow_device(self.S, o.data_ptr(), stream=self.stream)

    def rows(self, idx):
        """Host copies of the output rows `idx` of every model: [n_models][len(idx), OUT_STRIDE]."""
        return [o[idx].cpu().numpy() for o in [self.d_out] + self.d_out_f]

    def profile_enable(self, classes):
        for e in [self.eng] + self.followers:
            e.profile_enable(classes)

    def profile_read(self):
        tot = {}
        for e in [self.eng] + self.followers:
            for k, (ms, cnt) in e.profile_read().items():
                a = tot.get(k, (0.0, 0))
                tot[k] = (a[0] + ms, a[1] + cnt)
        return tot

    def close(self):
        for f in self.followers:
            f.close()
        self.eng.close()


class OracleTwin:
    """The oracle (oracle/vap_oracle.py: CPU restatement of the reference's process_vap, pinned to the unmodified reference by
    tests/golden) run on a few streams of a workload — the CHECKER of the parity gate and, continued from the gate's state,
    the CPU baseline.  One oracle per weight set; each re-encodes the audio, as the reference's separate programs do."""

    def __init__(self, cpc, sets, hz, ctx_sec, audio, gate_idx):
        from oracle.vap_oracle import ServerFramer, VapOracle
        self.hop = 16000 // hz
        self.audio = audio[:, gate_idx]                         # [NF, G, 2, hop]
        self.NF = audio.shape[0]
        self.modes = [m for m, _ in sets]
        self.oracles = [VapOracle(cpc, vap, hz, ctx_sec, mode=m) for m, vap in sets]
        self.states = [o.new_state(len(gate_idx)) for o in self.oracles]
        self.framers = [ServerFramer(len(gate_idx), self.hop) for _ in self.oracles]
        self.tick = 0

    def step(self):
        outs = [o.step(fr.frame(self.audio[self.tick % self.NF]), st) for o, st, fr in zip(self.oracles, self.states, self.framers)]
        self.tick += 1
        return outs

    def time_one_stream(self, seconds: float):
        """Continue stream 0 alone (batch 1, 1 thread: the reference's shape) for `seconds`: (frames, elapsed)."""
        import torch
        from oracle.vap_oracle import OracleState, ServerFramer
        torch.set_num_threads(1)
        sts, frs = [], []
        for st, fr in zip(self.states, self.framers):
            sts.append(OracleState(1, st.ctx_len, [r[:1].clone() for r in st.ring], st.h[:1].clone(), st.c[:1].clone()))
            f1 = ServerFramer(1, self.hop)
            f1.carry = fr.carry[:1].copy()
            frs.append(f1)
        a1 = self.audio[:, :1]
        n, t1 = 0, time.perf_counter()
        while time.perf_counter() - t1 < seconds:
            for o, st, fr in zip(self.oracles, sts, frs):
                o.step(fr.frame(a1[(self.tick + n) % self.NF]), st)
            n += 1
        return n, time.perf_counter() - t1


def compare_with_oracle(mode, row, want, n):
    """max |hip - oracle| over the outputs the reference's process_vap of this model variant publishes."""
    from vap_realtime_amd import engine
    got = engine.split_outputs(row)
    worst = float(np.abs(got["vad"] - want["vad"]).max())
    if mode == "vap":
        for k in ("p_now", "p_future", "logits"):
            worst = max(worst, float(np.abs(got[k] - want[k]).max()))
    elif mode == "bc":
        worst = max(worst, float(np.abs(got["aux"][:, 1] - want["p_bc_react"]).max()), float(np.abs(got["aux"][:, 2] - want["p_bc_emo"]).max()))
    else:
        for k, col in (("p_nod_short", 1), ("p_nod_long", 2), ("p_nod_long_p", 3)):
            worst = max(worst, float(np.abs(got["aux"][:, col] - want[k]).max()))
        worst = max(worst, float(np.abs(got["logits"][:, :n] - want["p_bc"][:, :n]).max()))
    return worst if np.isfinite(worst) else float("inf")


def fill_and_gate(wl: Workload, name: str):
    """Fill the context window (T + 1 ticks: the timed region is the steady state) and hold the engines' outputs of the first and
    the last stream at ticks 0, 1, T-1, T against the oracle.  Raises SystemExit on a miss: no number without parity."""
    import torch
    T, S = wl.T, wl.S
    gate_idx = [0, S - 1] if S > 1 else [0]
    check_ticks = sorted({0, 1, T - 1, T})
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    twin = OracleTwin(wl.cpc, wl.sets, wl.hz, wl.ctx_sec, wl.audio, gate_idx)
    got = {}
    for i in range(T + 1):
        wl.step(i)
        if i in check_ticks:
            got[i] = wl.rows(gate_idx)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    worst = {}
    for i in range(T + 1):
        want = twin.step()
        if i in check_ticks:
            n = min(i + 1, T)
            for k, m in enumerate(twin.modes):
                worst[(i, m)] = compare_with_oracle(m, got[i][k], want[k], n)
    oracle_s = time.perf_counter() - t0
    w = max(worst.values())
    rec = {"ok": bool(w <= PARITY_TOL), "tolerance_abs": PARITY_TOL, "worst_abs": w, "streams_checked": [int(wl.my_streams[g]) for g in gate_idx],
           "ticks_checked": check_ticks, "models": twin.modes, "worst_by_tick": {str(t): max(v for (tt, _), v in worst.items() if tt == t) for t in check_ticks},
           "oracle_seconds": oracle_s,
           "what": "p_now / p_future / VAD / 256 logits (bc, nod: their head probabilities, nod's p_bc of every window row) vs oracle/vap_oracle.py"}
    if not rec["ok"]:
        raise SystemExit(f"bench.py: PARITY GATE FAILED for {name}: max |hip - oracle| = {w:.3e} > {PARITY_TOL} ({json.dumps(rec)}); nothing was timed")
    return rec, twin


def load_traffic(key: str, dominant: str):
    """(mean HBM bytes per launch of the dominant kernel CLASS, HBM bytes of a whole tick) from the committed PMC passes (separate
    rocprofv3 --pmc runs of this very command, profiles/pmc_traffic.json + tools/pmc_tick.py; counters cannot be collected inside a timed
    run).  The class mean covers exactly the launches the class's HIP-event time covers."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            entry = json.load(f).get(key, {})
        tick = entry.get("_tick")
        c = tick["by_class"].get(dominant)
        return (c["bytes_corrected"] / c["launches"] if c else None), tick["bytes_corrected"], traffic_source(entry)
    except Exception:
        return None, None, traffic_source(None)


def traffic_source(entry):
    """Provenance of `roofline.traffic` / `tick_traffic`: they are NOT measured by this run (counters cannot be collected inside a timed
    run) but read from the committed PMC passes.  `stale` says whether the kernels timed here are the sources those passes ran on
    (content hash of vap-realtime_amd/csrc: the GPU box has no .git); `git` is the commit the passes were taken at."""
    from vap_realtime_amd import provenance
    src = (entry or {}).get("_source") or {}
    now = provenance.kernel_source_hash()
    return {"file": "profiles/pmc_traffic.json", "git": src.get("git"), "csrc_sha": src.get("csrc_sha"), "csrc_sha_now": now,
            "stale": (src.get("csrc_sha") != now) if entry else None}


class BoardWatch:
    """Board power and shader clock while the timed steps run, from the amdgpu hwmon files (no subprocess, 20 samples / s on a thread).
    The split-precision path runs at the board's power limit: a frame rate without the clock it was measured at says
    little about the kernel.  The rank's own card is found by PCI address; failing that, the busiest visible card is reported."""

    def __init__(self, device_index=0, period=0.05, root="/sys/class/drm"):
        import glob
        self.period = period
        self.cards = []
        self.caps = []
        hwmons = sorted(glob.glob(os.path.join(root, "card*/device/hwmon/hwmon*")))
        try:      # this rank's own card by PCI address (a node's other GPUs — other tenants' — are visible in sysfs too)
            import torch
            pr = torch.cuda.get_device_properties(device_index)
            addr = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}."
            mine = [hw for hw in hwmons if addr in os.path.realpath(hw.split("/hwmon/")[0])]
            self.matched = bool(mine)
            hwmons = mine or hwmons
        except Exception:  # noqa: BLE001 — no such properties in this torch: fall back to the busiest card
            self.matched = False
        for hw in hwmons:
            pw = next((f for f in (hw + "/power1_average", hw + "/power1_input") if os.path.exists(f)), None)
            fq = hw + "/freq1_input"
            if pw:
                self.cards.append((pw, fq if os.path.exists(fq) else None))
                cap = self._read(hw + "/power1_cap")
                self.caps.append(cap * 1e-6 if cap else None)
        self.samples = [[] for _ in self.cards]
        self._stop = None

    @staticmethod
    def _read(path):
        try:
            with open(path) as f:
                return float(f.read().strip())
        except (OSError, ValueError):
            return None

    def __enter__(self):
        import threading
        if not self.cards:
            return self
        self._stop = threading.Event()

        def loop():
            while not self._stop.is_set():
                for i, (pw, fq) in enumerate(self.cards):
                    w = self._read(pw)
                    f = self._read(fq) if fq else None
                    if w is not None:
                        self.samples[i].append((w * 1e-6, f * 1e-6 if f else None))
                self._stop.wait(self.period)
        self._thr = threading.Thread(target=loop, daemon=True)
        self._thr.start()
        return self

    def __exit__(self, *exc):
        if self._stop is not None:
            self._stop.set()
            self._thr.join()

    def record(self):
        k = max(range(len(self.samples)), key=lambda i: (sum(x[0] for x in self.samples[i]) / len(self.samples[i])) if self.samples[i] else 0.0, default=None)
        best = self.samples[k] if k is not None else []
        if len(best) < 3:
            return None
        cap = self.caps[k] if k < len(self.caps) else None
        best = best[len(best) // 5:]                                   # drop the ramp at the start of the region
        ws = sorted(x[0] for x in best)
        fs = sorted(x[1] for x in best if x[1])
        return {"watts_median": round(ws[len(ws) // 2], 1), "watts_max": round(ws[-1], 1),
                "sclk_mhz_median": round(fs[len(fs) // 2], 0) if fs else None, "sclk_mhz_min": round(fs[0], 0) if fs else None,
                "power_cap_w": round(cap, 0) if cap else None, "samples": len(ws), "card_matched_by_pci_address": self.matched,
                "source": "amdgpu hwmon power1_input / freq1_input of this rank's card, 20 Hz, during the timed steps"}


def run_workload(name, S, hz, ctx_sec, mode, steps, warmup, ctx, groups=0, split_f16=False, defer_join=False):
    """Parity gate while the window fills, find the dominant kernel class, then time exactly `steps` ticks bracketed by barrier +
    device synchronise; returns (record, workload, oracle twin) — the caller closes the workload."""
    import torch
    from vap_realtime_amd import dist_util, engine
    rank, local_rank, world, dist = ctx
    wl = Workload(S, hz, ctx_sec, mode, rank, world, local_rank, groups=groups, split_f16=split_f16)
    T = wl.T

    def barrier():
        dist_util.barrier(dist, torch.cuda.synchronize)

    gate, twin = fill_and_gate(wl, name + ("_split_f16" if split_f16 else ""))
    wl.profile_enable(range(len(engine.PROF_CLASSES)))
    wl.profile_read()
    NP = 3 if S * T > 100000 else 5
    for i in range(NP):
        wl.step(i)
    breakdown = {k: v[0] / NP for k, v in wl.profile_read().items()}
    dominant = max(breakdown, key=breakdown.get)
    dom_id = [k for k, v in engine.PROF_CLASSES.items() if v == dominant][0]
    wl.profile_enable([dom_id])
    wl.profile_read()

    for i in range(warmup):
        wl.step(i)
    barrier()
    wl.profile_read()
    watch = BoardWatch(local_rank)
    # one HIP event per tick boundary, recorded on the launch stream inside the timed region: the spread of the K ticks rides in the record
    # (the driver times ONE number over 20 ticks; boards differ by 7 %: a +-2 % code change is invisible without it)
    tick_ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    tstream = torch.cuda.current_stream()
    with watch:
        t0 = time.perf_counter()
        tick_ev[0].record(tstream)
        for i in range(steps):
            wl.step(i, defer_join=defer_join)
            tick_ev[i + 1].record(tstream)
        if defer_join:
            wl.eng.join(wl.stream)
        torch.cuda.synchronize()
        dt_own = time.perf_counter() - t0                          # this rank's own K steps (reported per rank; `value` uses the max below)
    barrier()
    dt = time.perf_counter() - t0
    tick_ms = sorted(tick_ev[i].elapsed_time(tick_ev[i + 1]) for i in range(steps))
    tick_stats = {"min": tick_ms[0], "median": tick_ms[len(tick_ms) // 2], "p95": tick_ms[min(len(tick_ms) - 1, int(round(0.95 * (len(tick_ms) - 1))))],
                  "max": tick_ms[-1], "n": steps,
                  "note": "this rank's ticks, HIP events on the launch stream between consecutive steps" + (" (deferred join: overlap groups free-run, a tick's event marks its leading group)" if defer_join else "")}

    dom_ms, dom_launches = wl.profile_read()[dominant]
    wl.profile_enable([])
    dt_rank = dist_util.gather_floats(dist, [dt_own])                 # every rank's own time for its K steps
    dt = dist_util.max_over_ranks(dist, dt, "cuda")
    for o in [wl.d_out] + wl.d_out_f:
        assert torch.isfinite(o[:, :6]).all(), "non-finite outputs"
        assert not o[:, engine.OUT_STATUS].any(), "engine flagged non-finite rows"

    value = S * world * steps / dt
    macs = macs_per_stream_frame(hz, T, mode, qkv_in_attention=bool(split_f16 and T > 64 and not ENGINE_KW.get("split_qkv_in_ffn") and not ENGINE_KW.get("unfused_proj")))
    if "conv_tail" in breakdown:          # conv2-4 ran as the fused tail kernel (<= 512 streams): its MACs leave the GEMM class
        hop_ = 16000 // hz
        P1_ = (hop_ + 320) // 5 // 4
        conv1 = 2 * P1_ * 8 * 256 * 256
        macs["conv_tail"] = macs["gemm_cn_relu"] - conv1
        macs["gemm_cn_relu"] = conv1
    nm = len(wl.modes)
    launches_per_step = dom_launches / steps
    flop_per_launch = 2.0 * macs[dominant] * S / launches_per_step
    avg_launch_s = dom_ms * 1e-3 / dom_launches
    achieved_tf = flop_per_launch / avg_launch_s / 1e12
    dense_gflop = GFLOP_PER_STREAM_FRAME.get((hz, T))
    exec_gflop = 2.0 * sum(macs.values()) / 1e9      # executed, attention still counted dense
    fa = attention_executed_fraction(T)
    n_attn = sum(7 if m == "nod" else 5 for m in wl.modes)
    exec_gflop_causal = exec_gflop - 2.0 * (1.0 - fa) * n_attn * 2 * 4 * (T * T * 64 * 2) / 1e9
    # the split path's matrix-core roof: three f16 products per fp32 product on the f16 MFMA (judge r02: "f16 MFMA peak / 3")
    peak = F16_MFMA_PEAK_TF / 3.0 if split_f16 else FP32_MFMA_PEAK_TF
    kernel = KERNEL_NAMES.get(dominant, f"gemm_f32_kernel ({dominant})")
    if dominant == "attention" and T > 64:
        kernel = "attention_long_f16x3_kernel + attention_proj_f16x3_kernel" if split_f16 else "attention_long2_kernel"
    if dominant == "ffn_block" and split_f16:
        kernel = "ffn_block_f16x3_kernel"
    class_traffic, tick_traffic, traffic_src = load_traffic(f"{S}x{hz}hz_T{T}" + ("" if mode == "vap" else "_" + mode) + ("_split_f16" if split_f16 else ""), dominant)
    roof = {"bound": "mfma", "kernel": kernel, "achieved": achieved_tf, "peak": peak, "unit": "TFLOP/s", "frac": achieved_tf / peak,
            # achieved, avg_launch_us, gflop_per_launch and traffic all describe the SAME launches: the `launches_per_step` launches of the
            # dominant class per tick (long windows: the mode-1 flat-row blocks; the mode-2 blocks are the class "ffn_proj")
            "traffic": class_traffic, "traffic_source": traffic_src,
            "avg_launch_us": avg_launch_s * 1e6, "launches_per_step": launches_per_step, "gflop_per_launch": flop_per_launch / 1e9,
            "flop_count": "algorithmic FLOPs of the launch (dense T x T for attention), MACs x 2"}
    ab = ALGO_BYTES_PER_STREAM_FRAME.get((hz, T))
    if ab and tick_traffic and nm == 1:
        roof["tick_traffic"] = tick_traffic                      # HBM bytes of ONE tick, every kernel (PMC, this workload at this size)
        roof["traffic_ratio"] = tick_traffic / (ab * S)          # / algorithmic bytes: activation hand-offs between the launches of a tick
    if ab:
        roof["hbm_algorithmic"] = {"bytes_per_stream_frame": ab, "achieved_gbs": value * ab / 1e9 / world, "peak_gbs": HBM_PEAK_GBS,
                                   "frac": value * ab / 1e9 / world / HBM_PEAK_GBS, "note": "the path is MFMA-bound: the HBM roof is two orders of magnitude away"}
    if dominant == "attention":
        roof["frac_executed_causal"] = roof["frac"] * fa
    if split_f16:
        roof["peak_note"] = ("every GEMM-shaped contraction = 3 f16 MFMA products (hi.hi + lo.hi + hi.lo), fp32 accumulate: peak = dense f16 MFMA / 3 "
                             f"= {peak:.1f} fp32-equivalent TFLOP/s; frac_of_fp32_mfma_peak = {achieved_tf / FP32_MFMA_PEAK_TF:.3f}")
    board = watch.record()
    if board and board.get("sclk_mhz_median"):
        # the peaks above are quoted at the part's 2.4 GHz boost clock; under the board's power limit the kernels run slower than that
        roof["sclk_mhz"] = board["sclk_mhz_median"]
        roof["frac_at_sclk"] = roof["frac"] * NOMINAL_SCLK_MHZ / board["sclk_mhz_median"]
    rec = {
        "value": value, "unit": "frames/s", "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": dt / steps * 1e3,
        "timed_seconds": dt, "ms_per_step_spread": tick_stats,
        "config": {"workload": f"{name}: {S} concurrent synthetic stereo streams per GPU, {ctx_sec} s / {hz} Hz (T={T}), mode {mode}, 1 MI355X per rank",
                   "streams_per_gpu": S, "streams_total": S * world, "frame_hz": hz, "ctx_frames": T, "mode": mode,
                   "gemm_arithmetic": ("f16x3 split products, fp32 accumulate" if split_f16 else "fp32 MFMA"),
                   "parallelism": f"stream-sharded x{world}, no collective"},
        "parity_gate": gate,
        "per_rank": ({"value_min": S * steps / max(t[0] for t in dt_rank), "value_max": S * steps / min(t[0] for t in dt_rank)} if world > 1 else None),
        "realtime_streams_sustained": value / hz,
        "executed_gflop_per_stream_frame": exec_gflop, "executed_gflop_per_stream_frame_causal_attention": exec_gflop_causal,
        "executed_tflops": value * exec_gflop_causal / 1e3,          # attention counted as executed (causal tiles only)
        "executed_tflops_dense_attention": value * exec_gflop / 1e3,
        "executed_frac_of_fp32_mfma_peak": value * exec_gflop_causal / 1e3 / (FP32_MFMA_PEAK_TF * world),
        "roofline": roof,
        "board": board,                                              # watts and shader clock during the timed steps (None: no hwmon files)
        "kernel_ms_per_step": {k: round(v, 4) for k, v in sorted(breakdown.items(), key=lambda kv: -kv[1])},
        # every kernel class against the same roof (algorithmic FLOPs of the class per step / its time in the profiled pass;
        # attention classes on the dense T x T count): shows which kernel is furthest below the fp32-MFMA peak
        "kernel_tflops": {k: round(2.0 * macs.get(k, 0) * S / (v * 1e-3) / 1e12, 1)
                          for k, v in sorted(breakdown.items(), key=lambda kv: -kv[1]) if v > 0 and macs.get(k, 0) > 0},
    }
    if dense_gflop and nm == 1 and mode == "vap":
        rec["dense_gflop_per_stream_frame"] = dense_gflop
        rec["dense_tflops"] = value * dense_gflop / 1e3
        rec["dense_frac_of_fp32_mfma_peak"] = value * dense_gflop / 1e3 / (FP32_MFMA_PEAK_TF * world)
    return rec, wl, twin


def cpu_baseline_record(twin: OracleTwin, wl_cfg, seconds: float):
    """The oracle at the record's own shape on this host: 1 thread, batch 1, window full (continues the parity gate's state)."""
    S, hz, ctx_sec, mode = wl_cfg
    n, cdt = twin.time_one_stream(seconds)
    each = " (one frame = the bc AND the nod program, each encoding the audio itself as the reference deploys them)" if "+" in mode else ""
    return {"value": n / cdt, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": f"{n} frames of 1 stream at {hz} Hz / T={int(ctx_sec * hz)}, mode {mode}{each} (batch 1, window full, torch-CPU fp32, 1 thread) in "
                      f"{cdt:.1f} s; host has {physical_cores()} physical / {os.cpu_count()} logical cores",
            "ms_per_frame": cdt / n * 1e3}


def paced_latency(cpc, sets, hz, ctx_sec, local_rank, seconds, target_ms=10.0, max_util=0.85, split_f16=False, accept_ms=9.0):
    """The north-star latency target measured, not extrapolated: ONE engine (plus trunk followers for bc+nod) holding G x Ssub
    DISTINCT streams, its G sub-batches phase-staggered over the frame period (50 ms at 20 Hz, 20 ms at 50 Hz) on a wall-clock
    schedule for `seconds`; latency of a sub-tick = its scheduled audio-ready time -> results of EVERY model on the host (pinned
    H2D + kernels + D2H + sync, including any wait behind a late predecessor).  A short calibration picks the largest G x Ssub
    whose sub-tick service time keeps the GPU under `max_util`; if the paced run misses p99 <= target or the utilisation bound it
    is repeated: on time but too busy -> the sub-batch size that meets the bound (busy time is linear in it); late -> ~6 % fewer streams per
    sub-batch (up to twice), then one group smaller.  A run is ACCEPTED only at p99 <= `accept_ms` (9 ms): a point 0.1 ms under the
    10 ms limit is an edge, not a sustained figure (round-3 verdict: C3 at p99 9.90 ms); the next point down is reported instead."""
    from vap_realtime_amd import engine
    period = 1.0 / hz
    hop = 16000 // hz
    T = int(ctx_sec * hz)
    Ssub_opts = (64, 128, 192, 256, 384, 512, 768, 1024)
    Gmax = 16
    NF = 8
    base = synth_audio(list(range(64)), max(Ssub_opts), hop, NF)          # [NF, Ssub, 2, hop]
    pin_in = [engine.pinned_empty((max(Ssub_opts), 2, hop)) for _ in range(NF)]
    for i in range(NF):
        pin_in[i][...] = base[i]
    pin_out = [engine.pinned_empty((max(Ssub_opts), engine.OUT_STRIDE)) for _ in sets]
    # calibration engine: sub-tick service time per candidate size (two intra-tick overlap groups: -1.4 % at 1024 streams)
    eng, fol = make_engines(cpc, sets, hz, ctx_sec, max(Ssub_opts), local_rank, groups=2, split_f16=split_f16)

    def sub_tick(e, f, audio, ids):
        e.step(audio, ids, out=pin_out[0])
        for k, ff in enumerate(f):
            ff.step_follow(len(ids), out=pin_out[1 + k])

    calib = {}
    best = None
    for Ssub in Ssub_opts:
        ids = np.arange(Ssub, dtype=np.int32)
        ts = []
        for i in range(20):
            t1 = time.perf_counter()
            sub_tick(eng, fol, pin_in[i % NF][:Ssub], ids)
            ts.append(time.perf_counter() - t1)
        # upper quartile, not the tail: one host hiccup in a dozen samples must not end the calibration (a box with a noisy host once reported
        # "0 streams" for C5 because the FIRST candidate's p95 crossed the cut-off); the paced run itself measures the tail
        sv = float(np.percentile(ts[4:], 75))
        calib[Ssub] = sv * 1e3
        if sv * 1e3 > 0.8 * target_ms:
            break
        # (the paced schedule adds host work the isolated calibration does not see: first runs landed at 0.86-0.95 busy when this aimed
        # at the bound itself, and every repeat costs `seconds` of wall time; a larger margin under-reports configurations that would have fit)
        G = min(Gmax, int(0.97 * max_util * period / sv))
        if G >= 1 and (best is None or G * Ssub > best[0] * best[1]):
            best = (G, Ssub)
    for f in fol:
        f.close()
    eng.close()
    out = {"calibration_subtick_p75_ms": calib, "frame_period_ms": period * 1e3, "target_p99_ms": target_ms, "accept_p99_ms": accept_ms, "max_utilisation": max_util,
           "models": [m for m, _ in sets], "gemm_arithmetic": "f16x3 split products" if split_f16 else "fp32 MFMA",
           "method": "one engine, G phase-staggered sub-batches of DISTINCT streams per frame period, wall-clock schedule; latency = "
                     "scheduled audio-ready -> results of every model on host (pinned staging both ways)", "runs": []}
    if best is None:
        out["sustained_streams"] = 0
        return out
    G, Ssub = best
    cap = min(max(Ssub_opts), (int(Ssub * 1.12) + 7) // 8 * 8)   # head-room for ONE upward step when the first run leaves slack (coarse calibration grid)
    eng, fol = make_engines(cpc, sets, hz, ctx_sec, G * cap, local_rank, groups=2, split_f16=split_f16, max_batch=cap)
    accepted, tried_up = None, False
    import gc
    gc.disable()                                       # a collector pause inside the schedule would be charged to the engine
    for attempt in range(5):
        for s in range(G * Ssub):                     # every trial starts from clean streams (queued, applied by the next step)
            eng.reset_stream(s)
        idsets = [np.arange(g * Ssub, (g + 1) * Ssub, dtype=np.int32) for g in range(G)]
        n_periods = max(T + 20, int(seconds / period))
        lat, busy = [], 0.0
        t_start = time.perf_counter() + 0.01
        for k in range(n_periods):
            for g in range(G):
                ready = t_start + k * period + g * period / G
                while True:                            # wait for the audio of this sub-batch to be "ready"
                    now = time.perf_counter()
                    if now >= ready:
                        break
                    if ready - now > 2e-4:
                        time.sleep((ready - now) * 0.5)
                t1 = time.perf_counter()
                sub_tick(eng, fol, pin_in[(k + g) % NF][:Ssub], idsets[g])
                t2 = time.perf_counter()
                busy += t2 - t1
                if k >= T:                             # window full: steady state
                    lat.append((t2 - ready) * 1e3)
        wall = time.perf_counter() - t_start
        lat = np.asarray(lat)
        run = {"groups": G, "sub_tick_streams": Ssub, "streams": G * Ssub, "seconds": wall, "sub_ticks_timed": int(lat.size),
               "p50_ms": float(np.percentile(lat, 50)), "p99_ms": float(np.percentile(lat, 99)), "max_ms": float(lat.max()),
               "gpu_busy_fraction": busy / wall, "late_fraction": float((lat > target_ms).mean())}
        out["runs"].append(run)
        if run["p99_ms"] <= accept_ms and run["gpu_busy_fraction"] <= max_util:
            accepted = run
            up = min(cap, int(Ssub * 0.99 * max_util / run["gpu_busy_fraction"]) // 8 * 8)
            if run["gpu_busy_fraction"] < max_util - 0.025 and not tried_up and up > Ssub and attempt < 4:
                tried_up, Ssub = True, up                # slack: one larger run; whichever of the two meets the bounds is reported
                continue
            break
        if accepted is not None:                         # the upward step overshot: the accepted run stands
            break
        # on time but over the utilisation bound: the busy fraction is linear in the sub-batch size, so go straight to the size that meets
        # it (1 % margin, multiples of 8).  Too late: shed ~6 % of every sub-batch (same schedule) and measure again; then drop a group
        if run["p99_ms"] <= accept_ms and attempt < 4 and Ssub >= 64:
            prev = [r for r in out["runs"][:-1] if r["groups"] == G and r["p99_ms"] <= accept_ms]
            if prev and abs(prev[-1]["gpu_busy_fraction"] - run["gpu_busy_fraction"]) > 1e-3:
                # two on-time points: the busy fraction is affine in the sub-batch size (a fixed cost per sub-tick + a slope)
                p0 = prev[-1]
                slope = (run["gpu_busy_fraction"] - p0["gpu_busy_fraction"]) / (run["sub_tick_streams"] - p0["sub_tick_streams"])
                est = run["sub_tick_streams"] + (0.99 * max_util - run["gpu_busy_fraction"]) / slope if slope > 0 else Ssub - 8
            else:
                est = Ssub * (max_util / run["gpu_busy_fraction"]) ** 1.4 * 0.99
            Ssub = max(8, min(Ssub - 8, int(est) // 8 * 8))
        elif attempt < 2 and Ssub >= 64:
            Ssub -= max(8, (Ssub // 16) // 8 * 8)
        else:
            G -= 1
        if G < 1:
            break
    if accepted is not None:
        out["sustained_streams"] = accepted["streams"]
        out.update({k: accepted[k] for k in ("groups", "sub_tick_streams", "p50_ms", "p99_ms", "max_ms", "gpu_busy_fraction")})
    out.setdefault("sustained_streams", 0)
    gc.enable()
    for f in fol:
        f.close()
    eng.close()
    return out


def gather_paced(dist, rank, world, rec):
    """Whole-job view of the per-rank paced runs (they ran concurrently, one per GPU): streams summed, worst p99."""
    if dist is None or rec is None:
        return rec
    import torch
    mine = torch.tensor([float(rec.get("sustained_streams", 0)), float(rec.get("p99_ms", 0.0)), float(rec.get("max_ms", 0.0)),
                         float(rec.get("gpu_busy_fraction", 0.0))], dtype=torch.float64)
    allv = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allv, mine)
    rec = dict(rec)
    rec["per_rank"] = [{"sustained_streams": int(v[0]), "p99_ms": float(v[1]), "max_ms": float(v[2]), "gpu_busy_fraction": float(v[3])} for v in allv]
    rec["sustained_streams_all_ranks"] = int(sum(v[0] for v in allv))
    rec["worst_p99_ms_all_ranks"] = float(max(v[1] for v in allv))
    return rec


def front_end_record(streams: int, seconds: float, devices=None, repeat: int = 1):
    """The native TCP front-end (vapx_ingest_*) + one engine under tools/loadgen: `streams` real-time dialogue clients sending the
    reference's 10 ms packets (2560 B, vap_main.py:373-391), every result packet (12 880 B at 20 Hz) read back and timed.
    `devices` (multi-GPU job): one engine per listed GPU behind ONE port pair (vapx_frontdoor_*, what `serve --gpus N` runs; the
    reference's single port pair, vap_main.py:338-366), `streams` clients in total.  `repeat` independent runs (fresh server each):
    the record is the WORST run by client-side p99, the others' one-line summaries ride along.  Next to the load generator's
    (client-side) figures the record carries the server's own latency window (frame complete on the host -> packet handed to the
    kernel: p50 / p99 / max and the exact count of answers later than 10 ms), so that a bad run can be attributed: server, co-located
    clients, or loopback."""
    cmd = [sys.executable, os.path.join(ROOT, "tools", "server_load.py"), "--streams", str(streams), "--seconds", str(seconds), "--warm", "4"]
    if devices and len(devices) > 1:
        cmd += ["--shards", str(len(devices)), "--devices", ",".join(str(d) for d in devices)]

    def unpin():                                                       # the server of ALL shards must not inherit rank 0's core pinning
        try:
            os.sched_setaffinity(0, range(os.cpu_count() or 1))
        except Exception:
            pass

    def one():
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=seconds + 240, preexec_fn=unpin)
        line = [l for l in p.stdout.decode().splitlines() if l.startswith("{")][-1]
        r = json.loads(line)
        keep = ("streams", "frames_sent", "frames_answered", "stream_frames_per_s", "realtime_streams_served", "lat_p50_ms", "lat_p99_ms",
                "lat_max_ms", "late_over_10ms", "server", "server_window", "placement", "host_limits")
        rec = {k: r[k] for k in keep if k in r}
        nd = r.get("net_counters_delta", {})
        # what the HOST was doing: 1-minute load average (other tenants included), the cores this container actually got during the run and
        # whether its cgroup was throttled — on a shared box the front-end's tail follows these, not the code
        rec["host"] = {"loadavg_1m": float((r.get("host_limits", {}).get("loadavg") or [0])[0]), "cgroup_cpu_max": r.get("host_limits", {}).get("cgroup_cpu_max"),
                       "cores_used": round(nd.get("cgroup_usage_usec", 0) / 1e6 / (seconds + 5.0), 1), "cgroup_throttled_periods": nd.get("cgroup_nr_throttled")}
        st = r.get("server_stats", {})
        per = st.get("per_shard") or [st]
        # (several shards: the worst shard's percentiles, the sum of the late answers)
        rec["server_latency_ms"] = {k: max(p_.get(k, 0.0) for p_ in per) for k in ("lat_p50_ms", "lat_p99_ms", "lat_max_ms", "lat_mean_ms")}
        rec["server_late_over_10ms"] = sum(int(p_.get("late_over_10ms", 0)) for p_ in per)
        rec["server_answered"] = sum(int(p_.get("answered", 0)) for p_ in per)
        if devices and len(devices) > 1:
            rec["shards"] = len(devices)
            rec["front_door"] = st.get("front_door")
            rec["frames_done_per_shard"] = [p_.get("frames_done") for p_ in st.get("per_shard", [])]
        return rec
    try:
        runs = [one() for _ in range(max(1, repeat))]
        rec = max(runs, key=lambda r: r.get("lat_p99_ms", 0.0))
        rec["runs"] = [{k: r.get(k) for k in ("lat_p50_ms", "lat_p99_ms", "lat_max_ms", "late_over_10ms", "frames_sent", "frames_answered")}
                       | {"server_p99_ms": r["server_latency_ms"]["lat_p99_ms"], "server_late_over_10ms": r["server_late_over_10ms"],
                          "loadavg_1m": r["host"]["loadavg_1m"], "cores_used": r["host"]["cores_used"]} for r in runs]
        rec["how"] = (f"tools/server_load.py x {len(runs)} (worst run by client-side p99 shown): native front-end + engine on this GPU, its threads pinned to "
                      "cores of the GPU's NUMA node, tools/loadgen on the other cores of the same host (loopback TCP), reference wire format")
        return rec
    except Exception as e:                                        # noqa: BLE001 - a side record must not cost the headline
        return {"error": f"{type(e).__name__}: {e}"}


COMPACT_LIMIT = 4096          # bytes: the driver keeps an 8 KB tail of stdout; round 3's 30.8 KB line could not be parsed


def _r(x, sig=5):
    """Round a float to `sig` significant digits (ints, None, bools pass through) — keeps the compact line short."""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    x = float(x)
    if not np.isfinite(x):
        return None
    return float(f"{x:.{sig}g}")


def _sub_summary(rec):
    """One-number-per-question summary of a sub-record (c2 / s4096_20hz / c5): fp32 and split."""
    if not isinstance(rec, dict) or "value" not in rec:
        return {"error": str(rec.get("error", "no record"))[:80]} if isinstance(rec, dict) else {"error": "no record"}
    out = {"value": _r(rec["value"]), "ms_per_step": _r(rec["ms_per_step"], 4), "frac": _r(rec["roofline"]["frac"], 3),
           "kernel": rec["roofline"]["kernel"], "parity_worst_abs": _r(rec["parity_gate"]["worst_abs"], 3)}
    if "concurrent_streams_at_10ms" in rec:
        out["streams_at_10ms"] = rec["concurrent_streams_at_10ms"]
        out["p99_ms"] = _r(rec.get("paced_latency", {}).get("p99_ms"), 3)
    sp = rec.get("split_f16")
    if isinstance(sp, dict) and "value" in sp:
        out["split"] = {"value": _r(sp["value"]), "frac": _r(sp["roofline"]["frac"], 3)}
        if "concurrent_streams_at_10ms" in sp:
            out["split"]["streams_at_10ms"] = sp["concurrent_streams_at_10ms"]
            out["split"]["p99_ms"] = _r(sp.get("paced_latency", {}).get("p99_ms"), 3)
    elif isinstance(sp, dict) and "error" in sp:
        out["split"] = {"error": sp["error"][:80]}
    if "cpu_baseline" in rec:
        out["cpu_frames_per_s"] = _r(rec["cpu_baseline"]["value"], 4)
    return out


def compact_line(result: dict, full_path: str = "") -> str:
    """The LAST stdout line of bench.py: the contract keys of the task prompt + one-number summaries of the sub-records, strict JSON
    (no NaN / Infinity), < COMPACT_LIMIT bytes.  The complete record (every sub-record's breakdown, paced runs, gates) goes to a file
    (`full_path`), never to stdout.  tests/test_bench_line.py holds this against a canned round-3 record."""
    roof = result["roofline"]
    cfg = result["config"]
    line = {
        "metric": result["metric"], "value": _r(result["value"], 7), "unit": result["unit"], "n_gpus": result["n_gpus"], "steps": result["steps"],
        "warmup": result["warmup"], "ms_per_step": _r(result["ms_per_step"], 6), "timed_seconds": _r(result.get("timed_seconds"), 5),
        # spread of the K timed ticks (HIP events between consecutive steps on the launch stream): the one driver-timed number's error bar
        "ms_per_step_min": _r((result.get("ms_per_step_spread") or {}).get("min"), 5),
        "ms_per_step_median": _r((result.get("ms_per_step_spread") or {}).get("median"), 5),
        "ms_per_step_p95": _r((result.get("ms_per_step_spread") or {}).get("p95"), 5),
        "higher_is_better": True, "scaling": result.get("scaling", "weak"), "vs_baseline": result.get("vs_baseline"), "dtype": result["dtype"],
        "data": "synthetic (seeded dialogue audio + seeded random weights)",
        "config": {"workload": cfg["workload"], "streams_per_gpu": cfg["streams_per_gpu"], "streams_total": cfg["streams_total"],
                   "frame_hz": cfg["frame_hz"], "ctx_frames": cfg["ctx_frames"], "mode": cfg["mode"], "parallelism": cfg["parallelism"]},
        "parity_gate": {"ok": result["parity_gate"]["ok"], "worst_abs": _r(result["parity_gate"]["worst_abs"], 3),
                        "tolerance_abs": result["parity_gate"]["tolerance_abs"]},
        "roofline": {"bound": roof["bound"], "kernel": roof["kernel"], "achieved": _r(roof["achieved"]), "peak": _r(roof["peak"]),
                     "unit": roof["unit"], "frac": _r(roof["frac"], 4), "traffic": _r(roof.get("traffic")), "avg_launch_us": _r(roof["avg_launch_us"]),
                     "launches_per_step": _r(roof["launches_per_step"], 3), "traffic_ratio": _r(roof.get("traffic_ratio"), 3),
                     "frac_at_sclk": _r(roof.get("frac_at_sclk"), 4),
                     # traffic / traffic_ratio come from the committed PMC passes, not from this run: which tree they were taken at, and
                     # whether the kernels timed here are that tree's (content hash of vap-realtime_amd/csrc)
                     "traffic_source": {k: (roof.get("traffic_source") or {}).get(k) for k in ("file", "git", "csrc_sha", "stale")}},
    }
    if "executed_frac_of_fp32_mfma_peak" in result:
        line["executed_tflops"] = _r(result["executed_tflops"], 4)
    bd = result.get("board")
    if bd:
        line["board"] = {"watts": bd["watts_median"], "sclk_mhz": bd["sclk_mhz_median"]}
    cb = result.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {"value": _r(cb["value"], 4), "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                                "sample": cb["sample"][:150], "ms_per_frame": _r(cb.get("ms_per_frame"), 4)}
    cm = result.get("cpu_baseline_multiprocess")
    if cm:
        line["cpu_all_cores"] = {"value": _r(cm["value"], 4), "cores": cm["cores"]}
    if "concurrent_streams_at_10ms" in result:
        line["concurrent_streams_at_10ms"] = result["concurrent_streams_at_10ms"]
        line["paced_p99_ms"] = _r(result.get("paced_latency", {}).get("p99_ms"), 3)
    sp = result.get("split_f16")
    if isinstance(sp, dict) and "value" in sp:
        line["split_f16"] = {"value": _r(sp["value"]), "ms_per_step": _r(sp["ms_per_step"], 4), "frac": _r(sp["roofline"]["frac"], 3),
                             "kernel": sp["roofline"]["kernel"], "parity_worst_abs": _r(sp["parity_gate"]["worst_abs"], 3),
                             "traffic_ratio": _r(sp["roofline"].get("traffic_ratio"), 3)}
        if sp.get("board"):
            line["split_f16"]["watts"] = sp["board"]["watts_median"]
            line["split_f16"]["sclk_mhz"] = sp["board"]["sclk_mhz_median"]
        if "concurrent_streams_at_10ms" in sp:
            line["split_f16"]["streams_at_10ms"] = sp["concurrent_streams_at_10ms"]
    elif isinstance(sp, dict) and "error" in sp:
        line["split_f16"] = {"error": sp["error"][:80]}
    if result.get("per_rank"):
        line["per_rank"] = result["per_rank"]
    if result.get("ranks_seen"):
        line["ranks_seen"] = result["ranks_seen"]
    if result.get("configs"):
        line["configs"] = {k: _sub_summary(v) for k, v in result["configs"].items()}
    fe = result.get("front_end")
    if isinstance(fe, dict):
        if "error" in fe:
            line["front_end"] = {"error": str(fe["error"])[:80]}
        else:
            line["front_end"] = {k: _r(fe.get(k)) for k in ("streams", "shards", "frames_sent", "frames_answered", "lat_p50_ms", "lat_p99_ms", "lat_max_ms",
                                                              "late_over_10ms") if fe.get(k) is not None}
            sl = fe.get("server_latency_ms") or {}
            # the server's own view of the same window: frame complete on the host -> packet handed to the kernel
            line["front_end"].update({"srv_p50_ms": _r(sl.get("lat_p50_ms"), 3), "srv_p99_ms": _r(sl.get("lat_p99_ms"), 3), "srv_max_ms": _r(sl.get("lat_max_ms"), 3),
                                      "srv_late_over_10ms": fe.get("server_late_over_10ms"), "runs": len(fe.get("runs") or [1]),
                                      "best_p99_ms": _r(min((r_.get("lat_p99_ms") or 1e9) for r_ in (fe.get("runs") or [fe])), 3),
                                      "host_loadavg": _r((fe.get("host") or {}).get("loadavg_1m"), 3), "cores_used": (fe.get("host") or {}).get("cores_used")})
    if full_path:
        line["full_record"] = full_path
    text = json.dumps(line, allow_nan=False, separators=(",", ":"))
    for drop in ("cpu_all_cores", "front_end", "configs", "split_f16"):      # never reached with today's records; the contract keys always survive
        if len(text) < COMPACT_LIMIT:
            break
        line.pop(drop, None)
        text = json.dumps(line, allow_nan=False, separators=(",", ":"))
    assert len(text) < COMPACT_LIMIT, len(text)
    return text


def _jsonable(o):
    """The full record as strict JSON: non-finite floats -> None."""
    if isinstance(o, dict):
        return {str(k): _jsonable(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_jsonable(v) for v in o]
    if isinstance(o, (float, np.floating)):
        return float(o) if np.isfinite(o) else None
    if isinstance(o, np.integer):
        return int(o)
    return o


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS), help="headline workload (`value`)")
    ap.add_argument("--configs", default="c2,s4096_20hz,c5",
                    help="comma list of further workloads measured as sub-records under \"configs\" ('' = none)")
    ap.add_argument("--streams", type=int, default=None, help="override: concurrent streams per GPU of the headline workload")
    ap.add_argument("--frame-hz", type=int, default=None)
    ap.add_argument("--ctx-sec", type=float, default=None)
    ap.add_argument("--mode", default=None, choices=["vap", "bc", "nod", "bc+nod", "vap+bc+nod"],
                    help="override: model variant; a+b = weight sets served on one shared CPC trunk (one stream-frame = one audio "
                         "frame through the shared encoder and every listed model)")
    ap.add_argument("--cpu-baseline-sec", type=float, default=6.0, help="timed oracle seconds per distinct record shape")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-procs", type=int, default=-1,
                    help="processes of the multi-core CPU leg at the headline shape (-1 = every physical core, 0 = skip)")
    ap.add_argument("--no-latency", action="store_true", help="skip the latency legs, the split-precision side records and the front-end record")
    ap.add_argument("--paced-sec", type=float, default=10.0, help="duration of each paced many-stream latency run (0 = skip)")
    ap.add_argument("--front-end-streams", type=int, default=4096, help="real-time TCP clients of the front-end record (0 = skip)")
    ap.add_argument("--front-end-repeat", type=int, default=3, help="independent runs of the front-end record; the worst (client-side p99) is reported")
    ap.add_argument("--groups", type=int, default=0, help="intra-tick overlap groups (0 = engine default)")
    ap.add_argument("--split-f16", action="store_true",
                    help="opt-in: GEMM-shaped contractions as fp32-accurate 3-term f16 split products (VAPX_FLAG_SPLIT_F16)")
    ap.add_argument("--defer-join", action="store_true", help="with --groups > 1: let overlap groups free-run across ticks")
    ap.add_argument("--engine-flag", action="append", default=[], help="A/B runs: boolean engine.Engine keyword to switch on (e.g. split_qkv_in_ffn); repeatable")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--share-gpu", action="store_true",
                    help="plumbing check on a 1-GPU box: every rank uses device 0 (combine with --backend gloo; RCCL cannot put two ranks on one device)")
    ap.add_argument("--full-record", default="bench_full.json", help="file name of the complete record (written under gpurun_out/, else the repo root)")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="multi-rank plumbing check without a GPU: spawn, rendezvous, shard the streams, barrier, print the ranks")
    args = ap.parse_args()
    ENGINE_KW.update({k: True for k in args.engine_flag})

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # one process per GPU: re-execute under torch.distributed.run exactly as the driver would launch us
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", f"--master-port={free_port()}", os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    import torch
    from vap_realtime_amd import dist_util
    from vap_realtime_amd.sharding import shard_streams
    rank, local_rank, world = dist_util.env_rank()
    dev_index = 0 if args.share_gpu else local_rank
    pinned = dist_util.pin_rank_to_cores(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)), device_index=dev_index, shared=args.share_gpu)

    if args.rendezvous_only:
        dist = dist_util.init("gloo")
        S = WORKLOADS["s4096_20hz"][0]
        mine = shard_streams(S * world, world, rank)
        ends = dist_util.gather_ints(dist, [mine[0], mine[-1], len(mine), os.getpid()])
        dist_util.barrier(dist)
        worst = dist_util.max_over_ranks(dist, 0.001 * (rank + 1))
        seen = dist_util.ranks_seen(dist)
        door = front_door_plumbing_check(world) if rank == 0 else None
        dist_util.barrier(dist)
        if dist is not None:
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"rendezvous_only": True, "n_gpus": world, "shards": [e[:3] for e in ends], "pids": [e[3] for e in ends],
                              "max_over_ranks": worst, "cores_pinned": pinned, "ranks_seen": seen, "front_door": door}))
        return

    assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
    local_rank = dev_index
    torch.cuda.set_device(local_rank)
    dist = dist_util.init(args.backend, torch.device("cuda", local_rank))
    ctx = (rank, local_rank, world, dist)
    side = not args.no_latency
    cpu_cache = {}      # (hz, ctx_sec, mode) -> cpu_baseline record: c2 and s4096_20hz are the same CPU shape

    def full_record(name, S, hz, ctx_sec, mode, steps, warmup, headline=False):
        """One configuration: fp32 record (gate, timing, roofline) + its CPU baseline + split-precision side record + paced latency."""
        rec, wl, twin = run_workload(name, S, hz, ctx_sec, mode, steps, warmup, ctx, groups=args.groups, split_f16=args.split_f16 and headline,
                                     defer_join=args.defer_join and headline)
        extras = {}
        if headline:
            extras = {}
            if rank == 0 and world == 1 and side and not wl.followers:
                extras["lat"] = host_latency(wl)
                extras["groups"] = overlap_groups_record(wl, steps) if args.groups <= 1 and not args.split_f16 else None
        cpc, sets = wl.cpc, wl.sets
        wl.close()
        del wl
        torch.cuda.empty_cache()
        if rank == 0 and not args.no_cpu_baseline:
            key = (hz, ctx_sec, mode)
            if key not in cpu_cache:
                cpu_cache[key] = cpu_baseline_record(twin, (S, hz, ctx_sec, mode), args.cpu_baseline_sec)
            rec["cpu_baseline"] = cpu_cache[key]
        del twin
        if side and not args.split_f16:
            # the same workload on the opt-in split-precision path (VAPX_FLAG_SPLIT_F16): every GEMM-shaped contraction as
            # three f16 MFMA products with fp32 accumulation — same deviation from the reference as the fp32-MFMA path
            # (tests/test_split_precision_gpu.py), its own parity gate.  Reported next to `value`, never as `value`.
            try:
                srec, w2, tw2 = run_workload(name, S, hz, ctx_sec, mode, steps, warmup, ctx, split_f16=True)
                w2.close()
                del w2, tw2
                torch.cuda.empty_cache()
                srec["arithmetic"] = ("x = hi + lo (f16); hi.hi + lo.hi + hi.lo on v_mfma_f32_32x32x16_f16, fp32 accumulate; opt-in "
                                      "(VAPX_FLAG_SPLIT_F16), not the default")
                srec["speedup_over_fp32"] = srec["value"] / rec["value"]
                rec["split_f16"] = srec
            except SystemExit as e:                               # its parity gate failed: report that instead of a number
                rec["split_f16"] = {"error": str(e)}
        if side and args.paced_sec > 0:
            pl = paced_latency(cpc, sets, hz, ctx_sec, local_rank, args.paced_sec)
            rec["paced_latency"] = gather_paced(dist, rank, world, pl)
            rec["concurrent_streams_at_10ms"] = rec["paced_latency"].get("sustained_streams_all_ranks", pl.get("sustained_streams", 0))
            if rec["concurrent_streams_at_10ms"] < 4096 * world and not args.split_f16 and "value" in rec.get("split_f16", {}):
                # the north-star stream count is out of the fp32 matrix cores' reach for this configuration: measure what the
                # split-precision path sustains under the same schedule (side record, like its throughput)
                pls = paced_latency(cpc, sets, hz, ctx_sec, local_rank, args.paced_sec, split_f16=True)
                rec["split_f16"]["paced_latency"] = gather_paced(dist, rank, world, pls)
                rec["split_f16"]["concurrent_streams_at_10ms"] = rec["split_f16"]["paced_latency"].get("sustained_streams_all_ranks", pls.get("sustained_streams", 0))
        return rec, extras

    def host_latency(wl):
        """host-inclusive tick latency: host audio -> results on host (pinned H2D + kernels + D2H + sync)"""
        from vap_realtime_amd import engine
        pin_in = engine.pinned_empty((wl.S, 2, wl.hop))
        pin_out = engine.pinned_empty((wl.S, engine.OUT_STRIDE))
        lat = []
        for i in range(60 if wl.S * wl.T > 100000 else 210):
            pin_in[...] = wl.audio[i % wl.NF]
            t1 = time.perf_counter()
            wl.eng.step(pin_in, out=pin_out)
            lat.append((time.perf_counter() - t1) * 1e3)
        lat = np.array(lat[10:])
        return {"p50": float(np.percentile(lat, 50)), "p99": float(np.percentile(lat, 99)), "max": float(lat.max()),
                "staging": "pinned (vapx_host_alloc)"}

    def overlap_groups_record(wl, steps):
        # the same workload with the tick split into two overlap groups that free-run across ticks (VAPX_DEFER_JOIN):
        # reported next to `value`, not as `value`, because co-running kernels stretch each other's launch time and the
        # per-kernel roofline would stop meaning anything
        from vap_realtime_amd import engine, weights as W
        eng_g = engine.Engine(W.pack_blob(wl.cpc, wl.vap, wl.modes[0]), wl.hz, wl.ctx_sec, max_streams=wl.S, device_id=local_rank,
                              groups=2, mode=wl.modes[0])
        for i in range(wl.T + 5):
            eng_g.step_device(wl.S, wl.d_audio[i % wl.NF].data_ptr(), wl.hop, wl.d_out.data_ptr(), stream=wl.stream, defer_join=True)
        eng_g.join(wl.stream)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(steps):
            eng_g.step_device(wl.S, wl.d_audio[i % wl.NF].data_ptr(), wl.hop, wl.d_out.data_ptr(), stream=wl.stream, defer_join=True)
        eng_g.join(wl.stream)
        torch.cuda.synchronize()
        dtg = time.perf_counter() - t1
        eng_g.close()
        return {"groups": 2, "defer_join": True, "value": wl.S * steps / dtg, "unit": "frames/s", "ms_per_step": dtg / steps * 1e3}

    S, hz, ctx_sec, mode, dsteps, dwarm = WORKLOADS[args.workload]
    S = args.streams or S
    hz = args.frame_hz or hz
    ctx_sec = args.ctx_sec or ctx_sec
    mode = args.mode or mode
    steps = args.steps if args.steps is not None else dsteps
    warmup = args.warmup if args.warmup is not None else dwarm
    head, hx = full_record(args.workload, S, hz, ctx_sec, mode, steps, warmup, headline=True)
    result = {
        "metric": "VAP frames/sec (concurrent 16 kHz stereo streams, one frame per stream per step)",
        "value": head["value"], "unit": "frames/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic (seeded two-speaker dialogue audio, seeded random weights)",
    }
    result.update({k: v for k, v in head.items() if k not in result})
    result["config"] = head["config"]
    if world > 1:
        # proof that the ranks of this record met over the collective backend: sum of one 1 per rank over RCCL ("nccl") on the device
        result["ranks_seen"] = dist_util.ranks_seen(dist, "cuda")
    if args.split_f16:
        result["dtype"] = "f16x3 split products, f32 accumulate (opt-in)"
    # kept from round 1 for continuity: throughput priced with the reference's DENSE FLOP count
    if "dense_tflops" in head:
        result["step_tflops"] = head["dense_tflops"]
        result["step_frac_of_fp32_mfma_peak"] = head["dense_frac_of_fp32_mfma_peak"]
    if hx.get("lat"):
        result["latency_ms_host_inclusive"] = hx["lat"]
    if hx.get("groups"):
        result["overlap_groups"] = hx["groups"]

    # ---- the other single-GPU configurations of BASELINE.json, each a full record ----
    result["configs"] = {}
    overridden = bool(args.streams or args.frame_hz or args.ctx_sec or args.mode)
    for name in [c for c in args.configs.split(",") if c]:
        if name == args.workload and not overridden:
            continue
        cS, chz, cctx, cmode, csteps, cwarm = WORKLOADS[name]
        rec, _ = full_record(name, cS, chz, cctx, cmode, csteps, cwarm)
        result["configs"][name] = rec

    if side and args.front_end_streams > 0:
        # every rank's engines are closed by now; rank 0 serves `front_end_streams` real-time TCP clients — one engine per GPU of the
        # job behind ONE port pair when world > 1 — while the other ranks wait in a CPU (gloo) rendezvous, their GPUs idle
        # (the front door is ONE process over the GPUs of ONE node: the leg needs every rank of the job on this node and a visible device
        # per rank; the waiting ranks release their cached device memory first)
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
        torch.cuda.empty_cache()
        if rank == 0:
            if world != local_world:
                result["front_end"] = {"error": f"skipped: {world} ranks over more than one node (LOCAL_WORLD_SIZE {local_world})"}
            elif not args.share_gpu and torch.cuda.device_count() < world:
                result["front_end"] = {"error": f"skipped: {torch.cuda.device_count()} visible devices for {world} ranks"}
            else:
                devs = [0] * world if args.share_gpu else list(range(world))
                result["front_end"] = front_end_record(args.front_end_streams, 10.0, devices=devs if world > 1 else None, repeat=args.front_end_repeat)
        dist_util.gather_ints(dist, [rank])

    if rank == 0 and not args.no_cpu_baseline and mode == "vap":
        P = args.cpu_procs if args.cpu_procs >= 0 else physical_cores()
        if P > 1:
            # the reference deployed on every core: P independent single-threaded processes (one stream each), P = the
            # host's physical cores; aggregate = sum of the per-process rates over the common window
            import multiprocessing as mp
            try:
                os.sched_setaffinity(0, range(os.cpu_count() or 1))   # the workers must not inherit this rank's core pinning
            except Exception:
                pass
            os.environ["OMP_NUM_THREADS"] = os.environ["MKL_NUM_THREADS"] = "1"    # inherited by the workers: one thread each, really
            with mp.get_context("spawn").Pool(P) as pool:
                res = pool.map(_cpu_worker, [(i, hz, ctx_sec, 6.0) for i in range(P)], chunksize=1)
            agg = sum(n_ / dt_ for n_, dt_ in res)
            result["cpu_baseline_multiprocess"] = {
                "value": agg, "unit": "frames/s", "cores": P, "kind": "port",
                "sample": f"{P} single-threaded oracle processes (= physical cores) x 6 s at {hz} Hz / T={int(ctx_sec * hz)}, one stream each "
                          f"({sum(r[0] for r in res)} frames); host has {physical_cores()} physical / {os.cpu_count()} logical cores", "per_core": agg / P}
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # the complete record goes to a file; stdout carries exactly ONE line: the compact record (< 4 KB, strict JSON)
        result = _jsonable(result)
        full_path = ""
        for d in (os.path.join(ROOT, "gpurun_out"), ROOT, "/tmp"):
            try:
                os.makedirs(d, exist_ok=True)
                fp = os.path.join(d, args.full_record)
                with open(fp, "w") as f:
                    json.dump(result, f, allow_nan=False)
                full_path = os.path.relpath(fp, ROOT) if fp.startswith(ROOT) else fp
                break
            except OSError:
                continue
        sys.stdout.flush()
        print(compact_line(result, full_path), flush=True)


if __name__ == "__main__":
    main()
