"""How much work one stream-frame is, and which arithmetic a GPU needs to serve a given load within the 10 ms bound.

``model_macs`` / ``macs_per_stream_frame`` / ``attention_executed_fraction`` are the EXECUTED-work model of the engine (the exact savings of
DESIGN.md section 4 taken off the reference's dense count); ``bench.py`` prices its roofline with them.  ``plan`` is
what ``python -m vap_realtime_amd.serve --precision auto`` decides with: the measured sustained rate of each arithmetic path
(``bench.py`` records on MI355X, profiles/r05d_bench_driver_cmd_full.json) against the load ``streams x frame_hz x executed GFLOP``."""
from __future__ import annotations

from typing import Dict, Optional


def model_macs(hz: int, T: int, mode: str = "vap", leader: bool = True, qkv_in_attention: bool = False) -> dict:
    """EXECUTED multiply-accumulates per stream-frame (both channels) by kernel class for ONE weight set of the default path.
    vap / bc: exact last-layer pruning, absorbed last-layer K/V projections, cached layer-0 Q|K|V.  nod emits p_bc for every
    window row (vap_nod_main.py:276), so it runs the FULL last layer and the Combinator on all rows.  leader = False: a trunk
    follower (shares the leader's CPC CNN + LSTM, runs only its own downsample).  The attention classes count the DENSE T x T
    products like SURVEY.md does (the kernels skip most of the causally masked tiles, see `attention_executed_fraction`).
    qkv_in_attention (split path, long windows): the self-attention Q|K|V projections of the layers after layer 0 run inside the attention
    kernel (csrc/attention_proj_f16x3.hip) instead of the previous layer's flat-row block — the same MACs, booked under "attention"."""
    hop = 16000 // hz
    L = hop + 320
    P0 = L // 5; P1 = P0 // 4; P2 = P1 // 2; P3 = P2 // 2; P4 = P3 // 2; ncpc = P4 - 2
    D = 256
    rows = 2 * T
    fused = T <= 64          # fused attention block (attention + projection + LN + cross-q) vs attention_long2_kernel + flat-row blocks
    full = mode == "nod"
    n_ffn = 4 if full else 3                       # FFN blocks executed on all rows
    n_next = 3 if full else 2                      # next-layer Q|K|V + cross K|V emitted by an FFN block
    n_attn = 7 if full else 5                      # attention blocks on all rows (l0 self, l1.. self + cross)
    n_proj = n_attn + (3 if full else 2)           # attention output projections + cross-attention query projections
    attn = n_attn * 2 * 4 * (T * T * 64 * 2)
    moved = rows * D * n_next * 768 if (qkv_in_attention and not fused) else 0
    m = {
        "conv0": 2 * P0 * D * 10 if leader else 0,
        "gemm_cn_relu": 2 * (P1 * 8 + P2 * 4 + P3 * 4 + ncpc * 4) * D * D if leader else 0,
        "lstm": (2 * ncpc * D * 4 * D + 2 * ncpc * D * D) if leader else 0,    # recurrence (K=256) + fused downsample
        # follower: its own downsample GEMM; nod: Combinator on all rows (two [T x 256 x 256] per stream)
        "gemm_bias_ln_gelu": (0 if leader else 2 * ncpc * D * D) + (2 * T * D * D if full else 0),
        # LSTM input projection (leader) + layer-0 QKV of the NEW row (others cached)
        "gemm_store": (2 * ncpc * D * 4 * D if leader else 0) + 2 * D * 768,
        "gemm_resid_ln": 0,
        # FFN + next layer's QKV / cross-KV (last pruned layer: absorbed); long windows: + the attention output projections and
        # the cross-attention query projections, which ride in the same flat-row blocks (fused_blocks.hip, modes 1 / 2)
        # (mode 1: one pre-projection per FFN block).  The mode-2 launches — attention output projection + LN_src + cross-attention query,
        # two per stereo layer executed on all rows — are a class of their own ("ffn_proj": csrc/engine.hip CLS_FFN_PROJ)
        "ffn_block": rows * D * (n_ffn * 2 * 768 + n_next * (768 + 512) + (0 if fused else n_ffn * D)) - moved,
        "ffn_proj": 0 if fused else rows * D * (n_proj - n_ffn) * D,
        # pruned layer 3 on one row per channel: 14 contractions (q, Wk^T q, Wv, proj, their cross twins, FFN) + two
        # 4-head single-query attentions over T rows of 256 (score + weighted sum)
        "last_row": 0 if full else 2 * (14 * D * D + 2 * 4 * T * D * 2),
        "gemm_gelu": 0, "gemm_resid": 0,
        # dense T x T attention (+ in the fused block: output projections, cross-q projections)
        "attention": attn + (rows * n_proj * D * D if fused else 0) + moved,
        "head": 3 * D * D + 2 * D,
        "gather_ln": 0,
    }
    return m


def macs_per_stream_frame(hz: int, T: int, mode: str = "vap", qkv_in_attention: bool = False) -> dict:
    """Sum of `model_macs` over the weight sets of `mode` ("bc+nod": the first leads the shared CPC trunk)."""
    tot = {}
    for k, md in enumerate(mode.split("+")):
        for c, v in model_macs(hz, T, md, leader=(k == 0), qkv_in_attention=qkv_in_attention).items():
            tot[c] = tot.get(c, 0) + v
    return tot


def attention_executed_fraction(T: int) -> float:
    """Share of the dense T x T score / PV products the attention kernels execute: 32-row tiles, causal tiles jt <= it only."""
    nt = (T + 31) // 32
    if T <= 64:
        return 1.0 if nt == 1 else 0.75       # fused block: 64 x 64 scores, tile (0,1) skipped
    return (nt * (nt + 1) / 2) / (nt * nt)


def executed_gflop_per_stream_frame(hz: int, T: int, mode: str = "vap", qkv_in_attention: bool = False) -> float:
    """Executed GFLOP of one stream-frame (all weight sets of ``mode``), causal attention tiles only — ``bench.py``'s ``executed_tflops`` unit."""
    macs = macs_per_stream_frame(hz, T, mode, qkv_in_attention=qkv_in_attention)
    n_attn = sum(7 if m == "nod" else 5 for m in mode.split("+"))
    dense_attn = n_attn * 2 * 4 * (T * T * 64 * 2)
    return 2.0 * (sum(macs.values()) - (1.0 - attention_executed_fraction(T)) * dense_attn) / 1e9


# Sustained executed TFLOP/s of a whole tick at thousands of streams per GPU (bench.py on MI355X, driver's command on the round-5 tree:
# profiles/r05d_bench_driver_cmd_full.json — C3 44.3 k frames/s x 2.763 GFLOP, 4096 x 20 Hz 177.1 k x 0.671, C5 87.0 k x 1.37; split 97.1 k /
# 361.8 k / 176.4 k).  Short windows (T <= 64, fused attention block) and long windows run different kernel chains, hence two columns.
MEASURED_TFLOPS = {"fp32": {"short": 118.8, "long": 122.4}, "split": {"short": 242.8, "long": 268.3}}
MAX_BUSY = 0.85          # a GPU that is more than 85 % busy has no headroom for the ragged ticks of real arrivals: the paced-latency legs of
                         # bench.py accept a point at p99 <= 9 ms AND <= 85 % busy ("The <= 10 ms target")


def sustained_frames_per_s(hz: int, T: int, mode: str, precision: str) -> float:
    g = executed_gflop_per_stream_frame(hz, T, mode, qkv_in_attention=(precision == "split" and T > 64))
    return MEASURED_TFLOPS[precision]["short" if T <= 64 else "long"] * 1e3 / g


def plan(streams: int, hz: int, ctx_sec: float, mode: str = "vap", dedicated: Optional[bool] = None) -> Dict:
    """Which arithmetic serves ``streams`` dialogues per GPU of the ``hz`` / ``ctx_sec`` model ``mode`` within 10 ms per frame.

    busy(path) = streams x hz / sustained frames/s of the path.  ``fp32`` when it stays <= 85 % busy (the reference's arithmetic, safe next to
    other tenants); else ``split`` when THAT stays <= 85 % (fp32-accurate 3-term f16 products: same <= 1e-4 parity, ~2.1 x the rate; for a
    dedicated GPU — DESIGN.md "co-running f16 MFMA"); else neither holds the bound: ``ok`` is False, ``precision`` is the faster path and
    ``max_streams`` says what one GPU of each path does hold."""
    T = int(ctx_sec * hz)
    out = {"streams": streams, "frame_hz": hz, "ctx_frames": T, "mode": mode, "max_busy": MAX_BUSY}
    for p in ("fp32", "split"):
        rate = sustained_frames_per_s(hz, T, mode, p)
        out[p] = {"frames_per_s": rate, "busy": streams * hz / rate, "max_streams": int(MAX_BUSY * rate / hz)}
    if out["fp32"]["busy"] <= MAX_BUSY:
        out.update(precision="fp32", ok=True, reason=f"fp32 path {100 * out['fp32']['busy']:.0f} % busy")
    elif out["split"]["busy"] <= MAX_BUSY:
        out.update(precision="split", ok=True,
                   reason=f"fp32 path would be {100 * out['fp32']['busy']:.0f} % busy (> {100 * MAX_BUSY:.0f} %: holds {out['fp32']['max_streams']} streams within "
                          f"10 ms); split-precision path {100 * out['split']['busy']:.0f} % busy")
    else:
        out.update(precision="split", ok=False,
                   reason=f"NEITHER path holds <= 10 ms per frame at {streams} streams per GPU: fp32 {100 * out['fp32']['busy']:.0f} % busy (holds "
                          f"{out['fp32']['max_streams']}), split {100 * out['split']['busy']:.0f} % busy (holds {out['split']['max_streams']}); "
                          f"spread the dialogues over more GPUs (--gpus) or lower --streams")
    return out
