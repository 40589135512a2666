"""Parity at production batch sizes (BASELINE.json configs 2, 3 and 5 at their real stream counts).

The goldens of the imported reference hold 1-3 streams.  Streams are independent (process_vap has no cross-stream
term), so a full-size batch whose slots are replicas of the golden streams must reproduce the golden in EVERY
replica: the golden audio is tiled over 1100 / 4096 stream slots under a shuffled slot assignment and every row of
every frame is compared with the reference's numbers (<= 1e-4 abs, north_star).  This runs the code that only
large batches reach under a checker: head_kernel<4> (> 1024 streams), the three-GEMM conv tail (> 512 streams),
the large-M GEMM tile choices, and scratch offsets beyond 4 GB.
"""
import numpy as np
import pytest

from golden_util import Case

pytestmark = pytest.mark.gpu

TOL = 1e-4


def _tiled_run(name, S, keys, extra=None, max_frames=None, seed=0, **engine_kw):
    from vap_realtime_amd import engine, weights as W
    c = Case(name)
    ns = len(c.streams)
    blob = W.pack_blob(c.cpc_sd, c.vap_sd, c.mode)
    eng = engine.Engine(blob, c.frame_hz, c.ctx_sec, max_streams=S + 7, max_batch=S, mode=c.mode, **engine_kw)
    rng = np.random.default_rng(seed + S)
    ids = rng.permutation(S + 7)[:S].astype(np.int32)            # shuffled slots, a few left unused
    src = (np.arange(S) % ns).astype(np.int64)                   # batch row k replays golden stream src[k]
    worst = {}
    spread = 0.0
    F_ = c.n_frames if max_frames is None else min(max_frames, c.n_frames)
    for f in range(F_):
        audio = np.ascontiguousarray(c.new_samples(f)[src])      # [S,2,hop]
        o = engine.split_outputs(eng.step(audio, ids))
        assert np.all(o["n"] == min(f + 1, c.T))
        for k in keys:
            want = c.z[k][f][src]
            got = o[k].reshape(want.shape)
            worst[k] = max(worst.get(k, 0.0), float(np.abs(got - want).max()))
        if extra is not None:
            extra(c, f, o, src, worst)
        probe = "p_now" if "p_now" in keys else "aux"
        spread = max(spread, float(np.abs(o[probe] - o[probe][:ns][src]).max()))
    eng.close()
    print(f"{name} x {S} slots: worst |hip - reference golden| = {worst}; replica spread {spread:.2e}")
    for k, v in worst.items():
        assert v <= TOL, (name, S, k, v)


@pytest.mark.parametrize("S,split", [(1100, False), (4096, False), (4096, True)], ids=["1100", "4096", "4096-split_f16"])
def test_multi3_tiled_over_full_size_batches(S, split):
    """C2 / C4 shape (20 Hz, T = 50) at 1100 and 4096 streams per engine (4096 also on the split-precision path)."""
    _tiled_run("multi3", S, ("p_now", "p_future", "vad", "logits"), split_f16=split)


def test_vap50_tiled_over_1100_slots():
    """C3 shape (50 Hz, T = 250: attention_long2_kernel + ffn_block_kernel modes 1 / 2, the long-window chain) at 1100 streams,
    all 256 frames (window fills and slides)."""
    _tiled_run("vap50", 1100, ("p_now", "p_future", "vad", "logits"))


def test_vap50_tiled_over_4096_slots():
    """C3 at its full size: 4096 streams x T = 250 (2 M transformer rows; scratch buffers of 2-6 GB each).  The first
    40 frames (the window is still filling: every frame has a different n) keep the run short."""
    _tiled_run("vap50", 4096, ("p_now", "p_future", "vad", "logits"), max_frames=40)


@pytest.mark.parametrize("split", [False, True], ids=["fp32", "split_f16"])
def test_vap50_full_window_at_4096_slots(split):
    """C3 at its full size WITH A FULL WINDOW (4096 streams x T = 250 = 2 M transformer rows per layer, scratch buffers of 2.1 GB each):
    what bench.py times, held against the reference's golden in every one of the 4096 slots.  Replaying 250 warm-up ticks at 4096 streams
    would cost minutes, so a 1-stream engine replays the golden to frame 245 (checked on the way), its state (ring, LSTM, carry:
    vapx_get_state) is imported into 4096 shuffled slots of the big engine (vapx_set_state rebuilds the layer-0 Q|K|V cache), and the big
    engine then steps frames 246-255: the window grows 247 -> 250 rows, is full at frame 249 and slides six times (vap_main.py:274-283)."""
    from vap_realtime_amd import engine, weights as W
    c = Case("vap50")
    S, F0 = 4096, 246
    keys = ("p_now", "p_future", "vad", "logits")
    ns = len(c.streams)
    blob = W.pack_blob(c.cpc_sd, c.vap_sd, c.mode)
    small = engine.Engine(blob, c.frame_hz, c.ctx_sec, max_streams=ns, mode=c.mode, split_f16=split)
    for f in range(F0):
        o = engine.split_outputs(small.step(np.ascontiguousarray(c.new_samples(f))))
        for k in keys:
            assert float(np.abs(o[k].reshape(c.z[k][f].shape) - c.z[k][f]).max()) <= TOL, (f, k)
    states = [small.get_state(i) for i in range(ns)]
    small.close()
    assert all(st["n_frames"] == F0 for st in states)
    big = engine.Engine(blob, c.frame_hz, c.ctx_sec, max_streams=S + 7, max_batch=S, mode=c.mode, split_f16=split)
    ids = np.random.default_rng(5).permutation(S + 7)[:S].astype(np.int32)
    src = (np.arange(S) % ns).astype(np.int64)
    for k in range(S):
        big.set_state(int(ids[k]), states[src[k]])
    worst = {}
    for f in range(F0, c.n_frames):
        o = engine.split_outputs(big.step(np.ascontiguousarray(c.new_samples(f)[src]), ids))
        assert np.all(o["n"] == min(f + 1, c.T)) and not o["status"].any()
        for k in keys:
            want = c.z[k][f][src]
            worst[k] = max(worst.get(k, 0.0), float(np.abs(o[k].reshape(want.shape) - want).max()))
    big.close()
    print(f"vap50 full window x {S} slots ({'split' if split else 'fp32'}): worst |hip - reference golden| = {worst}")
    assert c.n_frames - 1 >= c.T + 5                               # the run really slid the full window
    for k, v in worst.items():
        assert v <= TOL, (k, v)


@pytest.mark.parametrize("split", [False, True], ids=["fp32", "split_f16"])
def test_bc_and_nod_on_one_shared_trunk_at_4096_streams(split):
    """C5 EXACTLY as bench.py runs it: the bc engine leads the CPC trunk, the nod engine follows (vapx_attach_trunk: one CNN + LSTM pass
    per tick for both weight sets), 4096 shuffled slots, against the goldens of the reference's two programs run on ONE cpc_model file
    and the same audio (trunk_bc20 / trunk_nod20; vap_bc_main.py:272-277, vap_nod_main.py:273-279)."""
    from vap_realtime_amd import engine, weights as W
    cb, cn = Case("trunk_bc20"), Case("trunk_nod20")
    S = 4096
    ns = len(cb.streams)
    blobs = {"bc": W.pack_blob(cb.cpc_sd, cb.vap_sd, "bc"), "nod": W.pack_blob(cn.cpc_sd, cn.vap_sd, "nod")}
    grp = engine.TrunkGroup(blobs, cb.frame_hz, cb.ctx_sec, max_streams=S + 7, max_batch=S, split_f16=split)
    ids = np.random.default_rng(11).permutation(S + 7)[:S].astype(np.int32)
    src = (np.arange(S) % ns).astype(np.int64)
    worst = {}
    for f in range(cb.n_frames):
        res = grp.step(np.ascontiguousarray(cb.new_samples(f)[src]), ids)
        ob, on = engine.split_outputs(res["bc"]), engine.split_outputs(res["nod"])
        assert not ob["status"].any() and not on["status"].any()
        _bc_extra(cb, f, ob, src, worst)
        _nod_extra(cn, f, on, src, worst)
        if f % cb.z["meta.e_stride"] == 0:                      # the shared encoder's embedding, as both reference programs computed it
            want = cb.z["e"][f // int(cb.z["meta.e_stride"])][src]
            worst["e"] = max(worst.get("e", 0.0), float(np.abs(ob["e"] - want).max()))
    grp.close()
    print(f"bc + nod on one trunk x {S} slots ({'split' if split else 'fp32'}): worst |hip - reference golden| = {worst}")
    for k, v in worst.items():
        assert v <= TOL, (k, v)


@pytest.mark.parametrize("name,S", [("vap50", 77), ("vap20_10s", 301)])
def test_long_windows_tiled_on_the_split_precision_path(name, S):
    """The split-precision long-window chain (VAPX_FLAG_SPLIT_F16: attention_long_f16x3_kernel is PERSISTENT — one workgroup per CU walks
    over (stream, channel, head) items with the next item's rows prefetched — and ffn_block_f16x3 modes 1 / 2) at batch sizes the 1-stream
    goldens never reach: 77 x 8 = 616 and 301 x 8 = 2408 items on 256 workgroups (several items per workgroup, a ragged last round),
    shuffled slots (ring addressing through `ids`), every frame from the empty window to the sliding one, T = 250 and T = 200."""
    _tiled_run(name, S, ("p_now", "p_future", "vad", "logits"), split_f16=True)


def _bc_extra(c, f, o, src, worst):
    for k, col in (("p_bc_react", 1), ("p_bc_emo", 2)):
        want = c.z[k][f].reshape(-1)[src]
        worst[k] = max(worst.get(k, 0.0), float(np.abs(o["aux"][:, col] - want).max()))


def _nod_extra(c, f, o, src, worst):
    for k, col in (("p_nod_short", 1), ("p_nod_long", 2), ("p_nod_long_p", 3)):
        want = c.z[k][f].reshape(-1)[src]
        worst[k] = max(worst.get(k, 0.0), float(np.abs(o["aux"][:, col] - want).max()))
    n = min(f + 1, c.T)                                          # p_bc of every window row (reference quirk)
    want = c.z["p_bc"][f][src][:, :n]
    worst["p_bc"] = max(worst.get("p_bc", 0.0), float(np.abs(o["logits"][:, :n] - want).max()))


@pytest.mark.parametrize("split", [False, True], ids=["fp32", "split_f16"])
def test_bc_and_nod_heads_at_4096_streams(split):
    """C5 shape: the bc and nod variants at 4096 streams (nod runs the full last layer + the all-rows p_bc)."""
    _tiled_run("bc20", 4096, ("e",), extra=_bc_extra, split_f16=split)       # (the bc / nod programs never fill result_vad: no golden for it)
    _tiled_run("nod20", 4096, ("e",), extra=_nod_extra, split_f16=split)
