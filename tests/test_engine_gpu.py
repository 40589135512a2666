"""Parity of the HIP path (through the C ABI) against the golden vectors of the imported reference
and against the oracle, on the GPU.  Tolerance: 1e-4 abs on p_now / p_future / VAD / logits
(BASELINE.json north_star)."""
import numpy as np
import pytest

from golden_util import Case

pytestmark = pytest.mark.gpu

TOL = 1e-4


def make_engine(case, max_streams=None, **kw):
    from vap_realtime_amd import engine, weights as W
    blob = W.pack_blob(case.cpc_sd, case.vap_sd, case.mode)
    S = max_streams or len(case.streams)
    return engine.Engine(blob, case.frame_hz, case.ctx_sec, max_streams=S, mode=case.mode, **kw)


def run_engine(case, eng, ids=None):
    from vap_realtime_amd.engine import split_outputs
    outs = []
    for f in range(case.n_frames):
        audio = case.new_samples(f) if case.framing == "server" else case.window(f)
        outs.append(split_outputs(eng.step(audio, ids)))
    return outs


@pytest.mark.parametrize("name", ["vap20", "offline20", "vap10", "multi3", "vap50", "degenerate20", "vap20_10s"])
def test_step_matches_reference_golden(name):
    """degenerate20 = silence / near-denormal / clipping / 3e3 x / DC offset / dead channel as six independent runs of the unmodified
    reference (where ChannelNorm divides by ~0); vap20_10s = T = 200, the longest published window.  Same 1e-4 as everything else."""
    c = Case(name)
    eng = make_engine(c)
    outs = run_engine(c, eng)
    z = c.z
    es = int(z["meta.e_stride"]) if "meta.e_stride" in z.files else 1
    worst = {}
    for f, o in enumerate(outs):
        assert np.all(o["n"] == min(f + 1, c.T))
        for k in ("p_now", "p_future", "vad", "logits"):
            d = float(np.abs(o[k] - z[k][f]).max())
            worst[k] = max(worst.get(k, 0.0), d)
        if f % es == 0:
            worst["e"] = max(worst.get("e", 0.0), float(np.abs(o["e"] - z["e"][f // es]).max()))
    print(name, worst)
    for k, v in worst.items():
        assert v <= TOL, (name, k, v)
    eng.close()


def test_intermediates_match_reference():
    """Per-stage buffers of selected frames (CNN, LSTM, each transformer layer) vs hooks on the
    reference modules."""
    c = Case("vap20")
    eng = make_engine(c, full_last_layer=True)
    z = c.z
    inter_frames = sorted({int(k.split(".")[1][1:]) for k in z.files if k.startswith("inter.f")})
    for f in range(max(inter_frames) + 1):
        eng.step(c.new_samples(f))
        if f not in inter_frames:
            continue
        ncpc = 5
        zbuf = eng.peek("z", (2, ncpc, 256))
        cnn4 = z[f"inter.f{f}.cnn4"]                       # [256, 7] channel-1
        np.testing.assert_allclose(zbuf[0], cnn4[:, 1:-1].T, rtol=0, atol=2e-5)
        lo = eng.peek("lstm_out", (2, ncpc, 256))
        np.testing.assert_allclose(lo, z[f"inter.f{f}.lstm_out"], rtol=0, atol=2e-5)
        rows = z[f"inter.f{f}.rows"]
        for key, buf in (("o", "o"), ("stereo0", "stereo0"), ("stereo1", "stereo1"), ("stereo2", "stereo2")):
            got = eng.peek(buf, (2, c.T, 256))[:, rows]
            want = z[f"inter.f{f}.{key}"]
            np.testing.assert_allclose(got, want, rtol=3e-5, atol=1e-3, err_msg=f"frame {f} {key}")
    eng.close()


def test_fused_conv_tail_equals_three_gemms():
    """conv2-4 fused in LDS (default up to 512 streams) vs the three implicit-GEMM launches; h2/h3 are
    only peekable on the unfused path."""
    from vap_realtime_amd import engine
    c = Case("vap20")
    a, b = make_engine(c), make_engine(c, unfused_conv=True)
    for f in range(3):
        oa = engine.split_outputs(a.step(c.new_samples(f)))
        ob = engine.split_outputs(b.step(c.new_samples(f)))
        np.testing.assert_allclose(oa["e"], ob["e"], rtol=0, atol=5e-6)
        np.testing.assert_allclose(a.peek("z", (2, 5, 256)), b.peek("z", (2, 5, 256)), rtol=0, atol=5e-6)
    assert b.peek("h3", (2, 16, 256)).shape == (2, 16, 256)
    with pytest.raises(engine.VapxError, match="fused conv tail"):
        a.peek("h2", (2, 30, 256))
    with pytest.raises(engine.VapxError, match="FULL_LAST_LAYER"):
        a.peek("stereo2", (2, 50, 256))
    a.close(); b.close()


def test_pruned_last_layer_equals_full_last_layer():
    """Default path computes only the newest row of the last layer; it must agree with the
    full-layer path (VAPX_FLAG_FULL_LAST_LAYER) far below the parity tolerance."""
    c = Case("multi3")
    a, b = make_engine(c), make_engine(c, full_last_layer=True)
    for oa, ob in zip(run_engine(c, a), run_engine(c, b)):
        for k in ("p_now", "p_future", "vad", "logits"):
            np.testing.assert_allclose(oa[k], ob[k], rtol=0, atol=2e-5)
    a.close(); b.close()


def test_engine_equals_oracle_on_fresh_inputs():
    """Seeded inputs that are NOT in the golden set: HIP path vs oracle, 3 streams, ragged ids."""
    from oracle.vap_oracle import ServerFramer, VapOracle
    from vap_realtime_amd import engine, synth, weights as W
    cpc, vap = W.synthetic_weights(21, 20, "vap")
    o = VapOracle(cpc, vap, 20, 2.5)
    S, F_ = 3, 54
    audio = synth.dialogue_batch([40, 41, 42], 800 * F_)
    st, fr = o.new_state(S), ServerFramer(S, 800)
    eng = engine.Engine(W.pack_blob(cpc, vap), 20, 2.5, max_streams=8, max_batch=4)
    ids = [5, 0, 7]
    for f in range(F_):
        new = audio[:, :, f * 800:(f + 1) * 800]
        want = o.step(fr.frame(new), st)
        got = engine.split_outputs(eng.step(new, ids))
        for k in ("p_now", "p_future", "vad", "logits"):
            np.testing.assert_allclose(got[k], want[k], rtol=0, atol=TOL, err_msg=f"frame {f} {k}")
    eng.close()


@pytest.mark.parametrize("name", ["bc20", "nod20", "nod20_10s"])
def test_aux_heads(name):
    """nod20_10s: the published nod setting with the longest window (20 Hz x 10 s, T = 200, README.md:381)."""
    c = Case(name)
    eng = make_engine(c)
    outs = run_engine(c, eng)
    for f, o in enumerate(outs):
        if name == "bc20":
            np.testing.assert_allclose(o["aux"][:, 1], c.z["p_bc_react"][f].reshape(-1), rtol=0, atol=TOL)
            np.testing.assert_allclose(o["aux"][:, 2], c.z["p_bc_emo"][f].reshape(-1), rtol=0, atol=TOL)
        else:
            np.testing.assert_allclose(o["aux"][:, 1], c.z["p_nod_short"][f].reshape(-1), rtol=0, atol=TOL)
            np.testing.assert_allclose(o["aux"][:, 2], c.z["p_nod_long"][f].reshape(-1), rtol=0, atol=TOL)
            np.testing.assert_allclose(o["aux"][:, 3], c.z["p_nod_long_p"][f].reshape(-1), rtol=0, atol=TOL)
            n = min(f + 1, c.T)        # p_bc for every row of the window (batch-index quirk of the reference)
            np.testing.assert_allclose(o["logits"][:, :n], c.z["p_bc"][f][:, :n], rtol=0, atol=TOL)
    eng.close()


def test_state_roundtrip_and_reset():
    """get_state -> set_state into another slot reproduces the stream; reset == fresh stream."""
    from vap_realtime_amd.engine import split_outputs
    c = Case("vap20")
    eng = make_engine(c, max_streams=3)
    for f in range(53):
        eng.step(c.new_samples(f), [0])
    st = eng.get_state(0)
    assert st["n_frames"] == c.T
    eng.set_state(2, st)
    a = split_outputs(eng.step(c.new_samples(53), [0]))
    b = split_outputs(eng.step(c.new_samples(53), [2]))
    np.testing.assert_allclose(a["logits"], b["logits"], rtol=0, atol=1e-5)
    eng.reset_stream(0)
    fresh = split_outputs(eng.step(c.new_samples(0), [0]))
    np.testing.assert_allclose(fresh["logits"], c.z["logits"][0], rtol=0, atol=TOL)
    eng.close()


def test_5hz_k20_against_oracle():
    """5 Hz frames: L = 3520, 20 CPC frames per VAP frame (downsample K = 20), T = 15 (3 s)."""
    from oracle.vap_oracle import ServerFramer, VapOracle
    from vap_realtime_amd import engine, synth, weights as W
    cpc, vap = W.synthetic_weights(31, 5, "vap")
    o = VapOracle(cpc, vap, 5, 3.0)
    S, F_, hop = 2, 18, 3200
    audio = synth.dialogue_batch([60, 61], hop * F_)
    st, fr = o.new_state(S), ServerFramer(S, hop)
    eng = engine.Engine(W.pack_blob(cpc, vap), 5, 3.0, max_streams=S)
    assert eng.T == 15
    for f in range(F_):
        new = audio[:, :, f * hop:(f + 1) * hop]
        want = o.step(fr.frame(new), st)
        got = engine.split_outputs(eng.step(new))
        for k in ("p_now", "p_future", "vad", "logits", "e"):
            np.testing.assert_allclose(got[k], want[k], rtol=0, atol=TOL, err_msg=f"frame {f} {k}")
    eng.close()


def test_device_path_equals_host_path():
    """vapx_step with device audio / device output / device stream ids on a non-default HIP stream
    (what bench.py times) gives the same numbers as the host path."""
    import torch
    from vap_realtime_amd import engine
    c = Case("multi3")
    a, b = make_engine(c, max_streams=8), make_engine(c, max_streams=8)
    ids = [6, 1, 3]
    d_ids = torch.tensor(ids, dtype=torch.int32, device="cuda")
    d_out = torch.zeros(3, engine.OUT_STRIDE, device="cuda")
    side = torch.cuda.Stream()
    for f in range(8):
        new = c.new_samples(f)
        want = a.step(new, ids)
        d_audio = torch.from_numpy(np.ascontiguousarray(new)).cuda()
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            b.step_device(3, d_audio.data_ptr(), c.hop, d_out.data_ptr(), ids_ptr=d_ids.data_ptr(), stream=side.cuda_stream)
        side.synchronize()
        np.testing.assert_array_equal(d_out.cpu().numpy(), want)
    a.close(); b.close()


@pytest.mark.parametrize("hz,ctx,S", [(20, 2.5, 21), (50, 5.0, 3)])
def test_fused_last_row_block_equals_the_ten_launch_path(hz, ctx, S):
    """last_block_kernel (whole last layer on the newest row, one launch) vs the unfused gathers / GEMMs / single-query
    attentions; windows still filling, an odd batch (partial 16-row tile) and T = 250 (keys beyond one wave)."""
    from vap_realtime_amd import engine, synth, weights as W
    cpc, vap = W.synthetic_weights(21, hz, "vap")
    blob = W.pack_blob(cpc, vap)
    hop = 16000 // hz
    F_ = 7 if hz == 20 else 70
    audio = synth.noise_batch(S, hop * F_, seed=9) * np.linspace(0.3, 2.0, S, dtype=np.float32)[:, None, None]
    fused = engine.Engine(blob, hz, ctx, max_streams=S)
    plain = engine.Engine(blob, hz, ctx, max_streams=S, unfused_last_row=True)
    worst = 0.0
    for f in range(F_):
        a = audio[:, :, f * hop:(f + 1) * hop]
        got, want = fused.step(a), plain.step(a)
        worst = max(worst, float(np.abs(got[:, :272] - want[:, :272]).max()))
    print("fused vs unfused last row: worst |diff| =", worst)
    assert worst <= 2e-5
    fused.close(); plain.close()


@pytest.mark.parametrize("defer", [False, True])
def test_overlap_groups_equal_single_stream_path(defer):
    """Intra-tick overlap groups (vapx_config.flags bits 0-3) split the batch over HIP streams; with VAPX_DEFER_JOIN the
    groups free-run across ticks and vapx_join orders the consumer.  Outputs must be bit-identical to groups = 1."""
    import torch
    from vap_realtime_amd import engine, synth, weights as W
    cpc, vap = W.synthetic_weights(5, 20, "vap")
    blob = W.pack_blob(cpc, vap)
    S, F_ = 96, 6
    audio = synth.noise_batch(S, 800 * F_, seed=3)
    one = engine.Engine(blob, 20, 2.5, max_streams=S)
    grp = engine.Engine(blob, 20, 2.5, max_streams=S, groups=3)
    d_out = [torch.zeros(S, engine.OUT_STRIDE, device="cuda") for _ in range(F_)]
    d_audio = [torch.from_numpy(np.ascontiguousarray(audio[:, :, f * 800:(f + 1) * 800])).cuda() for f in range(F_)]
    want = [one.step(audio[:, :, f * 800:(f + 1) * 800]) for f in range(F_)]
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    for f in range(F_):          # all ticks enqueued back to back, no host sync in between
        grp.step_device(S, d_audio[f].data_ptr(), 800, d_out[f].data_ptr(), stream=side.cuda_stream, defer_join=defer)
    grp.join(side.cuda_stream)
    side.synchronize()
    for f in range(F_):
        np.testing.assert_array_equal(d_out[f].cpu().numpy(), want[f])
    one.close(); grp.close()


def test_error_paths():
    from vap_realtime_amd import engine
    c = Case("vap20")
    eng = make_engine(c, max_streams=2)
    with pytest.raises(engine.VapxError, match="samples_per_ch"):
        eng.step(np.zeros((1, 2, 123), np.float32))
    with pytest.raises(engine.VapxError, match="out of range"):
        eng.step(np.zeros((1, 2, 800), np.float32), [5])
    with pytest.raises(engine.VapxError):
        eng.step(np.zeros((3, 2, 800), np.float32))
    with pytest.raises(engine.VapxError):
        eng.reset_stream(2)
    with pytest.raises(engine.VapxError, match="appears twice"):
        eng.step(np.zeros((2, 2, 800), np.float32), [1, 1])
    eng.close()


def test_non_finite_state_fails_loudly_and_reset_recovers():
    """The host output path refuses to hand NaNs to the caller (VAPX_E_NUMERIC) — per stream, and a reset recovers the stream.  A NaN
    sample poisons a stream exactly as it poisons the reference (next test)."""
    from vap_realtime_amd import engine
    c = Case("vap20")
    eng = make_engine(c, max_streams=2)
    good = np.concatenate([c.new_samples(0), c.new_samples(1)], axis=0)          # two streams
    eng.step(good)
    bad = good.copy()
    bad[1, 0, 17] = np.nan
    with pytest.raises(engine.VapxError, match="non-finite outputs for batch slot 1"):
        eng.step(bad)
    assert eng.bad_slots() == [1]
    with pytest.raises(engine.VapxError, match="non-finite outputs for batch slot 1"):      # the LSTM state keeps it (as the reference's does)
        eng.step(good)
    # per stream, not per call: the block is complete, the healthy row is valid and flagged ok
    ref = make_engine(c, max_streams=2)
    ref.step(good); ref.step(good); ref.step(good)
    want = ref.step(good)
    got = eng.step(good, on_numeric="status")
    assert got[:, engine.OUT_STATUS].tolist() == [0.0, 1.0] and eng.bad_slots() == [1]
    np.testing.assert_array_equal(got[0], want[0])
    eng.reset_stream(1)
    out = eng.step(good)
    assert np.isfinite(out[:, :10]).all() and not out[:, engine.OUT_STATUS].any() and eng.bad_slots() == []
    eng.close(); ref.close()


def test_poisoned_sample_behaves_like_the_reference():
    """poison20 golden: the unmodified reference fed ONE NaN (stream 1) / ONE Inf (stream 2) sample in frame 3.  Its outputs turn NaN
    from that frame on and stay NaN (ChannelNorm + torch.relu propagate, the LSTM state keeps it) except the VAD of the other
    channel.  The HIP path must be non-finite at exactly the same positions, equal within 1e-4 everywhere else, flag exactly those
    streams (VAPX_OUT_STATUS) and leave the clean stream of the same batch untouched."""
    from vap_realtime_amd import engine
    c = Case("poison20")
    eng = make_engine(c)
    z = c.z
    worst = 0.0
    for f in range(c.n_frames):
        o = engine.split_outputs(eng.step(c.new_samples(f), on_numeric="status"))
        for k in ("p_now", "p_future", "vad", "logits"):
            want = z[k][f]
            assert np.array_equal(np.isnan(want), ~np.isfinite(o[k])), (f, k)
            fin = np.isfinite(want)
            worst = max(worst, float(np.abs(o[k][fin] - want[fin]).max()))
        assert o["status"].tolist() == [0, int(f >= 3), int(f >= 3)], (f, o["status"])
        assert eng.bad_slots() == ([1, 2] if f >= 3 else [])
    print("poison20: worst finite deviation", worst)
    assert worst <= TOL
    eng.close()


@pytest.mark.parametrize("split", [False, True], ids=["fp32", "split_f16"])
def test_poisoned_sample_in_a_long_window_stays_in_its_stream(split):
    """The same contract at 50 Hz / 5 s (long-window chain: flat-row blocks over rows of SEVERAL streams, attention items of several
    streams per workgroup on the split path) with per-tile / per-item operand scales taken from maxima: one NaN sample in stream 1 must
    flag stream 1 only and leave streams 0 and 2 BIT-identical to a run that never saw it; a reset brings stream 1 back."""
    from vap_realtime_amd import engine, synth, weights as W
    cpc, vap = W.synthetic_weights(23, 50, "vap")
    blob = W.pack_blob(cpc, vap)
    S, hop, F_ = 3, 320, 12
    audio = synth.dialogue_batch([90, 91, 92], hop * F_)
    bad = audio.copy()
    bad[1, 0, 3 * hop + 17] = np.nan
    clean_eng = engine.Engine(blob, 50, 5.0, max_streams=S, split_f16=split)
    eng = engine.Engine(blob, 50, 5.0, max_streams=S, split_f16=split)
    for f in range(F_):
        sl = slice(f * hop, (f + 1) * hop)
        want = clean_eng.step(np.ascontiguousarray(audio[:, :, sl]))
        if f == 8:
            eng.reset_stream(1)
        src = bad if f < 8 else audio
        got = eng.step(np.ascontiguousarray(src[:, :, sl]), on_numeric="status")
        o = engine.split_outputs(got)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[2], want[2]), f      # the neighbours never notice
        poisoned = 3 <= f < 8
        assert o["status"].tolist() == [0, int(poisoned), 0], (f, o["status"])
        assert np.isfinite(o["logits"][1]).all() != poisoned
    clean_eng.close(); eng.close()


def test_reset_is_stream_ordered_and_leaves_the_other_streams_alone():
    """vapx_reset_stream only queues the request; the next step applies it on its own HIP stream.  Resetting one stream of
    a 1024-stream engine mid-run: every other stream's outputs are bit-identical to an engine that saw no reset, the reset
    stream restarts from a fresh context, and the reset call itself costs microseconds (no device synchronisation)."""
    import time
    import torch
    from vap_realtime_amd import engine, synth, weights as W
    cpc, vap = W.synthetic_weights(5, 20, "vap")
    blob = W.pack_blob(cpc, vap)
    S, F_ = 1024, 8
    audio = synth.noise_batch(S, 800 * F_, seed=4)
    a, b = engine.Engine(blob, 20, 2.5, max_streams=S), engine.Engine(blob, 20, 2.5, max_streams=S)
    fresh = engine.Engine(blob, 20, 2.5, max_streams=1)
    d_out = torch.zeros(S, engine.OUT_STRIDE, device="cuda")
    victim = 517
    for f in range(F_):
        new = np.ascontiguousarray(audio[:, :, f * 800:(f + 1) * 800])
        if f == 5:
            d_audio = torch.from_numpy(new).cuda()
            a.step_device(S, d_audio.data_ptr(), 800, d_out.data_ptr())      # a tick in flight on the GPU ...
            t0 = time.perf_counter()
            a.reset_stream(victim)                                           # ... does not block the reset call
            dt = time.perf_counter() - t0
            torch.cuda.synchronize()
            assert dt < 2e-3, f"reset_stream took {dt * 1e3:.2f} ms: it must not synchronise the device"
            b.step(new)
            continue
        oa, ob = a.step(new), b.step(new)
        if f < 5:
            np.testing.assert_array_equal(oa, ob)
        else:
            keep = np.arange(S) != victim
            np.testing.assert_array_equal(oa[keep], ob[keep])
            of = fresh.step(new[victim:victim + 1])
            np.testing.assert_allclose(oa[victim, :272], of[0, :272], rtol=0, atol=2e-5)   # batch 1024 vs 1: different conv-tail kernels
            assert oa[victim, engine.OUT_NVALID] == f - 5
    a.close(); b.close(); fresh.close()


def test_pinned_host_blocks_take_the_direct_dma_path():
    """Audio / out in vapx_host_alloc memory (engine.pinned_empty) give the same numbers as pageable numpy arrays."""
    from vap_realtime_amd import engine
    c = Case("multi3")
    a, b = make_engine(c), make_engine(c)
    pin_in = engine.pinned_empty((3, 2, c.hop))
    pin_out = engine.pinned_empty((3, engine.OUT_STRIDE))
    for f in range(6):
        new = c.new_samples(f)
        want = a.step(new)
        pin_in[...] = new
        got = b.step(pin_in, out=pin_out)
        assert got.ctypes.data == pin_out.ctypes.data
        np.testing.assert_array_equal(got, want)
    a.close(); b.close()
    del pin_in, pin_out, got


def test_changing_batch_size_under_deferred_join_is_safe():
    """VAPX_DEFER_JOIN with a batch size that changes between ticks re-slices the shared scratch: the engine joins the
    previous tick's groups itself.  Host-staged steps ignore the flag."""
    import torch
    from vap_realtime_amd import engine, synth, weights as W
    cpc, vap = W.synthetic_weights(5, 20, "vap")
    blob = W.pack_blob(cpc, vap)
    S, F_ = 128, 6
    audio = synth.noise_batch(S, 800 * F_, seed=3)
    one = engine.Engine(blob, 20, 2.5, max_streams=S)
    grp = engine.Engine(blob, 20, 2.5, max_streams=S, groups=2)
    sizes = [128, 96, 128, 64, 128, 80]
    d_audio = [torch.from_numpy(np.ascontiguousarray(audio[:n, :, f * 800:(f + 1) * 800])).cuda() for f, n in enumerate(sizes)]
    d_out = [torch.zeros(n, engine.OUT_STRIDE, device="cuda") for n in sizes]
    want = [one.step(audio[:n, :, f * 800:(f + 1) * 800]) for f, n in enumerate(sizes)]
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    for f, n in enumerate(sizes):
        grp.step_device(n, d_audio[f].data_ptr(), 800, d_out[f].data_ptr(), stream=side.cuda_stream, defer_join=True)
    grp.join(side.cuda_stream)
    side.synchronize()
    for f in range(F_):   # (not bit-equal: a 64-stream group takes other GEMM tile shapes than a 128-stream batch; a race would be gross)
        np.testing.assert_allclose(d_out[f].cpu().numpy()[:, :272], want[f][:, :272], rtol=0, atol=5e-5)
    one.close(); grp.close()


def test_long_run_does_not_drift():
    """220 frames (11 s): the persistent LSTM state and the ring / QKV cache wrap four times; the HIP path
    must stay within tolerance of the oracle all the way (no error accumulation)."""
    from oracle.vap_oracle import ServerFramer, VapOracle
    from vap_realtime_amd import engine, synth, weights as W
    cpc, vap = W.synthetic_weights(77, 20, "vap")
    o = VapOracle(cpc, vap, 20, 2.5)
    S, F_ = 2, 220
    audio = synth.dialogue_batch([90, 91], 800 * F_)
    st, fr = o.new_state(S), ServerFramer(S, 800)
    eng = engine.Engine(W.pack_blob(cpc, vap), 20, 2.5, max_streams=S)
    worst = 0.0
    for f in range(F_):
        new = audio[:, :, f * 800:(f + 1) * 800]
        want = o.step(fr.frame(new), st)
        got = engine.split_outputs(eng.step(new))
        for k in ("p_now", "p_future", "vad", "logits"):
            worst = max(worst, float(np.abs(got[k] - want[k]).max()))
    print("long run worst |diff| =", worst)
    assert worst <= TOL
    # the exported LSTM state still matches the oracle's
    stt = eng.get_state(0)
    np.testing.assert_allclose(stt["lstm"][:, 0], st.h[0].numpy(), rtol=0, atol=1e-5)
    np.testing.assert_allclose(stt["lstm"][:, 1], st.c[0].numpy(), rtol=0, atol=1e-4)
    eng.close()


def test_streams_joining_at_different_times():
    """Streams that start at different ticks sit in one batch with different window fills n (one still
    warming up, one already sliding) and a third one is reset mid-run: each must equal its own
    independent oracle run."""
    from oracle.vap_oracle import ServerFramer, VapOracle
    from vap_realtime_amd import engine, synth, weights as W
    cpc, vap = W.synthetic_weights(5, 20, "vap")
    F_ = 70
    audio = synth.dialogue_batch([200, 201, 202], 800 * F_)
    start = [0, 37, 12]                   # first tick of each stream
    eng = engine.Engine(W.pack_blob(cpc, vap), 20, 2.5, max_streams=4)
    oracles = [VapOracle(cpc, vap, 20, 2.5) for _ in range(3)]
    states = [o.new_state(1) for o in oracles]
    framers = [ServerFramer(1, 800) for _ in range(3)]
    for f in range(F_):
        if f == 55:                       # stream 2 hangs up and a new dialogue takes its slot
            eng.reset_stream(3)
            states[2], framers[2] = oracles[2].new_state(1), ServerFramer(1, 800)
        live = [s for s in range(3) if f >= start[s]]
        ids = [[0, 2, 3][s] for s in live]
        new = np.stack([audio[s, :, (f - start[s]) * 800:(f - start[s] + 1) * 800] for s in live])
        got = engine.split_outputs(eng.step(new, ids))
        for k, s in enumerate(live):
            want = oracles[s].step(framers[s].frame(new[k:k + 1]), states[s])
            for key in ("p_now", "p_future", "vad", "logits"):
                np.testing.assert_allclose(got[key][k], want[key][0], rtol=0, atol=TOL, err_msg=f"frame {f} stream {s} {key}")
    eng.close()


@pytest.mark.parametrize("S", [67, 300])
def test_odd_batch_sizes_hit_every_tile_tail(S):
    """Batches that are not multiples of any tile size (32/64-row GEMM tiles, 16-row LSTM tiles, 8-stream
    head groups, 4-row gather blocks), with a shuffled slot assignment: every stream must match the
    batched oracle."""
    from oracle.vap_oracle import ServerFramer, VapOracle
    from vap_realtime_amd import engine, synth, weights as W
    cpc, vap = W.synthetic_weights(11, 20, "vap")
    o = VapOracle(cpc, vap, 20, 2.5)
    F_ = 4
    audio = synth.noise_batch(S, 800 * F_, seed=S) * np.linspace(0.2, 3.0, S, dtype=np.float32)[:, None, None]
    st, fr = o.new_state(S), ServerFramer(S, 800)
    eng = engine.Engine(W.pack_blob(cpc, vap), 20, 2.5, max_streams=S + 5)
    ids = np.random.default_rng(S).permutation(S + 5)[:S].astype(np.int32)
    for f in range(F_):
        new = audio[:, :, f * 800:(f + 1) * 800]
        want = o.step(fr.frame(new), st)
        got = engine.split_outputs(eng.step(new, ids))
        for k in ("p_now", "p_future", "vad", "logits", "e"):
            np.testing.assert_allclose(got[k], want[k], rtol=0, atol=TOL, err_msg=f"S={S} frame {f} {k}")
    eng.close()


def test_long_window_fused_projections_equal_the_gemm_chain():
    """T = 250: the attention output projections (+ residual + LN, + cross-attention query projection) ride in the fused flat-row
    blocks (ffn_block modes 1 / 2) by default; VAPX_FLAG_UNFUSED_PROJ keeps the separate GEMM launches.  Same numbers, odd batch,
    window still filling and then sliding."""
    from vap_realtime_amd import engine, synth, weights as W
    cpc, vap = W.synthetic_weights(21, 50, "vap")
    blob = W.pack_blob(cpc, vap)
    S, F_, hop = 5, 260, 320
    audio = synth.noise_batch(S, hop * F_, seed=2) * np.linspace(0.3, 2.0, S, dtype=np.float32)[:, None, None]
    fused = engine.Engine(blob, 50, 5.0, max_streams=S)
    plain = engine.Engine(blob, 50, 5.0, max_streams=S, unfused_proj=True)
    worst = 0.0
    for f in range(F_):
        a = audio[:, :, f * hop:(f + 1) * hop]
        got, want = fused.step(a), plain.step(a)
        worst = max(worst, float(np.abs(got[:, :272] - want[:, :272]).max()))
    print("fused vs unfused long-window projections: worst |diff| =", worst)
    assert worst <= 3e-5
    fused.close(); plain.close()


@pytest.mark.parametrize("T", [32, 33, 36, 37, 44, 49, 52, 53, 64])
def test_short_windows_around_the_small_tile_projection(T):
    """The fused attention block runs its projections on 32 + 5 x 4 rows for windows of 33..52 frames (csrc/fused_blocks.hip, round 5) and on two
    32-row tiles otherwise: every boundary (32 | 33, 52 | 53), a window that fills a 4-row group exactly (36, 44, 52) and one that leaves one
    to three rows of a group empty (33, 37, 49), plus the 64-row limit of the fused path — window filling, full and sliding, against the oracle."""
    from oracle.vap_oracle import ServerFramer, VapOracle
    from vap_realtime_amd import engine, synth, weights as W
    hz = 20
    ctx = T / hz
    cpc, vap = W.synthetic_weights(17, hz, "vap")
    o = VapOracle(cpc, vap, hz, ctx)
    hop = 16000 // hz
    S, F_ = 3, T + 4
    audio = synth.dialogue_batch([80, 81, 82], hop * F_)
    st, fr = o.new_state(S), ServerFramer(S, hop)
    eng = engine.Engine(W.pack_blob(cpc, vap), hz, ctx, max_streams=S)
    assert eng.T == T
    worst = 0.0
    for f in range(F_):
        new = audio[:, :, f * hop:(f + 1) * hop]
        want = o.step(fr.frame(new), st)
        got = engine.split_outputs(eng.step(new))
        for k in ("p_now", "p_future", "vad", "logits"):
            worst = max(worst, float(np.abs(got[k] - want[k]).max()))
    print(f"T={T}: worst |hip - oracle| = {worst:.2e}")
    assert worst <= TOL
    eng.close()


@pytest.mark.parametrize("split", [False, True], ids=["fp32", "split_f16"])
@pytest.mark.parametrize("hz,ctx", [(20, 5.0), (20, 7.5), (10, 10.0)])
def test_mid_length_windows_against_the_oracle(hz, ctx, split):
    """Windows between the fused-block limit (64) and 256 rows — the reference's published bc (20 Hz / 5 s, T = 100) and nod
    (10 Hz / 10 s, T = 100) settings, and T = 150 (5 key tiles: a partly used second chunk of the online softmax) — run the
    long-window attention kernel + flat-row projection blocks; window filling, full, and sliding.  On the split-precision path
    the same windows leave some of the persistent attention kernel's waves without a valid query tile (7 or 10 of its 16)."""
    from oracle.vap_oracle import ServerFramer, VapOracle
    from vap_realtime_amd import engine, synth, weights as W
    cpc, vap = W.synthetic_weights(13, hz, "vap")
    o = VapOracle(cpc, vap, hz, ctx)
    T, hop = int(ctx * hz), 16000 // hz
    S, F_ = 2, T + 5
    audio = synth.dialogue_batch([70, 71], hop * F_)
    st, fr = o.new_state(S), ServerFramer(S, hop)
    eng = engine.Engine(W.pack_blob(cpc, vap), hz, ctx, max_streams=S, split_f16=split)
    assert eng.T == T
    worst = 0.0
    for f in range(F_):
        new = audio[:, :, f * hop:(f + 1) * hop]
        want = o.step(fr.frame(new), st)
        got = engine.split_outputs(eng.step(new))
        for k in ("p_now", "p_future", "vad", "logits"):
            worst = max(worst, float(np.abs(got[k] - want[k]).max()))
    print(f"T={T} @ {hz} Hz: worst |hip - oracle| = {worst:.2e}")
    assert worst <= TOL
    eng.close()


@pytest.mark.parametrize("hz,ctx,split", [(50, 6.0, False), (50, 6.0, True), (50, 10.24, False)], ids=["T300_fp32", "T300_split", "T512_fp32"])
def test_windows_beyond_256_frames_against_the_oracle(hz, ctx, split):
    """The reference's ALiBi transformer takes any T (modules.py:303-308); windows of 257 .. 512 frames run through attention_xl_kernel
    (csrc/vap_kernels.hip: K / V from L2, online softmax over up to 16 key tiles) on both precision paths: T = 300 (a partly used tenth key
    tile) and the limit T = 512, window filling, full and sliding."""
    from oracle.vap_oracle import ServerFramer, VapOracle
    from vap_realtime_amd import engine, synth, weights as W
    cpc, vap = W.synthetic_weights(19, hz, "vap")
    o = VapOracle(cpc, vap, hz, ctx)
    T, hop = int(ctx * hz), 16000 // hz
    S, F_ = 2, T + 3
    audio = synth.dialogue_batch([90, 91], hop * F_)
    st, fr = o.new_state(S), ServerFramer(S, hop)
    eng = engine.Engine(W.pack_blob(cpc, vap), hz, ctx, max_streams=S, split_f16=split)
    assert eng.T == T and T > 256
    worst = 0.0
    check = set(range(0, F_, 37)) | set(range(T - 3, F_))        # the oracle recomputes the whole window per frame: compare a sample of the fill, then full + sliding
    for f in range(F_):
        new = audio[:, :, f * hop:(f + 1) * hop]
        want = o.step(fr.frame(new), st)
        got = engine.split_outputs(eng.step(new))
        if f in check:
            for k in ("p_now", "p_future", "vad", "logits"):
                worst = max(worst, float(np.abs(got[k] - want[k]).max()))
    print(f"T={T} @ {hz} Hz: worst |hip - oracle| = {worst:.2e}")
    assert worst <= TOL
    eng.close()


def test_xl_attention_kernel_equals_the_long_window_kernel(monkeypatch):
    """attention_xl_kernel on a window the tuned kernel also takes (T = 250): the same outputs to rounding (the key tiles are visited in the
    same order with the same arithmetic; only where K / V come from differs)."""
    from vap_realtime_amd import engine, synth, weights as W
    hz, ctx = 50, 5.0
    cpc, vap = W.synthetic_weights(23, hz, "vap")
    blob = W.pack_blob(cpc, vap)
    T, hop = int(ctx * hz), 16000 // hz
    S, F_ = 3, T + 2
    audio = synth.dialogue_batch([5, 6, 7], hop * F_)
    outs = []
    for force in (False, True):
        if force:
            monkeypatch.setenv("VAPX_FORCE_ATTENTION_XL", "1")
        eng = engine.Engine(blob, hz, ctx, max_streams=S)
        got = [eng.step(audio[:, :, f * hop:(f + 1) * hop]).copy() for f in range(F_)]
        outs.append(np.stack(got))
        eng.close()
    worst = float(np.abs(outs[0][:, :, :272] - outs[1][:, :, :272]).max())
    print("attention_xl vs attention_long2: worst |diff| =", worst)
    assert worst <= 2e-5


@pytest.mark.parametrize("mode,hz,ctx", [("nod", 10, 10.0), ("bc", 20, 5.0)])
def test_published_bc_and_nod_settings_against_the_oracle(mode, hz, ctx):
    """The reference's README settings for the fine-tuned heads — vap_bc_main at 20 Hz / 5 s and vap_nod_main at 10 Hz / 10 s,
    both T = 100 — run the long-window kernels; nod additionally runs the FULL last layer and emits p_bc for every window row
    (vap_nod_main.py:276)."""
    from oracle.vap_oracle import ServerFramer, VapOracle
    from vap_realtime_amd import engine, synth, weights as W
    cpc, vap = W.synthetic_weights(17, hz, mode)
    o = VapOracle(cpc, vap, hz, ctx, mode=mode)
    T, hop = int(ctx * hz), 16000 // hz
    S, F_ = 2, T + 3
    audio = synth.dialogue_batch([80, 81], hop * F_)
    st, fr = o.new_state(S), ServerFramer(S, hop)
    eng = engine.Engine(W.pack_blob(cpc, vap, mode), hz, ctx, max_streams=S, mode=mode)
    worst = 0.0
    for f in range(F_):
        new = audio[:, :, f * hop:(f + 1) * hop]
        want = o.step(fr.frame(new), st)
        got = engine.split_outputs(eng.step(new))
        n = min(f + 1, T)
        worst = max(worst, float(np.abs(got["vad"] - want["vad"]).max()))
        if mode == "bc":
            worst = max(worst, float(np.abs(got["aux"][:, 1] - want["p_bc_react"]).max()), float(np.abs(got["aux"][:, 2] - want["p_bc_emo"]).max()))
        else:
            for k, col in (("p_nod_short", 1), ("p_nod_long", 2), ("p_nod_long_p", 3)):
                worst = max(worst, float(np.abs(got["aux"][:, col] - want[k]).max()))
            worst = max(worst, float(np.abs(got["logits"][:, :n] - want["p_bc"][:, :n]).max()))
    print(f"{mode} T={T} @ {hz} Hz: worst |hip - oracle| = {worst:.2e}")
    assert worst <= TOL
    eng.close()
