"""Opt-in split-precision FFN block (VAPX_FLAG_SPLIT_F16): fp32-accurate products from three f16 MFMA terms.
It must meet the same 1e-4 bar against the reference goldens as the default fp32-MFMA path, and its deviation from
the goldens must be of the same size as the fp32 path's (no precision is traded)."""
import numpy as np
import pytest

from golden_util import Case

pytestmark = pytest.mark.gpu

TOL = 1e-4


def _run(case, **kw):
    from vap_realtime_amd import engine, weights as W
    eng = engine.Engine(W.pack_blob(case.cpc_sd, case.vap_sd, case.mode), case.frame_hz, case.ctx_sec,
                        max_streams=len(case.streams), mode=case.mode, **kw)
    worst = {k: 0.0 for k in ("p_now", "p_future", "vad", "logits")}
    outs = []
    for f in range(case.n_frames):
        audio = case.new_samples(f) if case.framing == "server" else case.window(f)
        o = engine.split_outputs(eng.step(audio))
        outs.append(o["logits"].copy())
        for k in worst:
            worst[k] = max(worst[k], float(np.abs(o[k] - case.z[k][f]).max()))
    eng.close()
    return worst, np.stack(outs)


@pytest.mark.parametrize("name", ["vap20", "multi3", "vap10", "vap50", "offline20", "degenerate20", "vap20_10s"])
def test_split_f16_path_meets_the_reference_tolerance(name):
    c = Case(name)
    w32, l32 = _run(c)
    w16, l16 = _run(c, split_f16=True)
    print(name, "fp32 MFMA:", w32, "| f16x3 split:", w16, "| split vs fp32 logits:", float(np.abs(l16 - l32).max()))
    for k, v in w16.items():
        assert v <= TOL, (name, k, v)
    # same error class as the fp32 path: within 3x of its deviation from the reference (both ~1e-5 on logits)
    assert w16["logits"] <= 3.0 * max(w32["logits"], 4e-6)
    assert float(np.abs(l16 - l32).max()) > 0.0          # the flag really switched kernels


def test_split_f16_with_full_last_layer_and_unfused_variants():
    """The f16x3 FFN block also feeds the unfused / unpruned consumers (Q|K|V and cross K|V stores, 2-chunk K|V)."""
    from vap_realtime_amd import engine, weights as W
    c = Case("vap20")
    blob = W.pack_blob(c.cpc_sd, c.vap_sd)
    ref = engine.Engine(blob, 20, 2.5, max_streams=1)
    var = [engine.Engine(blob, 20, 2.5, max_streams=1, split_f16=True, full_last_layer=True),
           engine.Engine(blob, 20, 2.5, max_streams=1, split_f16=True, unfused_last_row=True)]
    for f in range(12):
        a = c.new_samples(f)
        want = ref.step(a)
        for e in var:
            np.testing.assert_allclose(e.step(a)[:, :272], want[:, :272], rtol=0, atol=3e-5)
    for e in var + [ref]:
        e.close()


def test_rounding_error_against_a_float64_ground_truth():
    """Both arithmetic variants against the float64 restatement of the step (the same algorithm in double precision):
    the split path's error must not exceed the fp32-MFMA path's by more than a small factor — it is a different
    summation, not a lower precision.  The torch-CPU fp32 oracle is measured alongside for scale."""
    import torch
    from oracle.vap_oracle import ServerFramer, VapOracle
    from vap_realtime_amd import engine, synth, weights as W
    cpc, vap = W.synthetic_weights(31, 20, "vap")
    S, F_ = 2, 56
    audio = synth.dialogue_batch([40, 41], 800 * F_)
    o64 = VapOracle(cpc, vap, 20, 2.5, dtype=torch.float64)
    o32 = VapOracle(cpc, vap, 20, 2.5)
    s64, s32 = o64.new_state(S), o32.new_state(S)
    f64, f32 = ServerFramer(S, 800), ServerFramer(S, 800)
    blob = W.pack_blob(cpc, vap)
    e32 = engine.Engine(blob, 20, 2.5, max_streams=S)
    e16 = engine.Engine(blob, 20, 2.5, max_streams=S, split_f16=True)
    err = {"hip fp32 MFMA": 0.0, "hip f16x3 split": 0.0, "torch-cpu fp32": 0.0}
    rms = dict.fromkeys(err, 0.0)
    for f in range(F_):
        new = audio[:, :, f * 800:(f + 1) * 800]
        truth = o64.step(f64.frame(new), s64)["logits"]
        cand = {"hip fp32 MFMA": engine.split_outputs(e32.step(new))["logits"],
                "hip f16x3 split": engine.split_outputs(e16.step(new))["logits"],
                "torch-cpu fp32": o32.step(f32.frame(new), s32)["logits"]}
        for k, v in cand.items():
            d = np.abs(v.astype(np.float64) - truth)
            err[k] = max(err[k], float(d.max()))
            rms[k] += float((d ** 2).mean()) / F_
    print("max |logits - float64 truth|:", {k: f"{v:.2e}" for k, v in err.items()},
          "rms:", {k: f"{np.sqrt(v):.2e}" for k, v in rms.items()})
    assert err["hip fp32 MFMA"] <= 1e-4 and err["hip f16x3 split"] <= 1e-4
    assert err["hip f16x3 split"] <= 2.0 * err["hip fp32 MFMA"] + 2e-6
    assert np.sqrt(rms["hip f16x3 split"]) <= 2.0 * np.sqrt(rms["hip fp32 MFMA"]) + 5e-7
    e32.close(); e16.close()


@pytest.mark.parametrize("name", ["bc20", "nod20", "nod20_10s"])
def test_split_f16_aux_heads(name):
    """bc / nod weight sets on the split path (nod runs the full last layer: every FFN-block variant is exercised)."""
    from vap_realtime_amd import engine, weights as W
    c = Case(name)
    eng = engine.Engine(W.pack_blob(c.cpc_sd, c.vap_sd, c.mode), c.frame_hz, c.ctx_sec, max_streams=1, mode=c.mode, split_f16=True)
    for f in range(c.n_frames):
        o = engine.split_outputs(eng.step(c.new_samples(f)))
        if name == "bc20":
            np.testing.assert_allclose(o["aux"][:, 1], c.z["p_bc_react"][f].reshape(-1), rtol=0, atol=TOL)
            np.testing.assert_allclose(o["aux"][:, 2], c.z["p_bc_emo"][f].reshape(-1), rtol=0, atol=TOL)
        else:
            for col, key in ((1, "p_nod_short"), (2, "p_nod_long"), (3, "p_nod_long_p")):
                np.testing.assert_allclose(o["aux"][:, col], c.z[key][f].reshape(-1), rtol=0, atol=TOL)
            n = min(f + 1, c.T)
            np.testing.assert_allclose(o["logits"][:, :n], c.z["p_bc"][f][:, :n], rtol=0, atol=TOL)
    eng.close()


def test_split_f16_five_hz_and_odd_batch_against_the_oracle():
    """5 Hz (K = 20 CPC frames per VAP frame, conv GEMMs with the unfused tail) and a batch that is no multiple of any tile."""
    from oracle.vap_oracle import ServerFramer, VapOracle
    from vap_realtime_amd import engine, synth, weights as W
    cpc, vap = W.synthetic_weights(17, 5, "vap")
    S, F_, hop = 37, 5, 3200
    o = VapOracle(cpc, vap, 5, 4.0)
    audio = synth.noise_batch(S, hop * F_, seed=4) * np.linspace(0.2, 2.5, S, dtype=np.float32)[:, None, None]
    st, fr = o.new_state(S), ServerFramer(S, hop)
    eng = engine.Engine(W.pack_blob(cpc, vap), 5, 4.0, max_streams=S, split_f16=True)
    for f in range(F_):
        new = audio[:, :, f * hop:(f + 1) * hop]
        want = o.step(fr.frame(new), st)
        got = engine.split_outputs(eng.step(new))
        for k in ("p_now", "p_future", "vad", "logits"):
            np.testing.assert_allclose(got[k], want[k], rtol=0, atol=TOL, err_msg=f"{k} frame {f}")
    eng.close()


@pytest.mark.parametrize("hz,ctx", [(20, 2.5), (50, 5.0)])
def test_split_f16_handles_activations_far_beyond_the_f16_range(hz, ctx):
    """f16 operands overflow at 65504 — the split path must never let an input get there.  Context rows of magnitude 1e6 and 3e9 (inputs
    the fp32-MFMA path digests) hit every raw-row operand: the cross K / V projection of the FFN block, the attention output (a convex
    combination of such V rows), the long-window projection blocks.  Each is scaled per row / per slab by a power of two and un-scaled
    exactly in the accumulator, so the split engine must stay finite, flag nothing and agree with the fp32 engine like on ordinary
    input.  (Round 2 reported such a stream as VAPX_E_NUMERIC instead.)"""
    from vap_realtime_amd import engine, synth, weights as W
    cpc, vap = W.synthetic_weights(41, hz, "vap")
    blob = W.pack_blob(cpc, vap)
    hop = 16000 // hz
    f32 = engine.Engine(blob, hz, ctx, max_streams=3)
    f16 = engine.Engine(blob, hz, ctx, max_streams=3, split_f16=True)
    audio = synth.dialogue_batch([90, 91, 92], hop * 9)
    frames = [np.ascontiguousarray(audio[:, :, f * hop:(f + 1) * hop]) for f in range(9)]
    for f in range(5):
        f32.step(frames[f]); f16.step(frames[f])
    for eng in (f32, f16):
        for sid, mag in ((1, 1e6), (2, 3e9)):
            st = eng.get_state(sid)
            st["ring"][:, :st["n_frames"]] *= mag / max(1e-9, float(np.abs(st["ring"]).max()))
            eng.set_state(sid, st)
    worst = 0.0
    for f in range(5, 9):
        want = f32.step(frames[f])
        got = f16.step(frames[f])                                 # raises on VAPX_E_NUMERIC
        assert np.isfinite(got[:, :272]).all() and not got[:, engine.OUT_STATUS].any()
        worst = max(worst, float(np.abs(got[:, :272] - want[:, :272]).max()))
    print(f"{hz} Hz / T={int(hz * ctx)}: split vs fp32 with 1e6 / 3e9 context rows: worst |diff| = {worst:.2e}")
    assert worst <= 1e-4
    f32.close(); f16.close()


def test_split_f16_hidden_scale_from_the_weight_bound():
    """The GELU hidden row is bounded by the weights alone (its input is a LayerNorm output).  With ordinary weights the static scale is 1;
    with a first FFN matrix 300 x larger the bound passes 2^15 and vapx_create picks a power-of-two down-scale — the results still
    agree with the fp32 engine run on the same weights (relative: the FFN output itself is huge)."""
    from vap_realtime_amd import engine, synth, weights as W
    cpc, vap = W.synthetic_weights(43, 20, "vap")
    vap = dict(vap)
    for k in list(vap):
        if k.endswith("ffnetwork.0.weight"):
            vap[k] = (vap[k] * 300.0).astype(np.float32)
        if k.endswith("ffnetwork.3.weight"):
            vap[k] = (vap[k] / 300.0).astype(np.float32)        # keeps the residual stream in its usual range
    blob = W.pack_blob(cpc, vap)
    f32 = engine.Engine(blob, 20, 2.5, max_streams=2)
    f16 = engine.Engine(blob, 20, 2.5, max_streams=2, split_f16=True)
    audio = synth.dialogue_batch([93, 94], 800 * 6)
    worst = 0.0
    for f in range(6):
        new = np.ascontiguousarray(audio[:, :, f * 800:(f + 1) * 800])
        want, got = f32.step(new), f16.step(new)
        assert np.isfinite(got[:, :272]).all()
        worst = max(worst, float(np.abs(got[:, :272] - want[:, :272]).max()))
    print(f"hidden scale < 1: split vs fp32 worst |diff| = {worst:.2e}")
    assert worst <= 1e-4
    f32.close(); f16.close()


@pytest.mark.parametrize("name,kw", [("vap50", {}), ("vap20_10s", {}), ("vap50", {"full_last_layer": True}), ("vap50", {"unfused_last_row": True})])
def test_qkv_projected_inside_the_attention_kernel_equals_qkv_from_the_ffn_block(name, kw):
    """Round 5: on long windows the self-attention of layers 1-2 (and 3 when the last layer runs on all rows) projects its own Q|K|V
    (csrc/attention_proj_f16x3.hip: the previous layer's flat-row block only writes LN_self(x)).  Both routes — and the A/B flag
    VAPX_FLAG_SPLIT_QKV_IN_FFN that keeps the round-4 route — meet the reference goldens, and they differ only in summation order."""
    c = Case(name)
    w_new, l_new = _run(c, split_f16=True, **kw)
    w_old, l_old = _run(c, split_f16=True, split_qkv_in_ffn=True, **kw)
    for w in (w_new, w_old):
        for k, v in w.items():
            assert v <= TOL, (name, kw, k, v)
    d = float(np.abs(l_new - l_old).max())
    print(name, kw, "new", w_new["logits"], "old", w_old["logits"], "new vs old", d)
    assert 0.0 < d <= 5e-5                               # the flag really switches kernels, and nothing but rounding moves
