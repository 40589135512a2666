"""Pin the oracle (oracle/vap_oracle.py) against golden vectors produced by the imported,
unmodified reference (tools/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest

from golden_util import Case
from oracle.vap_oracle import ServerFramer, VapOracle

TOL = 2e-5          # logits / embeddings: fp32 summation-order noise between two CPU formulations
TOL_P = 2e-6        # probabilities
TOL_RESID = 1e-3    # raw residual-stream tensors (magnitudes ~1e2)


def run_oracle(case, collect_frames=()):
    o = VapOracle(case.cpc_sd, case.vap_sd, case.frame_hz, case.ctx_sec, case.mode)
    S = len(case.streams)
    st = o.new_state(S)
    fr = ServerFramer(S, case.hop)
    outs, cols = [], {}
    for f in range(case.n_frames):
        col = {} if f in collect_frames else None
        buf = fr.frame(case.new_samples(f)) if case.framing == "server" else case.window(f)
        outs.append(o.step(buf, st, col))
        if col is not None:
            cols[f] = col
    return outs, cols


@pytest.mark.parametrize("name", ["vap20", "vap10", "offline20", "multi3", "vap50", "degenerate20", "poison20", "vap20_10s"])
def test_oracle_matches_reference_outputs(name):
    """degenerate20: silence / near-denormal / clipping / 3e3 x / DC offset / dead channel — where ChannelNorm divides by ~0
    (encoder_components.py:64-66); poison20: one NaN and one Inf SAMPLE — the reference's outputs turn NaN for good (equal_nan
    comparison: same positions); vap20_10s: T = 200, the longest published window (README.md:381)."""
    c = Case(name)
    z = c.z
    es = int(z["meta.e_stride"]) if "meta.e_stride" in z.files else 1
    inter_frames = sorted({int(k.split(".")[1][1:]) for k in z.files if k.startswith("inter.f")})
    outs, cols = run_oracle(c, inter_frames)
    for f, out in enumerate(outs):
        np.testing.assert_allclose(out["p_now"], z["p_now"][f], rtol=0, atol=TOL_P)
        np.testing.assert_allclose(out["p_future"], z["p_future"][f], rtol=0, atol=TOL_P)
        np.testing.assert_allclose(out["vad"], z["vad"][f], rtol=0, atol=TOL_P)
        np.testing.assert_allclose(out["logits"], z["logits"][f], rtol=0, atol=TOL)
        if f % es == 0:
            np.testing.assert_allclose(out["e"], z["e"][f // es], rtol=0, atol=TOL)
    for f in inter_frames:
        col = cols[f]
        rows = z[f"inter.f{f}.rows"]
        np.testing.assert_allclose(col["cnn4"][0].numpy(), z[f"inter.f{f}.cnn4"], rtol=0, atol=TOL)
        np.testing.assert_allclose(col["lstm_out"][0].numpy(), z[f"inter.f{f}.lstm_out"], rtol=0, atol=TOL)
        np.testing.assert_allclose(col["o"][0].numpy()[:, rows], z[f"inter.f{f}.o"], rtol=0, atol=TOL_RESID)
        for l in range(3):
            np.testing.assert_allclose(col[f"stereo{l}"][0].numpy()[:, rows], z[f"inter.f{f}.stereo{l}"],
                                       rtol=2e-5, atol=TOL_RESID)
        np.testing.assert_allclose(col["comb"][0].numpy()[rows], z[f"inter.f{f}.comb"], rtol=0, atol=TOL)


def test_oracle_bc_heads():
    c = Case("bc20")
    outs, _ = run_oracle(c)
    for f, out in enumerate(outs):
        np.testing.assert_allclose(out["p_bc_react"], c.z["p_bc_react"][f].reshape(-1), rtol=0, atol=TOL_P)
        np.testing.assert_allclose(out["p_bc_emo"], c.z["p_bc_emo"][f].reshape(-1), rtol=0, atol=TOL_P)


@pytest.mark.parametrize("name", ["nod20", "nod20_10s"])
def test_oracle_nod_heads(name):
    c = Case(name)
    outs, _ = run_oracle(c)
    for f, out in enumerate(outs):
        for k in ("p_nod_short", "p_nod_long", "p_nod_long_p"):
            np.testing.assert_allclose(out[k], c.z[k][f].reshape(-1), rtol=0, atol=TOL_P)
        n = min(f + 1, c.T)
        np.testing.assert_allclose(out["p_bc"][:, :n], c.z["p_bc"][f][:, :n], rtol=0, atol=TOL_P)


def test_oracle_models_sharing_one_cpc_file():
    """trunk_* goldens: three reference programs (vap, bc, nod) loaded from ONE cpc_model file, same audio."""
    cv, cb, cn = Case("trunk_vap20"), Case("trunk_bc20"), Case("trunk_nod20")
    for k in cv.cpc_sd:
        assert np.array_equal(cv.cpc_sd[k], cb.cpc_sd[k]) and np.array_equal(cv.cpc_sd[k], cn.cpc_sd[k])
    assert not np.array_equal(cv.vap_sd["ar.layers.0.mha.key.weight"], cb.vap_sd["ar.layers.0.mha.key.weight"])
    ov, _ = run_oracle(cv)
    ob, _ = run_oracle(cb)
    on, _ = run_oracle(cn)
    for f in range(cv.n_frames):
        np.testing.assert_allclose(ov[f]["logits"], cv.z["logits"][f], rtol=0, atol=TOL)
        np.testing.assert_allclose(ov[f]["p_now"], cv.z["p_now"][f], rtol=0, atol=TOL_P)
        np.testing.assert_allclose(ob[f]["p_bc_react"], cb.z["p_bc_react"][f].reshape(-1), rtol=0, atol=TOL_P)
        np.testing.assert_allclose(ob[f]["p_bc_emo"], cb.z["p_bc_emo"][f].reshape(-1), rtol=0, atol=TOL_P)
        for k in ("p_nod_short", "p_nod_long", "p_nod_long_p"):
            np.testing.assert_allclose(on[f][k], cn.z[k][f].reshape(-1), rtol=0, atol=TOL_P)


def test_batched_oracle_equals_independent_runs():
    """multi3 golden = three independent reference processes; the oracle runs them as one batch."""
    c = Case("multi3")
    outs, _ = run_oracle(c)
    # different streams must actually differ (guards against a broadcast bug)
    assert np.abs(outs[-1]["logits"][0] - outs[-1]["logits"][1]).max() > 1e-2


def test_poison_golden_records_what_the_reference_does_with_a_nan_sample():
    """Pins the FACT the HIP path has to reproduce (tests/test_engine_gpu.py::test_poisoned_sample_behaves_like_the_reference):
    from the frame holding the NaN / Inf sample on, p_now / p_future / logits and the VAD of the poisoned channel are NaN for
    good; the other channel's VAD (ar_channel output of channel 2 only, vap_main.py:292-293) and the clean stream stay finite."""
    c = Case("poison20")
    z = c.z
    assert c.kinds == ["clean", "nan_sample", "inf_sample"]
    for s in (1, 2):
        assert np.isfinite(z["logits"][:3, s]).all() and np.isnan(z["logits"][3:, s]).all()
        assert np.isnan(z["p_now"][3:, s]).all() and np.isnan(z["p_future"][3:, s]).all()
        assert np.isnan(z["vad"][3:, s, 0]).all() and np.isfinite(z["vad"][:, s, 1]).all()
    assert np.isfinite(z["logits"][:, 0]).all() and np.isfinite(z["vad"][:, 0]).all()


def test_every_fixture_carries_every_key_the_current_generator_writes():
    """The fixtures are regenerated by tools/make_golden.py from the unmodified reference; a fixture written by an older version of
    the script (round 4: vap20 / bc20 lacked meta.cpc_seed / meta.e_stride) must not linger: script and data have to agree."""
    import importlib.util
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "tools", "make_golden.py")
    src = open(path).read()
    written = set(re.findall(r'out\["(meta\.[a-z_]+)"\]', src))
    assert {"meta.cpc_seed", "meta.e_stride", "meta.weights_fp", "meta.audio_fp"} <= written
    spec = importlib.util.spec_from_file_location("make_golden", path)
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    for name, cfg in mg.CASES.items():
        z = np.load(os.path.join(root, "tests", "golden", f"{name}.npz"))
        need = written - ({"meta.kinds"} if "kinds" not in cfg else set())
        missing = sorted(need - set(z.files))
        assert not missing, (name, missing)
        assert int(z["meta.e_stride"]) == cfg.get("e_stride", 1) and int(z["meta.cpc_seed"]) == cfg.get("cpc_seed", cfg["seed"])
        assert int(z["meta.n_frames"]) == cfg["n_frames"] and [int(s) for s in z["meta.streams"]] == cfg["streams"]
