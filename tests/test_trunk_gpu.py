"""Shared-trunk multi-model serving: vap + bc + nod weight sets on one CPC CNN + LSTM pass per tick,
against goldens of the three reference programs run side by side on one cpc_model file and the same audio."""
import numpy as np
import pytest

from golden_util import Case

pytestmark = pytest.mark.gpu

TOL = 1e-4


def _group(order=("vap", "bc", "nod"), max_streams=2, **kw):
    from vap_realtime_amd import engine, weights as W
    cases = {"vap": Case("trunk_vap20"), "bc": Case("trunk_bc20"), "nod": Case("trunk_nod20")}
    blobs = {m: W.pack_blob(cases[m].cpc_sd, cases[m].vap_sd, m) for m in order}
    c = cases["vap"]
    return cases, engine.TrunkGroup(blobs, c.frame_hz, c.ctx_sec, max_streams=max_streams, **kw)


def _check_frame(cases, res, f, sl=slice(None)):
    from vap_realtime_amd.engine import split_outputs
    cv, cb, cn = cases["vap"], cases["bc"], cases["nod"]
    if "vap" in res:
        o = split_outputs(res["vap"])
        for k in ("p_now", "p_future", "vad", "logits"):
            np.testing.assert_allclose(o[k][sl], cv.z[k][f], rtol=0, atol=TOL, err_msg=f"vap {k} frame {f}")
    if "bc" in res:
        o = split_outputs(res["bc"])
        np.testing.assert_allclose(o["aux"][sl, 1], cb.z["p_bc_react"][f].reshape(-1), rtol=0, atol=TOL)
        np.testing.assert_allclose(o["aux"][sl, 2], cb.z["p_bc_emo"][f].reshape(-1), rtol=0, atol=TOL)
    if "nod" in res:
        o = split_outputs(res["nod"])
        np.testing.assert_allclose(o["aux"][sl, 1], cn.z["p_nod_short"][f].reshape(-1), rtol=0, atol=TOL)
        np.testing.assert_allclose(o["aux"][sl, 2], cn.z["p_nod_long"][f].reshape(-1), rtol=0, atol=TOL)
        np.testing.assert_allclose(o["aux"][sl, 3], cn.z["p_nod_long_p"][f].reshape(-1), rtol=0, atol=TOL)
        n = min(f + 1, cn.T)
        np.testing.assert_allclose(o["logits"][sl, :n], cn.z["p_bc"][f][:, :n], rtol=0, atol=TOL)


@pytest.mark.parametrize("order", [("vap", "bc", "nod"), ("nod", "vap"), ("bc", "nod")])
def test_three_models_one_trunk_match_three_reference_programs(order):
    cases, grp = _group(order)
    c = cases["vap"]
    for f in range(c.n_frames):
        _check_frame(cases, grp.step(c.new_samples(f)), f)
    grp.close()


def test_follower_equals_standalone_engine_with_ids_and_reset():
    """Followers with permuted stream ids in a larger state table, one stream reset mid-run: every output equals a
    stand-alone engine of that model fed the same audio (which is golden-pinned by test_engine_gpu)."""
    from vap_realtime_amd import engine, weights as W
    cases, grp = _group(("vap", "nod"), max_streams=5)
    c = cases["vap"]
    solo = engine.Engine(W.pack_blob(cases["nod"].cpc_sd, cases["nod"].vap_sd, "nod"), c.frame_hz, c.ctx_sec, max_streams=5, mode="nod")
    ids = [4, 1]
    for f in range(30):
        if f == 17:
            grp.reset_stream(4)
            solo.reset_stream(4)
        a = c.new_samples(f)
        got = grp.step(a, ids)["nod"]
        want = solo.step(a, ids)
        np.testing.assert_allclose(got[:, :16], want[:, :16], rtol=0, atol=2e-5, err_msg=f"frame {f}")
        n = int(want[0, 10])
        np.testing.assert_allclose(got[:, 16:16 + n], want[:, 16:16 + n], rtol=0, atol=2e-5)
    assert int(got[0, 10]) == 13 and int(got[1, 10]) == 30        # stream 4 restarted at frame 17
    st = grp.engines["nod"].get_state(1)
    assert st["n_frames"] == 30 and st["lstm"] is None
    np.testing.assert_allclose(st["ring"], solo.get_state(1)["ring"], rtol=0, atol=2e-5)
    solo.close()
    grp.close()


def test_trunk_error_paths():
    from vap_realtime_amd import engine, weights as W
    from vap_realtime_amd.engine import VapxError
    cases, grp = _group(("vap", "bc"))
    c = cases["vap"]
    fol = grp.engines["bc"]
    with pytest.raises(VapxError, match="leader first"):
        fol.step_follow(2)                                  # no encoder output yet
    with pytest.raises(VapxError, match="no audio"):
        fol.step(c.new_samples(0))
    grp.step(c.new_samples(0))
    with pytest.raises(VapxError, match="leader first"):
        fol.step_follow(2)                                  # same tick twice
    grp.leader.step(c.new_samples(1)[:1])
    with pytest.raises(VapxError, match="differs from the leader"):
        fol.step_follow(2)
    with pytest.raises(VapxError, match="trunk leader"):
        fol.reset_stream(0)
    # different CPC weights: nothing to share
    other_cpc, _ = W.synthetic_weights(123, c.frame_hz)
    e2 = engine.Engine(W.pack_blob(other_cpc, cases["bc"].vap_sd, "bc"), c.frame_hz, c.ctx_sec, max_streams=2, mode="bc")
    lead2 = engine.Engine(W.pack_blob(c.cpc_sd, c.vap_sd, "vap"), c.frame_hz, c.ctx_sec, max_streams=2)
    with pytest.raises(VapxError, match="CPC encoder weights differ"):
        e2.attach_trunk(lead2)
    e3 = engine.Engine(W.pack_blob(c.cpc_sd, cases["bc"].vap_sd, "bc"), c.frame_hz, c.ctx_sec, max_streams=3, mode="bc")
    with pytest.raises(VapxError, match="must match the leader"):
        e3.attach_trunk(lead2)
    with pytest.raises(VapxError, match="not itself a follower"):
        lead2.attach_trunk(fol)
    for e in (e2, e3, lead2):
        e.close()
    grp.close()
