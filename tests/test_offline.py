"""Offline driver: framing / timestamps / CSV text follow rvap/vap_main/vap_offline.py (host logic on CPU with
a test double; the same driver against the real engine and the offline20 golden on the GPU)."""
import os
import wave

import numpy as np
import pytest

from vap_realtime_amd import offline


class FakeVap:
    hop = 800

    def __init__(self):
        self.calls = []

    def process(self, frames, ids):
        self.calls.append((frames.copy(), list(ids)))
        m = frames.mean(axis=2)
        return {"p_now": m, "p_future": -m}


def test_frame_starts_match_reference_loop():
    frame = 1120
    for n in (0, 1119, 1120, 1121, 1920, 5000, 16000 * 3 + 17):
        want = [i for i in range(0, n, frame - 320) if i + frame <= n]      # vap_offline.py:51-54
        assert list(offline.frame_starts(n, frame)) == want


def test_ragged_dialogues_and_timestamps(tmp_path):
    vap = FakeVap()
    a = np.arange(1120 + 800 * 3, dtype=np.float32) / 1e4
    b = np.arange(1120 + 800 * 1, dtype=np.float32) / 1e4
    res = offline.run_offline(vap, [(a, a + 1), (b, b + 2)])
    assert [len(r) for r in res] == [4, 2]
    assert [c[1] for c in vap.calls] == [[0, 1], [0, 1], [0], [0]]
    np.testing.assert_array_equal(vap.calls[1][0][1, 0], b[800:800 + 1120])
    assert res[0][0]["t"] == 1120 / 16000 and res[0][3]["t"] == (2400 + 1120) / 16000   # 0.07 s, like README.md:240
    p = tmp_path / "o.txt"
    offline.write_csv(str(p), res[1])
    lines = p.read_text().splitlines()
    assert lines[0] == "time_sec,p_now(0=left),p_now(1=right),p_future(0=left),p_future(1=right)"
    assert lines[1].split(",")[0] == "0.07" and len(lines) == 3


def test_read_wav_int16(tmp_path):
    x = (np.sin(np.arange(1600) / 10) * 12000).astype("<i2")
    p = tmp_path / "a.wav"
    with wave.open(str(p), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000); w.writeframes(x.tobytes())
    y = offline.read_wav_mono(str(p))
    np.testing.assert_allclose(y, x.astype(np.float32) / 32768.0)


@pytest.mark.gpu
def test_offline_driver_matches_reference_offline_golden():
    from golden_util import Case
    from vap_realtime_amd.realtime import ManyStreamVAP
    c = Case("offline20")
    vap = ManyStreamVAP(c.cpc_sd, c.vap_sd, c.frame_hz, c.ctx_sec, n_streams=2)
    n = c.hop * c.n_frames + 320
    res = offline.run_offline(vap, [(c.audio[0, 0, :n], c.audio[0, 1, :n]), (c.audio[0, 0, :n // 2], c.audio[0, 1, :n // 2])])
    assert len(res[0]) == c.n_frames
    for f, r in enumerate(res[0]):
        np.testing.assert_allclose(r["p_now"], c.z["p_now"][f][0], rtol=0, atol=1e-4)
        np.testing.assert_allclose(r["p_future"], c.z["p_future"][f][0], rtol=0, atol=1e-4)
    for f, r in enumerate(res[1]):     # the shorter copy of the same dialogue gives the same prefix
        np.testing.assert_allclose(r["p_now"], res[0][f]["p_now"], rtol=0, atol=1e-5)


@pytest.mark.gpu
def test_offline_poisoned_recording_fails_by_default_and_follows_the_reference_on_request():
    """A NaN sample in one recording: the reference's vap_offline.py keeps writing nan rows for that file (no check, vap_offline.py:62-73);
    here that is `on_numeric="reference"` (nan rows for the poisoned dialogue only, the other dialogue bit-identical to a clean run), and
    the default fails loudly."""
    from golden_util import Case
    from vap_realtime_amd.engine import VapxError
    from vap_realtime_amd.realtime import ManyStreamVAP
    c = Case("offline20")
    n = c.hop * 12 + 320
    clean = (c.audio[0, 0, :n].copy(), c.audio[0, 1, :n].copy())
    bad = (clean[0].copy(), clean[1].copy())
    bad[0][c.hop * 5 + 400] = np.nan                                   # inside frame 5's new samples
    vap = ManyStreamVAP(c.cpc_sd, c.vap_sd, c.frame_hz, c.ctx_sec, n_streams=2)
    ref = offline.run_offline(vap, [clean, clean])
    for s in (0, 1):
        vap.reset(s)
    with pytest.raises(VapxError):
        offline.run_offline(vap, [clean, bad])
    for s in (0, 1):
        vap.reset(s)
    res = offline.run_offline(vap, [clean, bad], on_numeric="reference")
    assert [r["p_now"] for r in res[0]] == [r["p_now"] for r in ref[0]]            # the healthy dialogue is untouched
    assert all(np.isfinite(r["p_now"]).all() for r in res[1][:5])
    assert all(np.isnan(r["p_now"]).all() and np.isnan(r["p_future"]).all() for r in res[1][5:]) and len(res[1]) == 12
