"""Drop-in surfaces on the GPU: VAPRealTime (level 2), the model-attribute surface (level 1) driven
by an orchestration that reads like the reference's process_vap, and the TCP front-end end to end."""
import socket
import struct
import time

import numpy as np
import pytest

from golden_util import Case

pytestmark = pytest.mark.gpu
TOL = 1e-4


def test_vaprealtime_dropin_matches_golden():
    """Same constructor / process_vap / result attributes as rvap/vap_main/vap_main.py:185-335, fed
    the way proc_serv_in feeds it (float64 carry + new samples)."""
    import torch
    from vap_realtime_amd.realtime import VAPRealTime
    c = Case("vap20")
    rt = VAPRealTime(c.vap_sd, {"weights": c.cpc_sd}, torch.device("cuda", 0), c.frame_hz, c.ctx_sec)
    assert rt.audio_frame_size == 1120 and rt.frame_contxt_padding == 320 and rt.audio_context_len == 50
    cur1, cur2 = np.zeros(320), np.zeros(320)
    for f in range(c.n_frames):
        new = c.new_samples(f)[0].astype(np.float64)
        cur1, cur2 = np.concatenate([cur1, new[0]]), np.concatenate([cur2, new[1]])
        assert len(cur1) == rt.audio_frame_size
        before = rt.process_time_abs
        rt.process_vap(cur1 if f % 2 else cur1.tolist(), cur2 if f % 2 else cur2.tolist())   # list and ndarray inputs
        assert rt.process_time_abs != before
        np.testing.assert_allclose(rt.result_p_now, c.z["p_now"][f][0], rtol=0, atol=TOL)
        np.testing.assert_allclose(rt.result_p_future, c.z["p_future"][f][0], rtol=0, atol=TOL)
        np.testing.assert_allclose(rt.result_vad, c.z["vad"][f][0], rtol=0, atol=TOL)
        assert len(rt.current_x1_audio) == 800
        cur1, cur2 = cur1[-320:], cur2[-320:]
    r = rt.get_result()
    assert set(r) == {"t", "x1", "x2", "p_now", "p_future", "vad"}
    with pytest.raises(ValueError):
        rt.process_vap(np.zeros(100), np.zeros(100))


def test_level1_model_surface_under_reference_style_orchestration():
    """The calls process_vap makes on self.vap (vap_main.py:272-307), against the HIP-backed VapGPT."""
    import torch
    from vap_realtime_amd.realtime import VapGPT
    c = Case("vap20")
    vap = VapGPT(c.cpc_sd, c.vap_sd, c.frame_hz, c.ctx_sec).to("cuda").eval()
    e1_context, e2_context = [], []
    carry = np.zeros((2, 320), np.float32)
    for f in range(c.n_frames):
        buf = np.concatenate([carry, c.new_samples(f)[0]], axis=1)
        carry = buf[:, -320:]
        x1_ = torch.from_numpy(buf[0]).cuda().unsqueeze(0).unsqueeze(0)
        x2_ = torch.from_numpy(buf[1]).cuda().unsqueeze(0).unsqueeze(0)
        e1, e2 = vap.encode_audio(x1_, x2_)
        e1_context.append(e1); e2_context.append(e2)
        e1_context, e2_context = e1_context[-c.T:], e2_context[-c.T:]
        x1c, x2c = torch.cat(e1_context, dim=1), torch.cat(e2_context, dim=1)
        o1 = vap.ar_channel(x1c, attention=False)
        o2 = vap.ar_channel(x2c, attention=False)
        out = vap.ar(o1["x"], o2["x"], attention=False)
        logits = vap.vap_head(out["x"])
        vad1 = vap.va_classifier(o1["x"]).sigmoid()[:, -1]
        vad2 = vap.va_classifier(o2["x"]).sigmoid()[:, -1]
        probs = logits.softmax(dim=-1)
        p_now = vap.objective.probs_next_speaker_aggregate(probs, from_bin=0, to_bin=1)
        p_future = vap.objective.probs_next_speaker_aggregate(probs, from_bin=2, to_bin=3)
        np.testing.assert_allclose(logits[0, -1].cpu().numpy(), c.z["logits"][f][0], rtol=0, atol=TOL)
        np.testing.assert_allclose(p_now[0, -1].cpu().numpy(), c.z["p_now"][f][0], rtol=0, atol=TOL)
        np.testing.assert_allclose(p_future[0, -1].cpu().numpy(), c.z["p_future"][f][0], rtol=0, atol=TOL)
        np.testing.assert_allclose([float(vad1), float(vad2)], c.z["vad"][f][0], rtol=0, atol=TOL)
        if f"inter.f{f}.rows" in c.z.files:
            rows = c.z[f"inter.f{f}.rows"]
            np.testing.assert_allclose(out["x"][0].cpu().numpy()[rows], c.z[f"inter.f{f}.comb"], rtol=0, atol=TOL)
            np.testing.assert_allclose(out["x1"][0].cpu().numpy()[rows], c.z[f"inter.f{f}.stereo2"][0], rtol=3e-5, atol=1e-3)


def test_tcp_front_end_end_to_end_on_gpu():
    """2560-byte packets in, length-prefixed result packets out, three dialogues on one engine."""
    from vap_realtime_amd import wire
    from vap_realtime_amd.realtime import ManyStreamVAP
    from vap_realtime_amd.server import ManyStreamServer
    c = Case("multi3")
    vap = ManyStreamVAP(c.cpc_sd, c.vap_sd, c.frame_hz, c.ctx_sec, n_streams=3)
    srv = ManyStreamServer(vap, port_in=0, port_out=0, max_wait_s=1.0).start()
    try:
        ins = []
        for _ in range(3):
            ins.append(socket.create_connection(("127.0.0.1", srv.port_in)))
            time.sleep(0.05)
        outs = []
        for _ in range(3):
            outs.append(socket.create_connection(("127.0.0.1", srv.port_out)))
            time.sleep(0.05)
        for f in range(6):
            new = c.new_samples(f).astype(np.float64)
            for p in range(5):
                for s in range(3):
                    ins[s].sendall(wire.encode_input(new[s, 0, p * 160:(p + 1) * 160], new[s, 1, p * 160:(p + 1) * 160]))
            for s in range(3):
                outs[s].settimeout(20)
                hdr = b""
                while len(hdr) < 4:
                    hdr += outs[s].recv(4 - len(hdr))
                ln = struct.unpack("<I", hdr)[0]
                assert ln == 12876
                payload = b""
                while len(payload) < ln:
                    payload += outs[s].recv(ln - len(payload))
                r = wire.decode_result(payload)
                np.testing.assert_allclose(r["p_now"], c.z["p_now"][f][s], rtol=0, atol=TOL)
                np.testing.assert_allclose(r["p_future"], c.z["p_future"][f][s], rtol=0, atol=TOL)
                np.testing.assert_allclose(r["vad"], c.z["vad"][f][s], rtol=0, atol=TOL)
                np.testing.assert_array_equal(r["x1"], new[s, 0])
    finally:
        srv.stop()
