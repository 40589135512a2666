"""Wire codec == the reference's rvap/common/util.py, byte for byte (goldens from the imported
reference, tools/make_golden_wire.py)."""
import os

import numpy as np
import pytest

from vap_realtime_amd import wire

Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "wire.npz"))


def test_input_packet_roundtrip():
    b = bytes(Z["in.bytes"])
    assert len(b) == wire.INPUT_PACKET_BYTES == 2560
    assert wire.encode_input(Z["in.x1"], Z["in.x2"]) == b
    x1, x2 = wire.decode_input(b)
    np.testing.assert_array_equal(x1, Z["in.x1"])
    np.testing.assert_array_equal(x2, Z["in.x2"])


@pytest.mark.parametrize("mode,keys", [("vap", ("p_now", "p_future", "vad")), ("bc", ("p_bc_react", "p_bc_emo")),
                                       ("nod", ("p_bc", "p_nod_short", "p_nod_long", "p_nod_long_p"))])
def test_result_packet_bytes(mode, keys):
    res = {"t": float(Z["vap.t"]), "x1": Z["vap.x1"], "x2": Z["vap.x2"]}
    for k in keys:
        res[k] = Z[f"{mode}.{k}"].tolist()
    want = bytes(Z[f"{mode}.bytes"])
    got = wire.encode_result(res, mode)
    assert got == want
    if mode == "vap":
        assert len(want) == 12876            # 20 Hz payload, 12 880 on the wire
        assert len(wire.frame_result(res)) == 12880
        assert wire.frame_result(res)[:4] == (12876).to_bytes(4, "little")
    back = wire.decode_result(want, mode)
    assert back["t"] == res["t"]
    for k in keys:
        np.testing.assert_array_equal(back[k], res[k])


def test_bad_lengths():
    with pytest.raises(ValueError):
        wire.decode_input(b"\0" * 17)
    with pytest.raises(ValueError):
        wire.encode_input([0.0] * 3, [0.0] * 4)


def test_packet_assembler_matches_server_framing():
    """Five 10 ms packets make one 20 Hz frame; float64 gain multiply happens before the f32 cast."""
    rng = np.random.default_rng(0)
    hop = 800
    pa = wire.PacketAssembler(3, hop, gain=1.5)
    x = rng.standard_normal((3, 2, hop))
    for s in (2, 0):
        for p in range(5):
            pkt = wire.encode_input(x[s, 0, p * 160:(p + 1) * 160], x[s, 1, p * 160:(p + 1) * 160])
            done = pa.push(s, pkt)
            assert done == (p == 4)
    ready = pa.ready()
    assert list(ready) == [0, 2]
    frames = pa.pop(ready)
    np.testing.assert_array_equal(frames[0], (x[0] * 1.5).astype(np.float32))
    np.testing.assert_array_equal(frames[1], (x[2] * 1.5).astype(np.float32))
    assert len(pa.ready()) == 0
    with pytest.raises(ValueError):
        for _ in range(6):
            pa.push(1, wire.encode_input(np.zeros(160), np.zeros(160)))


def test_batched_result_packets_equal_the_per_stream_encoder():
    rng = np.random.default_rng(3)
    R, n = 5, 800
    echo = rng.standard_normal((R, 2, n))
    heads = [rng.random((R, 2)).astype(np.float32) for _ in range(3)]
    t = 1727000000.123456
    buf = wire.frame_results_batch(t, echo, *heads)
    for k in range(R):
        want = wire.frame_result({"t": t, "x1": echo[k, 0], "x2": echo[k, 1], "p_now": heads[0][k], "p_future": heads[1][k],
                                  "vad": heads[2][k]})
        assert buf[k].tobytes() == want
