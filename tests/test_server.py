"""Host logic of the many-stream TCP front-end, with a test double in place of the GPU model
(the real model has no CPU path).  Checks routing, frame assembly and packet framing."""
import os
import socket
import struct
import time

import numpy as np

from vap_realtime_amd import wire
from vap_realtime_amd.server import ManyStreamServer


class FakeVap:
    """Deterministic stand-in: p_now = [mean|x1|, mean|x2|] of the frame, so routing errors show."""
    mode = "vap"

    def __init__(self, n_streams, hop):
        self.n_streams, self.hop = n_streams, hop
        self.calls, self.resets = [], []

    def process(self, frames, ids, on_numeric="raise"):
        self.calls.append((frames.shape, list(ids)))
        m = np.abs(frames).mean(axis=2)
        return {"p_now": m, "p_future": m[:, ::-1], "vad": (m > 0.5).astype(np.float32)}

    def reset(self, sid):
        self.resets.append(sid)


def _recv_exact(sock, n):
    b = b""
    while len(b) < n:
        chunk = sock.recv(n - len(b))
        assert chunk, "socket closed"
        b += chunk
    return b


def test_two_streams_routed_and_framed():
    hop = 800
    vap = FakeVap(4, hop)
    srv = ManyStreamServer(vap, port_in=0, port_out=0, max_wait_s=0.5).start()
    try:
        ins = [socket.create_connection(("127.0.0.1", srv.port_in)) for _ in range(2)]
        time.sleep(0.1)
        outs = [socket.create_connection(("127.0.0.1", srv.port_out)) for _ in range(2)]
        time.sleep(0.1)
        rng = np.random.default_rng(5)
        x = rng.standard_normal((2, 2, 2 * hop)) * np.array([0.1, 2.0])[:, None, None]
        for f in range(2):
            for p in range(5):
                for s in range(2):
                    seg = slice(f * hop + p * 160, f * hop + (p + 1) * 160)
                    ins[s].sendall(wire.encode_input(x[s, 0, seg], x[s, 1, seg]))
            for s in range(2):
                outs[s].settimeout(5)
                ln = struct.unpack("<I", _recv_exact(outs[s], 4))[0]
                assert ln == 12876
                r = wire.decode_result(_recv_exact(outs[s], ln))
                np.testing.assert_array_equal(r["x1"], x[s, 0, f * hop:(f + 1) * hop])     # float64 echo
                want = np.abs(x[s, :, f * hop:(f + 1) * hop].astype(np.float32)).mean(axis=1)
                np.testing.assert_allclose(r["p_now"], want, rtol=1e-6)
                np.testing.assert_allclose(r["p_future"], want[::-1], rtol=1e-6)
        assert vap.resets == [0, 1]
        assert all(ids == [0, 1] for _, ids in vap.calls) and len(vap.calls) == 2
    finally:
        srv.stop()


def test_ragged_tick_when_one_stream_lags():
    hop = 320  # 50 Hz
    vap = FakeVap(2, hop)
    srv = ManyStreamServer(vap, port_in=0, port_out=0, max_wait_s=0.05).start()
    try:
        a = socket.create_connection(("127.0.0.1", srv.port_in))
        b = socket.create_connection(("127.0.0.1", srv.port_in))
        time.sleep(0.1)
        z = np.zeros(160)
        for _ in range(2):
            a.sendall(wire.encode_input(z, z))
        b.sendall(wire.encode_input(z, z))          # b has only half a frame
        deadline = time.time() + 3
        while not vap.calls and time.time() < deadline:
            time.sleep(0.01)
        assert vap.calls and vap.calls[0][1] == [0]  # only stream 0 was stepped
        b.sendall(wire.encode_input(z, z))
        while len(vap.calls) < 2 and time.time() < deadline:
            time.sleep(0.01)
        assert vap.calls[1][1] == [1]
    finally:
        srv.stop()


def test_single_stream_broadcasts_like_reference():
    vap = FakeVap(1, 800)
    srv = ManyStreamServer(vap, port_in=0, port_out=0).start()
    try:
        i = socket.create_connection(("127.0.0.1", srv.port_in))
        time.sleep(0.05)
        outs = [socket.create_connection(("127.0.0.1", srv.port_out)) for _ in range(3)]
        time.sleep(0.1)
        z = np.ones(160)
        for _ in range(5):
            i.sendall(wire.encode_input(z, z))
        for o in outs:
            o.settimeout(5)
            ln = struct.unpack("<I", _recv_exact(o, 4))[0]
            r = wire.decode_result(_recv_exact(o, ln))
            assert r["p_now"] == [1.0, 1.0]
    finally:
        srv.stop()


def test_connection_burst_larger_than_the_default_backlog():
    """A whole shard of clients connecting at once (more than the classic listen(128) backlog): every stream must get its
    slot and its result route; one closed-loop round over all of them completes."""
    import threading
    S, hop = 400, 800
    vap = FakeVap(S, hop)
    srv = ManyStreamServer(vap, port_in=0, port_out=0, max_wait_s=0.01).start()
    ins, outs, errs = [None] * S, [None] * S, []

    def connect(lo, hi, port, dst):
        try:
            for i in range(lo, hi):
                dst[i] = socket.create_connection(("127.0.0.1", port), timeout=10)
        except Exception as e:          # noqa: BLE001
            errs.append(e)

    def burst(port, dst):
        th = [threading.Thread(target=connect, args=(k * 50, (k + 1) * 50, port, dst)) for k in range(S // 50)]
        for t in th:
            t.start()
        for t in th:
            t.join()

    try:
        burst(srv.port_in, ins)
        deadline = time.time() + 10
        while len(srv.in_conn) < S and time.time() < deadline:
            time.sleep(0.01)
        burst(srv.port_out, outs)
        deadline = time.time() + 10
        while sum(len(l) for l in srv.out_conns) < S and time.time() < deadline:
            time.sleep(0.01)
        assert not errs and len(srv.in_conn) == S and all(len(l) == 1 for l in srv.out_conns)
        frame = wire.encode_input(np.full(hop, 0.25), np.full(hop, -0.5))
        for s in ins:
            s.sendall(frame)
        for s in outs:
            s.settimeout(10)
            n = struct.unpack("<I", _recv_exact(s, 4))[0]
            r = wire.decode_result(_recv_exact(s, n))
            assert abs(r["p_now"][0] - 0.25) < 1e-6 and abs(r["p_now"][1] - 0.5) < 1e-6
    finally:
        for s in ins + outs:
            if s is not None:
                s.close()
        srv.stop()


class FakeAuxVap:
    """bc / nod stand-in returning what ManyStreamVAP.process returns (engine.split_outputs keys): constant head
    values taken from the wire golden, so the packet bytes can be compared with the reference codec's output."""

    def __init__(self, mode, z, hop=800):
        self.mode, self.n_streams, self.hop, self.z = mode, 2, hop, z
        self.resets = []

    def process(self, frames, ids, on_numeric="raise"):
        R = len(ids)
        aux = np.zeros((R, 4), np.float32)
        logits = np.zeros((R, 256), np.float32)
        if self.mode == "bc":
            aux[:, 1], aux[:, 2] = self.z["bc.p_bc_react"][0], self.z["bc.p_bc_emo"][0]
        else:
            aux[:, 1], aux[:, 2], aux[:, 3] = self.z["nod.p_nod_short"][0], self.z["nod.p_nod_long"][0], self.z["nod.p_nod_long_p"][0]
            logits[:, :50] = self.z["nod.p_bc"]
        return {"aux": aux, "logits": logits, "n": np.full(R, 50, np.int32), "status": np.zeros(R, np.int32)}

    def reset(self, sid):
        self.resets.append(sid)


def _f32_exact(a):
    """The goldens' head values as float32-representable doubles (the engine's outputs are float32)."""
    return np.asarray(a, np.float32).astype(np.float64)


def test_bc_and_nod_result_packets_carry_the_reference_bytes():
    """bc: u32 1 | p_bc_react | u32 1 | p_bc_emo (util.py:193-211); nod: u32 n | p_bc of EVERY window row | three u32 1
    blocks (util.py:213-237, vap_nod_main.py:276,398-406).  The packet body after the time stamp must equal what the
    reference codec produces for the same values."""
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "wire.npz"))
    for mode in ("bc", "nod"):
        vap = FakeAuxVap(mode, z)
        srv = ManyStreamServer(vap, port_in=0, port_out=0, max_wait_s=0.5).start()
        try:
            i = socket.create_connection(("127.0.0.1", srv.port_in))
            time.sleep(0.05)
            o = socket.create_connection(("127.0.0.1", srv.port_out))
            time.sleep(0.1)
            i.sendall(wire.encode_input(z["vap.x1"], z["vap.x2"]))
            o.settimeout(5)
            ln = struct.unpack("<I", _recv_exact(o, 4))[0]
            got = _recv_exact(o, ln)
            res = {"t": 0.0, "x1": z["vap.x1"], "x2": z["vap.x2"]}
            keys = ("p_bc_react", "p_bc_emo") if mode == "bc" else ("p_bc", "p_nod_short", "p_nod_long", "p_nod_long_p")
            for k in keys:
                res[k] = _f32_exact(z[f"{mode}.{k}"])
            want = wire.encode_result(res, mode)
            assert ln == len(want) == len(bytes(z[f"{mode}.bytes"]))
            assert got[8:] == want[8:]
            r = wire.decode_result(got, mode)
            if mode == "nod":
                assert len(r["p_bc"]) == 50
        finally:
            srv.stop()


def test_non_finite_stream_is_reset_and_skipped_while_the_others_are_served():
    hop = 800

    class Poisoned(FakeVap):
        def process(self, frames, ids, on_numeric="raise"):
            r = super().process(frames, ids)
            r["status"] = np.array([1 if s == 1 else 0 for s in ids], np.int32)
            return r

    vap = Poisoned(2, hop)
    srv = ManyStreamServer(vap, port_in=0, port_out=0, max_wait_s=0.5, reset_on_connect=False).start()
    try:
        ins = [socket.create_connection(("127.0.0.1", srv.port_in)) for _ in range(2)]
        time.sleep(0.1)
        outs = [socket.create_connection(("127.0.0.1", srv.port_out)) for _ in range(2)]
        time.sleep(0.1)
        for f in range(2):
            for s in range(2):
                ins[s].sendall(wire.encode_input(np.full(hop, 0.5), np.full(hop, 0.25)))
            outs[0].settimeout(5)
            ln = struct.unpack("<I", _recv_exact(outs[0], 4))[0]
            r = wire.decode_result(_recv_exact(outs[0], ln))
            assert r["p_now"] == [0.5, 0.25]
        outs[1].settimeout(0.3)
        try:
            data = outs[1].recv(4)
        except socket.timeout:
            data = b""
        assert data == b""                                   # the poisoned stream got no packet ...
        assert vap.resets == [1, 1] and srv.numeric_resets == 2   # ... and was reset each time; the server kept running
    finally:
        srv.stop()


def test_process_contract_is_decided_once_and_an_inner_type_error_is_not_retried():
    """Round 4 called process(.., on_numeric="status") inside `try: ... except TypeError: process(frames, ids)`: a TypeError raised INSIDE a
    model that does take on_numeric was swallowed and the model stepped a second time on the same frames (advisor r04).  The contract now
    comes from the signature, once, and an inner TypeError surfaces after exactly one call."""
    class TwoArg(FakeVap):
        def process(self, frames, ids):
            return super().process(frames, ids)

    class Broken(FakeVap):
        def process(self, frames, ids, on_numeric="raise"):
            self.calls.append("x")
            raise TypeError("a dtype bug inside the model")
    hop = 800
    for cls, want in ((FakeVap, True), (TwoArg, False), (Broken, True)):
        srv = ManyStreamServer(cls(1, hop), port_in=0, port_out=0)
        assert srv._status_contract is want
        srv.lin.close(); srv.lout.close()
    vap = Broken(1, hop)
    srv = ManyStreamServer(vap, port_in=0, port_out=0, max_wait_s=0.0)
    srv.in_conn[0] = None
    srv.asm.push(0, wire.encode_input(np.zeros(hop), np.zeros(hop)))
    srv.first_ready_t = time.time() - 1.0
    try:
        srv._maybe_tick()
        raise AssertionError("the model's TypeError was swallowed")
    except TypeError as e:
        assert "dtype bug" in str(e)
    assert vap.calls == ["x"]                 # stepped once, not twice
    srv.lin.close(); srv.lout.close()


def test_front_door_process_over_worker_processes_with_standin_engines():
    """The host half of BASELINE config 4 in miniature, across REAL processes: a front-door process (vapx_frontdoor_open_links) passes every
    accepted connection to one of two worker processes (vapx_ingest_attach_link), each a passive native front-end over the native stand-in for
    the GPU tick; two load-generator processes play 64 real-time dialogues with in-band latency stamps.  Every frame must come back, every output
    socket must keep hearing one dialogue, both workers must carry half the dialogues."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "server_load.py"), "--standin", "--shards", "2", "--worker-procs", "--streams", "64",
                          "--seconds", "3", "--warm", "1.5", "--loadgen-procs", "2", "--client-threads", "2", "--rx-threads", "2", "--tx-threads", "2"],
                         capture_output=True, text=True, timeout=180, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["streams"] == 64 and d["procs"] == 2 and d["inband"] == 1
    assert d["frames_sent"] > 64 * 20 * 3 and d["frames_answered"] == d["frames_sent"]
    assert d["route_changes"] == 0 and d["inband_unreadable"] == 0
    assert d["server_stats"]["front_door"] == {"accepted_in": 64, "accepted_out": 64, "refused": 0}
    per = d["server_stats"]["per_shard"]
    assert len(per) == 2 and all(abs(p["frames_done"] - d["frames_answered"] / 2) <= 64 for p in per)      # dialogue k -> worker k mod 2
    assert "worker processes" in d["server"]
