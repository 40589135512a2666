"""Native front-end (libvapx vapx_ingest_* / vapx_wire_*): codec bytes == the reference's rvap/common/util.py output (goldens of
the imported reference, tests/golden/wire.npz), and the socket plumbing — routing, ragged ticks, segmentation, back-pressure,
broadcast, non-finite streams — over a Python step function (the real model has no CPU path)."""
import ctypes as C
import os
import socket
import struct
import threading
import time

import numpy as np
import pytest

from vap_realtime_amd import engine, ingest, wire

Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "wire.npz"))


def _row(mode):
    row = np.zeros(engine.OUT_STRIDE, np.float32)
    if mode == "vap":
        row[0:2], row[2:4], row[4:6] = Z["vap.p_now"], Z["vap.p_future"], Z["vap.vad"]
    elif mode == "bc":
        row[engine.OUT_AUX + 1], row[engine.OUT_AUX + 2] = Z["bc.p_bc_react"][0], Z["bc.p_bc_emo"][0]
    else:
        row[engine.OUT_NVALID] = 50
        row[engine.OUT_LOGITS:engine.OUT_LOGITS + 50] = Z["nod.p_bc"]
        row[engine.OUT_AUX + 1], row[engine.OUT_AUX + 2], row[engine.OUT_AUX + 3] = Z["nod.p_nod_short"][0], Z["nod.p_nod_long"][0], Z["nod.p_nod_long_p"][0]
    return row


def test_native_input_decode_equals_reference_codec():
    b = bytes(Z["in.bytes"])
    x1, x2, f1, f2 = ingest.decode_input(b)
    np.testing.assert_array_equal(x1, Z["in.x1"])
    np.testing.assert_array_equal(x2, Z["in.x2"])
    np.testing.assert_array_equal(f1, Z["in.x1"].astype(np.float32))        # the f64 -> f32 cast of vap_main.py:266-270
    g1, _, h1, _ = ingest.decode_input(b, gain=1.5)
    np.testing.assert_array_equal(g1, Z["in.x1"] * 1.5)                     # float64 multiply before the cast (:393-395)
    np.testing.assert_array_equal(h1, (Z["in.x1"] * 1.5).astype(np.float32))
    with pytest.raises(ValueError):
        ingest.decode_input(b"\0" * 17)


@pytest.mark.parametrize("mode", ["vap", "bc", "nod"])
def test_native_result_packet_bytes_equal_reference_codec(mode):
    """The golden bytes were produced by util.conv_vapresult_2_bytearray(/_bc/_nod) from float64 head values; the engine's
    heads are float32, so the comparison uses the float32-representable values of the same numbers on both sides."""
    keys = {"vap": ("p_now", "p_future", "vad"), "bc": ("p_bc_react", "p_bc_emo"), "nod": ("p_bc", "p_nod_short", "p_nod_long", "p_nod_long_p")}[mode]
    res = {"t": float(Z["vap.t"]), "x1": Z["vap.x1"], "x2": Z["vap.x2"]}
    for k in keys:
        res[k] = np.asarray(Z[f"{mode}.{k}"], np.float32).astype(np.float64)
    want = wire.frame_result(res, mode)                # wire.py is proven byte-identical to the reference in test_wire.py
    got = ingest.encode_result(mode, float(Z["vap.t"]), Z["vap.x1"], Z["vap.x2"], _row(mode))
    assert got == want
    assert len(got) == 4 + len(bytes(Z[f"{mode}.bytes"]))
    # where the golden's head values are exactly float32-representable the packet equals the REFERENCE's bytes outright
    ref = bytes(Z[f"{mode}.bytes"])
    n_echo = 8 + 2 * (4 + 8 * 800)
    assert got[4:4 + n_echo] == ref[:n_echo]


class Model:
    """p_now = mean |x| per channel of the frame, p_future reversed: routing errors show up in the numbers."""

    def __init__(self, poison=()):
        self.calls, self.resets, self.poison = [], [], set(poison)

    def step(self, ids, audio, out):
        self.calls.append(ids.tolist())
        m = np.abs(audio).mean(axis=2)
        out[:, 0:2] = m
        out[:, 2:4] = m[:, ::-1]
        out[:, 4:6] = (m > 0.5)
        for k, s in enumerate(ids):
            if int(s) in self.poison:
                out[k, engine.OUT_STATUS] = 1.0
        return 0

    def reset(self, sid):
        self.resets.append(sid)


def _recv_exact(sock, n):
    b = b""
    while len(b) < n:
        chunk = sock.recv(n - len(b))
        assert chunk, "socket closed"
        b += chunk
    return b


def _read_result(sock, mode="vap"):
    sock.settimeout(10)
    ln = struct.unpack("<I", _recv_exact(sock, 4))[0]
    return ln, wire.decode_result(_recv_exact(sock, ln), mode)


def _wait(cond, timeout=5.0):
    t0 = time.time()
    while not cond() and time.time() - t0 < timeout:
        time.sleep(0.005)
    assert cond()


def test_two_streams_routed_framed_and_segmentation_agnostic():
    hop = 800
    m = Model()
    srv = ingest.NativeServer.over_function(m.step, 4, 20, reset=m.reset, max_wait_s=0.5)
    try:
        ins = [socket.create_connection(("127.0.0.1", srv.port_in)) for _ in range(2)]
        _wait(lambda: srv.stats()["in_connections"] == 2)
        outs = [socket.create_connection(("127.0.0.1", srv.port_out)) for _ in range(2)]
        _wait(lambda: srv.stats()["out_connections"] == 2)
        rng = np.random.default_rng(5)
        x = rng.standard_normal((2, 2, 3 * hop)) * np.array([0.1, 2.0])[:, None, None]
        for f in range(3):
            for s in range(2):
                data = wire.encode_input(x[s, 0, f * hop:(f + 1) * hop], x[s, 1, f * hop:(f + 1) * hop])
                if f == 0:                                  # 2560-byte packets like the reference client
                    for p in range(5):
                        ins[s].sendall(data[p * 2560:(p + 1) * 2560])
                elif f == 1:                                # odd segmentation: sample pairs split across sends
                    for a, b in ((0, 7), (7, 5003), (5003, 5004), (5004, len(data))):
                        ins[s].sendall(data[a:b])
                        time.sleep(0.002)
                else:
                    ins[s].sendall(data)                    # the whole frame at once
            for s in range(2):
                ln, r = _read_result(outs[s])
                assert ln == 12876
                np.testing.assert_array_equal(r["x1"], x[s, 0, f * hop:(f + 1) * hop])     # float64 echo, bit-exact
                np.testing.assert_array_equal(r["x2"], x[s, 1, f * hop:(f + 1) * hop])
                want = np.abs(x[s, :, f * hop:(f + 1) * hop].astype(np.float32)).mean(axis=1)
                np.testing.assert_allclose(r["p_now"], want, rtol=1e-6)
                np.testing.assert_allclose(r["p_future"], want[::-1], rtol=1e-6)
                assert abs(r["t"] - time.time()) < 30
        assert sorted(m.resets) == [0, 1]
        assert all(sorted(c) == [0, 1] for c in m.calls) and len(m.calls) == 3
        st = srv.stats()
        assert st["frames_done"] == 6 and st["ticks"] == 3 and st["mean_batch"] == 2.0 and st["lat_p99_ms"] > 0
    finally:
        srv.close()


def test_ragged_tick_when_one_stream_lags_and_carry_only_reset():
    m = Model()
    srv = ingest.NativeServer.over_function(m.step, 2, 50, reset=m.reset, max_wait_s=0.05, reset_on_connect=False)
    try:
        a = socket.create_connection(("127.0.0.1", srv.port_in))
        b = socket.create_connection(("127.0.0.1", srv.port_in))
        _wait(lambda: srv.stats()["in_connections"] == 2)
        z = np.zeros(160)
        for _ in range(2):
            a.sendall(wire.encode_input(z, z))
        b.sendall(wire.encode_input(z, z))              # b has only half a 50 Hz frame
        _wait(lambda: len(m.calls) >= 1)
        assert m.calls[0] == [0]                        # only stream 0 was stepped
        b.sendall(wire.encode_input(z, z))
        _wait(lambda: len(m.calls) >= 2)
        assert m.calls[1] == [1]
        assert sorted(m.resets) == [-2, -1]             # reset_on_connect=False: carry-only resets, like vap_main.py:368-369
    finally:
        srv.close()


def test_single_stream_broadcasts_like_the_reference():
    m = Model()
    srv = ingest.NativeServer.over_function(m.step, 1, 20)
    try:
        i = socket.create_connection(("127.0.0.1", srv.port_in))
        outs = [socket.create_connection(("127.0.0.1", srv.port_out)) for _ in range(3)]
        _wait(lambda: srv.stats()["out_connections"] == 3 and srv.stats()["in_connections"] == 1)
        z = np.ones(800)
        i.sendall(wire.encode_input(z, z))
        for o in outs:
            _, r = _read_result(o)
            assert r["p_now"] == [1.0, 1.0]
    finally:
        srv.close()


def test_non_finite_stream_is_reset_and_skipped_the_others_are_served():
    m = Model(poison=[1])
    srv = ingest.NativeServer.over_function(m.step, 2, 20, reset=m.reset, max_wait_s=0.5, reset_on_connect=False)
    try:
        ins = [socket.create_connection(("127.0.0.1", srv.port_in)) for _ in range(2)]
        _wait(lambda: srv.stats()["in_connections"] == 2)
        outs = [socket.create_connection(("127.0.0.1", srv.port_out)) for _ in range(2)]
        _wait(lambda: srv.stats()["out_connections"] == 2)
        for f in range(2):
            for s in range(2):
                ins[s].sendall(wire.encode_input(np.full(800, 0.5), np.full(800, 0.25)))
            _, r = _read_result(outs[0])
            assert r["p_now"] == [0.5, 0.25]
        outs[1].settimeout(0.3)
        with pytest.raises(socket.timeout):
            outs[1].recv(4)
        assert [r for r in m.resets if r >= 0] == [1, 1] and srv.stats()["numeric_resets"] == 2
    finally:
        srv.close()


def test_a_sender_that_outruns_the_engine_is_paused_not_dropped():
    """Ten frames pushed at once into a model that takes 30 ms per tick: the connection is back-pressured (at most three
    frames buffered), every frame is answered, in order."""
    hop = 800

    class Slow(Model):
        def step(self, ids, audio, out):
            time.sleep(0.03)
            return super().step(ids, audio, out)

    m = Slow()
    srv = ingest.NativeServer.over_function(m.step, 1, 20, max_wait_s=0.001)
    try:
        i = socket.create_connection(("127.0.0.1", srv.port_in))
        o = socket.create_connection(("127.0.0.1", srv.port_out))
        _wait(lambda: srv.stats()["out_connections"] == 1 and srv.stats()["in_connections"] == 1)
        level = [0.1 * (k + 1) for k in range(10)]
        blob = b"".join(wire.encode_input(np.full(hop, v), np.full(hop, v)) for v in level)
        threading.Thread(target=i.sendall, args=(blob,), daemon=True).start()
        for v in level:
            _, r = _read_result(o)
            np.testing.assert_allclose(r["p_now"], [np.float32(v)] * 2, rtol=1e-6)
        assert srv.stats()["overruns"] >= 1 and srv.stats()["frames_done"] == 10
    finally:
        srv.close()


def test_a_partial_frame_after_an_overrun_pause_is_still_delivered():
    """Round-5 advisor finding: SO_RCVLOWAT stayed at its pre-pause value across pause -> resume, so the tail of the frame that was open
    when the connection resumed could sit unread in the socket.  Deterministic form: 1 byte (arms a whole-frame threshold), then a burst
    of nine frames + 4000 bytes into a slow model (overrun -> pause), then the missing 1120 bytes — fewer than the stale threshold.  All
    ten frames must be answered."""
    hop = 800

    class Slow(Model):
        def step(self, ids, audio, out):
            time.sleep(0.03)
            return super().step(ids, audio, out)

    for _ in range(3):
        m = Slow()
        srv = ingest.NativeServer.over_function(m.step, 1, 20, max_wait_s=0.001)
        try:
            i = socket.create_connection(("127.0.0.1", srv.port_in))
            i.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            o = socket.create_connection(("127.0.0.1", srv.port_out))
            _wait(lambda: srv.stats()["out_connections"] == 1 and srv.stats()["in_connections"] == 1)
            level = [0.1 * (k + 1) for k in range(10)]
            blob = b"".join(wire.encode_input(np.full(hop, v), np.full(hop, v)) for v in level)
            frame = len(blob) // 10
            i.sendall(blob[:1])
            time.sleep(0.05)
            i.sendall(blob[1:9 * frame + 4000])
            _wait(lambda: srv.stats()["overruns"] >= 1)
            time.sleep(0.05)
            i.sendall(blob[9 * frame + 4000:])
            for v in level:
                _, r = _read_result(o)
                np.testing.assert_allclose(r["p_now"], [np.float32(v)] * 2, rtol=1e-6)
            assert srv.stats()["frames_done"] == 10
        finally:
            srv.close()


def test_connection_burst_and_reconnect_reuse_slots():
    S, hop = 300, 800
    m = Model()
    srv = ingest.NativeServer.over_function(m.step, S, 20, reset=m.reset, max_wait_s=0.02)
    ins, outs = [], []
    try:
        for _ in range(S):
            ins.append(socket.create_connection(("127.0.0.1", srv.port_in), timeout=10))
        _wait(lambda: srv.stats()["in_connections"] == S, 10)
        for _ in range(S):
            outs.append(socket.create_connection(("127.0.0.1", srv.port_out), timeout=10))
        _wait(lambda: srv.stats()["out_connections"] == S, 10)
        for k, s in enumerate(ins):
            s.sendall(wire.encode_input(np.full(hop, 0.001 * (k + 1)), np.full(hop, -0.5)))
        for k, s in enumerate(outs):                      # the k-th output connection hears the k-th stream
            _, r = _read_result(s)
            assert abs(r["p_now"][0] - 0.001 * (k + 1)) < 1e-6 and abs(r["p_now"][1] - 0.5) < 1e-6
        ins[7].close()                                    # a dialogue ends, a new client takes the lowest free slot (7)
        _wait(lambda: srv.stats()["in_connections"] == S - 1)
        n = socket.create_connection(("127.0.0.1", srv.port_in))
        ins[7] = n
        _wait(lambda: srv.stats()["in_connections"] == S)
        n.sendall(wire.encode_input(np.full(hop, 0.9), np.full(hop, 0.9)))
        _, r = _read_result(outs[7])
        assert abs(r["p_now"][0] - 0.9) < 1e-6
        assert m.resets.count(7) == 2
    finally:
        for s in ins + outs:
            s.close()
        srv.close()


def test_random_segmentation_and_reconnects_keep_every_stream_exact():
    """Property test of the frame assembler: 24 streams send their audio cut at random byte positions (sample pairs split across
    sends, several frames in one send), some hang up mid-frame and a new client takes the slot — every answered frame must echo
    exactly the samples of ONE frame of ONE connection, in order, and carry that frame's numbers."""
    hop, S, F_ = 320, 24, 7            # 50 Hz frames: more frames per byte
    rng = np.random.default_rng(11)
    m = Model()
    srv = ingest.NativeServer.over_function(m.step, S, 50, reset=m.reset, max_wait_s=0.002)
    try:
        ins = [socket.create_connection(("127.0.0.1", srv.port_in)) for _ in range(S)]
        _wait(lambda: srv.stats()["in_connections"] == S)
        outs = [socket.create_connection(("127.0.0.1", srv.port_out)) for _ in range(S)]
        _wait(lambda: srv.stats()["out_connections"] == S)
        audio = rng.standard_normal((S, 2, hop * F_))
        blobs = [wire.encode_input(audio[s, 0], audio[s, 1]) for s in range(S)]
        pos = [0] * S
        expect = [list(range(F_)) for _ in range(S)]           # frame indices each listener must see, in order
        quitter = {5: 3 * hop * 16 + 100, 17: 1 * hop * 16 + 7}  # these hang up mid-frame (after 3 / 1 complete frames)
        live = set(range(S))
        while live:
            for s in list(live):
                limit = quitter.get(s, len(blobs[s]))
                n = int(rng.choice([1, 7, 16, 333, 2560, 5000, 20000]))
                n = min(n, limit - pos[s])
                ins[s].sendall(blobs[s][pos[s]:pos[s] + n])
                pos[s] += n
                if pos[s] >= limit:
                    live.discard(s)
        for s, cut in quitter.items():
            expect[s] = list(range(cut // (hop * 16)))
        for s in range(S):
            for f in expect[s]:
                _, r = _read_result(outs[s])
                np.testing.assert_array_equal(r["x1"], audio[s, 0, f * hop:(f + 1) * hop])
                np.testing.assert_array_equal(r["x2"], audio[s, 1, f * hop:(f + 1) * hop])
                want = np.abs(audio[s, :, f * hop:(f + 1) * hop].astype(np.float32)).mean(axis=1)
                np.testing.assert_allclose(r["p_now"], want, rtol=1e-6)
        # the quitters' partial frames vanish with the connection; a new client on the slot starts on a frame boundary
        for s in quitter:
            ins[s].close()
        _wait(lambda: srv.stats()["in_connections"] == S - len(quitter))
        for s in sorted(quitter):
            ins[s] = socket.create_connection(("127.0.0.1", srv.port_in))       # lowest free slot first: 5, then 17
            _wait(lambda: srv.stats()["in_connections"] >= S - len(quitter) + 1 + sorted(quitter).index(s))
            fresh = rng.standard_normal((2, hop))
            ins[s].sendall(wire.encode_input(fresh[0], fresh[1]))
            _, r = _read_result(outs[s])
            np.testing.assert_array_equal(r["x1"], fresh[0])
        assert srv.stats()["frames_done"] == sum(len(e) for e in expect) + len(quitter)
    finally:
        for c in ins + outs:
            c.close()
        srv.close()


def test_many_streams_four_senders_every_stream_sees_its_frames_in_order():
    """Per-stream order with several sender threads (advisor r04): 384 dialogues, 4 senders, 4 receivers, ticks back to back (the step
    function is instant, so consecutive ticks overlap on their way out).  A stream's packets always leave through sender slot % 4, which walks
    over the ticks in publication order: every listener must see its own frames 0, 1, 2, ... with non-decreasing time stamps — and the old
    config struct (without the placement fields) must still be accepted."""
    hop, S, F_ = 800, 384, 12
    seen = np.zeros(S, np.int64)

    def step(ids, audio, out):
        out[:, 0] = audio[:, 0, 0]            # p_now[0] = the frame's first sample = its (stream, frame) tag
        out[:, 1] = ids
        return 0
    srv = ingest.NativeServer.over_function(step, S, 20, max_wait_s=0.001, min_batch=1, rx_threads=4, tx_threads=4)
    try:
        ins = [socket.create_connection(("127.0.0.1", srv.port_in)) for _ in range(S)]
        _wait(lambda: srv.stats()["in_connections"] == S, 20)
        outs = [socket.create_connection(("127.0.0.1", srv.port_out)) for _ in range(S)]
        _wait(lambda: srv.stats()["out_connections"] == S, 20)
        got = [[] for _ in range(S)]
        errs = []

        def reader(lo, hi):
            try:
                for _ in range(F_):
                    for k in range(lo, hi):
                        _, r = _read_result(outs[k])
                        got[k].append((r["t"], r["p_now"][0], int(r["p_now"][1])))
            except Exception as e:            # noqa: BLE001
                errs.append(e)
        readers = [threading.Thread(target=reader, args=(a, min(S, a + 96))) for a in range(0, S, 96)]
        for t in readers:
            t.start()
        for f in range(F_):                   # frames of all streams back to back: ticks pile up behind the senders
            for k in range(S):
                x = np.zeros((2, hop))
                x[0, 0] = f + 1
                ins[k].sendall(wire.encode_input(x[0], x[1]))
        for t in readers:
            t.join(60)
        assert not errs, errs
        slot_of = {}
        for k in range(S):
            assert [g[1] for g in got[k]] == [float(f + 1) for f in range(F_)], (k, got[k])      # in order, none lost, none doubled
            assert all(b[0] >= a[0] for a, b in zip(got[k], got[k][1:]))                        # time stamps never go back
            assert len({g[2] for g in got[k]}) == 1                                              # one dialogue per listener
            slot_of[k] = got[k][0][2]
        assert sorted(slot_of.values()) == list(range(S))
        st = srv.stats()
        assert st["frames_done"] == S * F_ and st["answered"] == S * F_ and 0 <= st["late_over_10ms"] <= S * F_
    finally:
        srv.close()
    # a caller built against the first ABI-2 header passes the config WITHOUT the placement fields: still accepted, nothing pinned
    lib = engine.load_library()
    old = (C.c_int32 * 14)()
    cfg = ingest.NativeServer._cfg(0, 0, 1.0, 0.002, 0, True, None, 0, 0, False)
    C.memmove(old, C.byref(cfg), 56)
    old[0] = 56
    assert ingest._IngestConfig.cpu_first.offset == 56
    h = C.c_void_p()
    keep = ingest._STEP_FN(lambda u, n, i, a, o: 0)
    assert lib.vapx_ingest_open_fn(C.cast(keep, C.c_void_p), None, None, 2, 2, 20, 0, C.byref(old), C.byref(h)) == 0
    lib.vapx_ingest_close(h)
    old[0] = 48                               # any other length is a configuration error
    assert lib.vapx_ingest_open_fn(C.cast(keep, C.c_void_p), None, None, 2, 2, 20, 0, C.byref(old), C.byref(h)) != 0


def test_pinned_front_end_threads_sit_on_the_configured_cores():
    """vapx_ingest_config.cpu_first / cpu_count: tick (+ accept) on the first core of the range, then one per receive thread, then one per
    sender.  Read back from /proc: every thread of the process that is pinned to exactly one core of the range."""
    avail = sorted(os.sched_getaffinity(0))
    if len(avail) < 4:
        pytest.skip("needs four cores")
    first = avail[0]
    count = 0
    while count < len(avail) and avail[count] == first + count:
        count += 1
    if count < 4:
        pytest.skip("needs four consecutive cores")
    count = min(count, 6)

    def single_core_threads():
        cores = []
        for tid in os.listdir("/proc/self/task"):
            try:
                m = os.sched_getaffinity(int(tid))
            except OSError:
                continue
            if len(m) == 1:
                cores.append(next(iter(m)))
        return sorted(cores)
    before = single_core_threads()
    lib = engine.load_library()
    cfg = ingest.NativeServer._cfg(0, 0, 1.0, 0.002, 0, True, None, 2, 2, False, 0.9, (first, count), True)
    keep = ingest._STEP_FN(lambda u, n, i, a, o: 0)
    h = C.c_void_p()
    assert lib.vapx_ingest_open_fn(C.cast(keep, C.c_void_p), None, None, 4, 4, 20, 0, C.byref(cfg), C.byref(h)) == 0
    try:
        new = single_core_threads()
        for c in before:
            new.remove(c)
        want = sorted([first, first] + [first + (1 + k) % count for k in range(2)] + [first + (3 + k) % count for k in range(2)])
        assert new == want, (new, want)
    finally:
        lib.vapx_ingest_close(h)


class TaggedModel(Model):
    """Model whose VAD columns carry the shard tag: which back-end answered is visible in every result packet."""

    def __init__(self, tag):
        super().__init__()
        self.tag = tag

    def step(self, ids, audio, out):
        rc = super().step(ids, audio, out)
        out[:, 4] = self.tag
        out[:, 5] = ids
        return rc


def _open_door(shards, kind):
    """The in-process front door, or the multi-process one (vapx_frontdoor_open_links) with its shards linked over AF_UNIX socket pairs — here
    inside one process: the protocol does not care where the two ends live (tests/test_server.py runs it across real processes)."""
    if kind == "in-process":
        return ingest.FrontDoor(shards, port_in=0, port_out=0), []
    pairs = [ingest.link_pair() for _ in shards]
    for s_, (_, w) in zip(shards, pairs):
        s_.attach_link(w.fileno())
    door = ingest.RemoteFrontDoor([d for d, _ in pairs], port_in=0, port_out=0)
    return door, pairs


def _close_door(door, shards, kind):
    if kind == "in-process":
        door.close()
    else:
        door.close()
        for s_ in shards:
            s_.close()


@pytest.mark.parametrize("kind", ["in-process", "links"])
def test_one_front_door_routes_dialogues_over_two_back_ends_and_reconnects_stick(kind):
    """vapx_frontdoor_*: ONE port pair (the reference's, vap_main.py:338-366) in front of two passive front-ends (= two GPUs).  Dialogue k
    takes the lowest free GLOBAL slot g = local * N + shard, i.e. shard k mod 2; the k-th output connection hears the k-th dialogue; a
    dialogue that drops and reconnects gets its old slot back (lowest free) — the shard that holds its state — and a full house refuses."""
    hop = 800
    models = [TaggedModel(10.0), TaggedModel(20.0)]
    shards = [ingest.NativeServer.over_function(m.step, 2, 20, reset=m.reset, max_wait_s=0.05, port_in=-1, port_out=-1) for m in models]
    assert all(s.port_in == 0 and s.port_out == 0 for s in shards)         # passive: listening on nothing
    door, _pairs = _open_door(shards, kind)
    try:
        ins, outs = [], []
        for k in range(4):                                                  # connect one by one: arrival order = dialogue index
            ins.append(socket.create_connection(("127.0.0.1", door.port_in)))
            _wait(lambda: door.counts()["accepted_in"] == k + 1)
        for k in range(4):
            outs.append(socket.create_connection(("127.0.0.1", door.port_out)))
            _wait(lambda: door.counts()["accepted_out"] == k + 1)
        assert [s.stats()["in_connections"] for s in shards] == [2, 2] and [s.stats()["out_connections"] for s in shards] == [2, 2]
        rng = np.random.default_rng(11)
        x = rng.standard_normal((4, 2, hop)) * np.array([0.1, 0.5, 1.0, 2.0])[:, None, None]

        def frame_all(which):
            for k in which:
                ins[k].sendall(wire.encode_input(x[k, 0], x[k, 1]))
            res = {}
            for k in which:
                _, r = _read_result(outs[k])
                np.testing.assert_array_equal(r["x1"], x[k, 0])             # the k-th output connection hears the k-th dialogue
                res[k] = r
            return res

        res = frame_all(range(4))
        for k in range(4):
            shard, local = ingest.FrontDoor.owner_of(k, 2)
            assert res[k]["vad"] == [models[shard].tag, float(local)], (k, res[k]["vad"])
            np.testing.assert_allclose(res[k]["p_now"], np.abs(x[k].astype(np.float32)).mean(axis=1), rtol=1e-6)
        # a fifth dialogue finds every slot of every shard taken
        extra = socket.create_connection(("127.0.0.1", door.port_in))
        _wait(lambda: door.counts()["refused"] == 1)
        extra.settimeout(5)
        assert extra.recv(1) == b""                                         # closed by the front door
        extra.close()
        # dialogue 1 (shard 1, local 0) drops and reconnects: lowest free global slot = its old one -> same back-end, same slot
        ins[1].close()
        _wait(lambda: shards[1].stats()["in_connections"] == 1)
        ins[1] = socket.create_connection(("127.0.0.1", door.port_in))
        _wait(lambda: shards[1].stats()["in_connections"] == 2)
        assert shards[0].stats()["in_connections"] == 2
        res = frame_all([1])
        assert res[1]["vad"] == [20.0, 0.0]
        assert sorted(models[1].resets) == [0, 0, 1] and sorted(models[0].resets) == [0, 1]     # reset_on_connect per (re)connection
        assert door.counts() == {"accepted_in": 5, "accepted_out": 4, "refused": 1}
    finally:
        _close_door(door, shards, kind)


def test_linked_door_follows_dropped_listeners_and_a_worker_that_goes_away():
    """The multi-process door decides on MIRRORS of its shards: a listener the worker dropped must free its place in the mirror (the next
    output connection goes to that dialogue), and a worker whose link closes gets no new dialogues while the others keep serving."""
    hop = 800
    models = [TaggedModel(10.0), TaggedModel(20.0)]
    shards = [ingest.NativeServer.over_function(m.step, 2, 20, reset=m.reset, max_wait_s=0.02, port_in=-1, port_out=-1) for m in models]
    door, pairs = _open_door(shards, "links")
    try:
        ins = []
        for k in range(2):
            ins.append(socket.create_connection(("127.0.0.1", door.port_in)))
            _wait(lambda: door.counts()["accepted_in"] == k + 1)
        outs = []
        for k in range(2):
            outs.append(socket.create_connection(("127.0.0.1", door.port_out)))
            _wait(lambda: door.counts()["accepted_out"] == k + 1)
        x = np.random.default_rng(5).standard_normal((2, 2, hop))
        # dialogue 0's listener goes away; the worker notices when it cannot deliver (the socket is reset), and tells the door
        outs[0].setsockopt(socket.SOL_SOCKET, socket.SO_LINGER, struct.pack("ii", 1, 0))
        outs[0].close()
        for _ in range(3):
            ins[0].sendall(wire.encode_input(x[0, 0], x[0, 1]))
            time.sleep(0.1)
        _wait(lambda: shards[0].stats()["out_connections"] == 0)
        time.sleep(0.1)
        late = socket.create_connection(("127.0.0.1", door.port_out))       # fewest listeners, lowest global slot: dialogue 0 again
        _wait(lambda: shards[0].stats()["out_connections"] == 1)
        ins[0].sendall(wire.encode_input(x[0, 0], x[0, 1]))
        _, r = _read_result(late)
        np.testing.assert_array_equal(r["x1"], x[0, 0])
        assert r["vad"][0] == 10.0
        # worker 0 goes away (its process died): dialogue slots 0 / 2 are gone, new dialogues land on worker 1 only
        shards[0].close()
        pairs[0][1].close()
        time.sleep(0.3)
        newc = socket.create_connection(("127.0.0.1", door.port_in))
        _wait(lambda: shards[1].stats()["in_connections"] == 2)
        ins[1].sendall(wire.encode_input(x[1, 0], x[1, 1]))
        _, r = _read_result(outs[1])
        assert r["vad"][0] == 20.0
        newc.close()
    finally:
        door.close()
        shards[1].close()


@pytest.mark.parametrize("kind", ["in-process", "links"])
def test_front_door_over_single_slot_shards_spreads_the_output_connections(kind):
    """serve.py's default is --streams 1: every per-GPU front-end is then in BROADCAST mode (all its output connections hear its one
    dialogue, as the reference's single-client server does).  Behind the front door the k-th output connection must still hear dialogue k,
    i.e. GPU k mod N — round 3 skipped broadcast shards in the routing loop and every listener landed on shard 0 (advisor finding)."""
    hop = 800
    models = [TaggedModel(10.0), TaggedModel(20.0)]
    shards = [ingest.NativeServer.over_function(m.step, 1, 20, reset=m.reset, max_wait_s=0.05, port_in=-1, port_out=-1) for m in models]
    door, _pairs = _open_door(shards, kind)
    try:
        ins, outs = [], []
        for k in range(2):
            ins.append(socket.create_connection(("127.0.0.1", door.port_in)))
            _wait(lambda: door.counts()["accepted_in"] == k + 1)
        for k in range(4):                                                  # two listeners per dialogue
            outs.append(socket.create_connection(("127.0.0.1", door.port_out)))
            _wait(lambda: door.counts()["accepted_out"] == k + 1)
        assert [s.stats()["out_connections"] for s in shards] == [2, 2]
        x = np.random.default_rng(3).standard_normal((2, 2, hop)) * np.array([0.2, 1.5])[:, None, None]
        for k in range(2):
            ins[k].sendall(wire.encode_input(x[k, 0], x[k, 1]))
        for j in range(4):                                                  # output connection j hears dialogue j mod 2 (GPU j mod 2)
            _, r = _read_result(outs[j])
            np.testing.assert_array_equal(r["x1"], x[j % 2, 0])
            assert r["vad"][0] == models[j % 2].tag
    finally:
        _close_door(door, shards, kind)
        for s_ in ins + outs:
            s_.close()


def test_a_half_passive_front_end_is_a_configuration_error():
    m = TaggedModel(1.0)
    for pin, pout in ((-1, 0), (0, -1)):
        with pytest.raises(Exception):
            ingest.NativeServer.over_function(m.step, 2, 20, port_in=pin, port_out=pout)
