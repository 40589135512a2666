"""Vectorised TCP wire codec, byte-identical to the reference's ``rvap/common/util.py``.

Input packet (client -> :50007): 160 x { f64 ch1, f64 ch2 } little-endian = 2560 B per 10 ms
(util.py:52-62, 93-106; recv loop vap_main.py:373-391).  Output packet (:50008 -> clients), once
per VAP frame: ``u32 payload_len`` + ``f64 t | u32 n | n x f64 x1 | u32 n | n x f64 x2 | u32 2 |
p_now | u32 2 | p_future | u32 2 | vad`` (util.py:122-143; vap_main.py:446-448); bc and nod
variants per util.py:193-237.  The reference packs one ``struct.pack('<d')`` per sample in a Python
loop; here every array is one ``numpy`` buffer view, which is what lets a single host thread feed
thousands of streams (4096 streams x 32 kB/s = 1 GB/s of float64 samples).
"""
from __future__ import annotations

import struct
from typing import Dict, Sequence, Tuple

import numpy as np

SAMPLES_PER_PACKET = 160
INPUT_PACKET_BYTES = 8 * 2 * SAMPLES_PER_PACKET       # 2560


def decode_input(data: bytes) -> Tuple[np.ndarray, np.ndarray]:
    """bytes (multiple of 16) -> (x1, x2) float64 arrays; == util.conv_bytearray_2_2floatarray."""
    if len(data) % 16:
        raise ValueError("input packet length must be a multiple of 16 bytes")
    a = np.frombuffer(data, dtype="<f8").reshape(-1, 2)
    return a[:, 0].copy(), a[:, 1].copy()


def encode_input(x1: Sequence[float], x2: Sequence[float]) -> bytes:
    """== util.conv_2floatarray_2_bytearray."""
    x1 = np.asarray(x1, dtype="<f8")
    x2 = np.asarray(x2, dtype="<f8")
    if x1.shape != x2.shape:
        raise ValueError("Two arrays must have the same length")
    return np.stack([x1, x2], axis=1).tobytes()


def _arr(values) -> bytes:
    a = np.asarray([float(v) for v in values] if not isinstance(values, np.ndarray) else values, dtype="<f8").reshape(-1)
    return struct.pack("<I", a.size) + a.tobytes()


def encode_result(res: Dict, mode: str = "vap") -> bytes:
    """Result dict -> payload bytes; == util.conv_vapresult_2_bytearray (/_bc/_nod)."""
    b = struct.pack("<d", float(res["t"])) + _arr(res["x1"]) + _arr(res["x2"])
    if mode == "vap":
        keys = ("p_now", "p_future", "vad")
    elif mode == "bc":
        keys = ("p_bc_react", "p_bc_emo")
    elif mode == "nod":
        keys = ("p_bc", "p_nod_short", "p_nod_long", "p_nod_long_p")
    else:
        raise ValueError(mode)
    for k in keys:
        b += _arr(res[k])
    return b


def frame_result(res: Dict, mode: str = "vap") -> bytes:
    """Length-prefixed packet as proc_serv_out_dist sends it (vap_main.py:446-448)."""
    payload = encode_result(res, mode)
    return len(payload).to_bytes(4, "little") + payload


def frame_results_batch(t: float, echo: np.ndarray, p_now: np.ndarray, p_future: np.ndarray, vad: np.ndarray) -> np.ndarray:
    """All result packets of one tick at once (mode "vap"): ``echo`` float64 [R,2,n], the three heads float [R,2] ->
    uint8 [R, 4 + payload]; row k is byte-identical to ``frame_result`` of stream k.  One vectorised fill instead of
    R x 5 small encodes: the per-stream Python cost was the front-end's bottleneck."""
    R, _, n = echo.shape
    plen = 8 + 2 * (4 + 8 * n) + 3 * (4 + 16)
    buf = np.empty((R, 4 + plen), dtype=np.uint8)
    o = 0

    def put(raw: bytes):
        nonlocal o
        buf[:, o:o + len(raw)] = np.frombuffer(raw, dtype=np.uint8)
        o += len(raw)

    def put_f64(a: np.ndarray):                     # a: [R, m] -> u32 count + m little-endian doubles per row
        nonlocal o
        m = a.shape[1]
        put(struct.pack("<I", m))
        buf[:, o:o + 8 * m] = np.ascontiguousarray(a, dtype="<f8").view(np.uint8).reshape(R, 8 * m)
        o += 8 * m

    put(plen.to_bytes(4, "little"))
    put(struct.pack("<d", float(t)))
    put_f64(echo[:, 0])
    put_f64(echo[:, 1])
    put_f64(np.asarray(p_now, dtype=np.float64))
    put_f64(np.asarray(p_future, dtype=np.float64))
    put_f64(np.asarray(vad, dtype=np.float64))
    assert o == 4 + plen
    return buf


def decode_result(payload: bytes, mode: str = "vap") -> Dict:
    """== util.conv_bytearray_2_vapresult (/_bc/_nod)."""
    idx = 0
    out: Dict = {"t": struct.unpack_from("<d", payload, idx)[0]}
    idx += 8
    if mode == "vap":
        keys = ("x1", "x2", "p_now", "p_future", "vad")
    elif mode == "bc":
        keys = ("x1", "x2", "p_bc_react", "p_bc_emo")
    else:
        keys = ("x1", "x2", "p_bc", "p_nod_short", "p_nod_long", "p_nod_long_p")
    for k in keys:
        if idx >= len(payload):          # the library flavour omits the vad block (vap_realtime/util.py:195-213)
            break
        n = struct.unpack_from("<I", payload, idx)[0]
        idx += 4
        out[k] = np.frombuffer(payload, dtype="<f8", count=n, offset=idx).tolist()
        idx += 8 * n
    return out


class PacketAssembler:
    """Frame assembly of ``proc_serv_in`` for many streams (vap_main.py:368-409), vectorised.

    Each stream accumulates 160-sample packets; when ``hop`` new samples are present the frame is
    ready.  The 320-sample carry lives on the device (vapx_step with samples_per_ch == hop), so the
    host only ever moves NEW samples."""

    def __init__(self, n_streams: int, hop: int, gain: float = 1.0):
        assert hop % SAMPLES_PER_PACKET == 0
        self.hop = hop
        self.gain = gain
        self.buf = np.zeros((n_streams, 2, hop), dtype=np.float32)
        self.echo = np.zeros((n_streams, 2, hop), dtype=np.float64)   # x1/x2 echoed in result packets
        self.fill = np.zeros(n_streams, dtype=np.int64)

    def push(self, sid: int, data: bytes) -> bool:
        """Append one or more whole packets; returns True when the stream's frame is complete."""
        a = np.frombuffer(data, dtype="<f8").reshape(-1, 2)
        if self.gain != 1.0:
            a = a * self.gain                           # float64 multiply, as vap_main.py:393-395
        n = a.shape[0]
        f = int(self.fill[sid])
        if f + n > self.hop:
            raise ValueError("packet overruns the frame; pop the frame first")
        self.buf[sid, :, f:f + n] = a.T                 # float64 -> float32 cast == vap_main.py:266-270
        self.echo[sid, :, f:f + n] = a.T                # current_x{1,2}_audio keep the float64 samples (:258-259)
        self.fill[sid] = f + n
        return f + n == self.hop

    def ready(self) -> np.ndarray:
        return np.nonzero(self.fill == self.hop)[0]

    def pop(self, sids: np.ndarray) -> np.ndarray:
        """float32 [len(sids), 2, hop] new samples of complete frames; resets those streams."""
        out = self.buf[sids].copy()
        self.last_echo = self.echo[sids].copy()
        self.fill[sids] = 0
        return out
