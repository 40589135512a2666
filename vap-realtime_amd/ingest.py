"""ctypes binding of libvapx's native many-stream TCP front-end (include/vapx.h, ``vapx_ingest_*``).

Stands in for the reference's server threads (``proc_serv_in`` / ``proc_serv_out`` / ``proc_serv_out_dist``,
rvap/vap_main/vap_main.py:338-457) for thousands of dialogues per process: epoll receive threads decode the 2560-byte
packets straight into page-locked staging, a tick thread steps the ready streams, sender threads write result packets
byte-identical to ``rvap/common/util.py``'s.  ``server.ManyStreamServer`` is the readable Python twin of the same
behaviour (one thread, ~2 k real-time streams); this one carries a whole GPU.

``NativeServer(engine)`` serves an ``engine.Engine``; ``NativeServer.over_function(step, ...)`` runs the same front-end
over a Python step function (host-logic tests without a GPU).
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Optional

import numpy as np

from . import engine as _engine

MODE = _engine.MODE


class _IngestConfig(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("port_in", C.c_int32), ("port_out", C.c_int32), ("rx_threads", C.c_int32),
                ("tx_threads", C.c_int32), ("max_wait_us", C.c_int32), ("min_batch", C.c_int32), ("reset_on_connect", C.c_int32),
                ("broadcast", C.c_int32), ("bind_any", C.c_int32), ("gain", C.c_double), ("target_util_pct", C.c_int32),
                ("flags", C.c_int32), ("cpu_first", C.c_int32), ("cpu_count", C.c_int32)]


class _IngestStats(C.Structure):
    _fields_ = [("frames_done", C.c_int64), ("ticks", C.c_int64), ("rx_bytes", C.c_int64), ("tx_bytes", C.c_int64),
                ("in_connections", C.c_int64), ("out_connections", C.c_int64), ("dropped_listeners", C.c_int64),
                ("numeric_resets", C.c_int64), ("overruns", C.c_int64), ("mean_batch", C.c_double), ("lat_mean_ms", C.c_double),
                ("lat_p50_ms", C.c_double), ("lat_p99_ms", C.c_double), ("lat_max_ms", C.c_double), ("step_mean_ms", C.c_double)]


_STEP_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(C.c_float))
_RESET_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int32)


class NativeServer:
    def __init__(self, eng: "_engine.Engine", port_in: int = 50007, port_out: int = 50008, gain: float = 1.0, max_wait_s: float = 0.002,
                 min_batch: int = 0, reset_on_connect: bool = True, broadcast: Optional[bool] = None, rx_threads: int = 0,
                 tx_threads: int = 0, bind_any: bool = False, target_util: float = 0.9, cores: Optional[tuple] = None,
                 keep_nofile: bool = False, core_set: bool = False):
        """``cores`` = (first, count): pin the front-end's tick / receive / sender threads to that core range (the GPU's NUMA node; keep
        load generators and other tenants off it), one core each — or, with ``core_set``, all of them to the range as one affinity set.  ``keep_nofile``: never raise the process's RLIMIT_NOFILE (include/vapx.h)."""
        self.lib = _engine.load_library()
        self._keep = [eng]
        cfg = self._cfg(port_in, port_out, gain, max_wait_s, min_batch, reset_on_connect, broadcast, rx_threads, tx_threads, bind_any, target_util,
                        cores, keep_nofile, core_set)
        h = C.c_void_p()
        rc = self.lib.vapx_ingest_open(eng._h, C.byref(cfg), C.byref(h))
        if rc != 0:
            raise _engine.VapxError(f"vapx_ingest_open failed ({rc})")
        self._h = h
        self._ports()

    @staticmethod
    def _cfg(port_in, port_out, gain, max_wait_s, min_batch, reset_on_connect, broadcast, rx_threads, tx_threads, bind_any, target_util=0.9,
             cores=None, keep_nofile=False, core_set=False):
        first, count = (int(cores[0]), int(cores[1])) if cores else (0, 0)
        return _IngestConfig(C.sizeof(_IngestConfig), port_in, port_out, rx_threads, tx_threads, int(max_wait_s * 1e6), min_batch,
                             1 if reset_on_connect else 0, -1 if broadcast is None else int(bool(broadcast)), int(bool(bind_any)), gain,
                             int(round(target_util * 100)), (1 if keep_nofile else 0) | (2 if core_set else 0), first, count)

    @classmethod
    def over_function(cls, step: Callable, n_streams: int, frame_hz: int = 20, mode: str = "vap", max_batch: Optional[int] = None,
                      reset: Optional[Callable] = None, port_in: int = 0, port_out: int = 0, gain: float = 1.0, max_wait_s: float = 0.002,
                      min_batch: int = 0, reset_on_connect: bool = True, broadcast: Optional[bool] = None, rx_threads: int = 0,
                      tx_threads: int = 0, target_util: float = 1.0):
        """``step(ids int32[n], audio float32[n,2,hop], out float32[n,OUT_STRIDE]) -> int`` fills ``out`` in place;
        ``reset(stream_id)`` gets negative ids (``-(id+1)``) for carry-only resets."""
        self = cls.__new__(cls)
        self.lib = _engine.load_library()
        hop = 16000 // frame_hz

        def _step(_user, n, ids, audio, out):
            try:
                i = np.ctypeslib.as_array(ids, shape=(n,))
                a = np.ctypeslib.as_array(audio, shape=(n, 2, hop))
                o = np.ctypeslib.as_array(out, shape=(n, _engine.OUT_STRIDE))
                o[:, _engine.OUT_STATUS] = 0.0
                return int(step(i, a, o) or 0)
            except Exception:          # noqa: BLE001 — never unwind through the C caller
                import traceback
                traceback.print_exc()
                return -1

        def _reset(_user, sid):
            if reset is not None:
                reset(int(sid))

        self._keep = [_STEP_FN(_step), _RESET_FN(_reset)]
        cfg = cls._cfg(port_in, port_out, gain, max_wait_s, min_batch, reset_on_connect, broadcast, rx_threads, tx_threads, False, target_util)
        h = C.c_void_p()
        rc = self.lib.vapx_ingest_open_fn(C.cast(self._keep[0], C.c_void_p), C.cast(self._keep[1], C.c_void_p), None, n_streams,
                                          max_batch or n_streams, frame_hz, MODE[mode], C.byref(cfg), C.byref(h))
        if rc != 0:
            raise _engine.VapxError(f"vapx_ingest_open_fn failed ({rc})")
        self._h = h
        self._ports()
        return self

    @classmethod
    def over_native_function(cls, step_ptr: int, user_ptr: int, n_streams: int, frame_hz: int = 20, mode: str = "vap", max_batch: Optional[int] = None,
                             keep=(), port_in: int = 0, port_out: int = 0, gain: float = 1.0, max_wait_s: float = 0.002, min_batch: int = 0,
                             rx_threads: int = 0, tx_threads: int = 0, target_util: float = 0.9, cores: Optional[tuple] = None):
        """The same front-end over a C step function (address of a ``vapx_ingest_step_fn``, ``user_ptr`` handed to it): no Python on the tick
        thread.  ``tools/server_load.py --standin`` puts a stand-in for the GPU engine here (tools/standin_step.cpp) to load-test the host side of
        N x 4096 dialogues; ``port_in = port_out = -1`` makes it a passive shard of a ``FrontDoor``.  ``keep``: objects that must outlive it."""
        self = cls.__new__(cls)
        self.lib = _engine.load_library()
        self._keep = list(keep)
        cfg = cls._cfg(port_in, port_out, gain, max_wait_s, min_batch, True, None, rx_threads, tx_threads, False, target_util, cores)
        h = C.c_void_p()
        rc = self.lib.vapx_ingest_open_fn(C.c_void_p(step_ptr), None, C.c_void_p(user_ptr), n_streams, max_batch or n_streams, frame_hz, MODE[mode],
                                          C.byref(cfg), C.byref(h))
        if rc != 0:
            raise _engine.VapxError(f"vapx_ingest_open_fn failed ({rc})")
        self._h = h
        self._ports()
        return self

    def attach_link(self, link_fd: int):
        """Make this PASSIVE front-end a shard of a front door in another process (``RemoteFrontDoor``): ``link_fd`` is this process's end of an
        ``AF_UNIX / SOCK_SEQPACKET`` socket pair; accepted connections arrive over it as descriptors (include/vapx.h)."""
        rc = self.lib.vapx_ingest_attach_link(self._h, int(link_fd))
        if rc != 0:
            raise _engine.VapxError(f"vapx_ingest_attach_link failed ({rc}): the front-end must be passive (port_in = port_out = -1) and not linked yet")

    def _ports(self):
        a, b = C.c_int32(0), C.c_int32(0)
        self.lib.vapx_ingest_ports(self._h, C.byref(a), C.byref(b))
        self.port_in, self.port_out = a.value, b.value

    def stats(self, reset_latency_window: bool = False) -> dict:
        """Counters + the latency window (frame complete on the host -> packet handed to the kernel); ``late_over_10ms`` / ``answered`` are the
        exact counts of the window (read BEFORE an optional reset of the window)."""
        late, ans = C.c_int64(0), C.c_int64(0)
        self.lib.vapx_ingest_late_read(self._h, C.byref(late), C.byref(ans))
        st = _IngestStats()
        self.lib.vapx_ingest_stats_read(self._h, C.byref(st), 1 if reset_latency_window else 0)
        d = {k: getattr(st, k) for k, _ in _IngestStats._fields_}
        d["late_over_10ms"], d["answered"] = late.value, ans.value
        return d

    def close(self):
        if getattr(self, "_h", None):
            self.lib.vapx_ingest_close(self._h)
            self._h = None

    stop = close

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def decode_input(data: bytes, gain: float = 1.0):
    """Native twin of ``wire.decode_input`` (== util.conv_bytearray_2_2floatarray): bytes -> (x1 f64, x2 f64, x1 f32, x2 f32)."""
    lib = _engine.load_library()
    if len(data) % 16:
        raise ValueError("input packet length must be a multiple of 16 bytes")
    n = len(data) // 16
    buf = np.frombuffer(data, dtype=np.uint8)
    x1, x2 = np.empty(n, np.float64), np.empty(n, np.float64)
    f1, f2 = np.empty(n, np.float32), np.empty(n, np.float32)
    got = lib.vapx_wire_decode_input(buf.ctypes.data_as(C.c_void_p), len(data), gain, f1.ctypes.data_as(C.c_void_p), f2.ctypes.data_as(C.c_void_p),
                                     x1.ctypes.data_as(C.c_void_p), x2.ctypes.data_as(C.c_void_p))
    assert got == n
    return x1, x2, f1, f2


def encode_result(mode: str, t: float, x1: np.ndarray, x2: np.ndarray, out_row: np.ndarray) -> bytes:
    """Native twin of ``wire.frame_result``: one length-prefixed result packet for a ``vapx_step`` output row."""
    lib = _engine.load_library()
    x1 = np.ascontiguousarray(x1, np.float64)
    x2 = np.ascontiguousarray(x2, np.float64)
    row = np.ascontiguousarray(out_row, np.float32)
    assert row.size >= _engine.OUT_STRIDE and x1.size == x2.size
    need = lib.vapx_wire_encode_result(MODE[mode], t, x1.ctypes.data_as(C.c_void_p), x2.ctypes.data_as(C.c_void_p), x1.size,
                                       row.ctypes.data_as(C.c_void_p), None, 0)
    if need < 0:
        raise ValueError(f"vapx_wire_encode_result failed ({need})")
    dst = np.empty(need, np.uint8)
    got = lib.vapx_wire_encode_result(MODE[mode], t, x1.ctypes.data_as(C.c_void_p), x2.ctypes.data_as(C.c_void_p), x1.size,
                                      row.ctypes.data_as(C.c_void_p), dst.ctypes.data_as(C.c_void_p), need)
    assert got == need
    return dst.tobytes()


def link_pair():
    """(door end, worker end) of a front-door link: ``socket.socketpair(AF_UNIX, SOCK_SEQPACKET)``; hand the worker end to the process that owns
    the GPU (``subprocess.Popen(..., pass_fds=[worker.fileno()])``), keep both objects alive as long as the link is used."""
    import socket
    a, b = socket.socketpair(socket.AF_UNIX, socket.SOCK_SEQPACKET)
    a.set_inheritable(False)
    b.set_inheritable(True)
    return a, b


class RemoteFrontDoor:
    """The reference's ONE port pair in front of N per-GPU WORKER PROCESSES (``vapx_frontdoor_open_links``): this process only accepts, decides the
    dialogue's GPU and slot (same placement as ``FrontDoor``) and passes the socket on; engines, audio and results never come here.  ``links`` are
    the door ends of ``link_pair()``; every worker runs a passive ``NativeServer`` with ``attach_link(worker_end)``.  The constructor waits for the
    workers' greetings (they may still be loading weights)."""

    def __init__(self, links, port_in: int = 50007, port_out: int = 50008, bind_any: bool = False):
        self.lib = _engine.load_library()
        self._links = list(links)
        fds = [l if isinstance(l, int) else l.fileno() for l in self._links]
        arr = (C.c_int32 * len(fds))(*fds)
        h = C.c_void_p()
        rc = self.lib.vapx_frontdoor_open_links(arr, len(fds), port_in, port_out, int(bool(bind_any)), C.byref(h))
        if rc != 0:
            raise _engine.VapxError(f"vapx_frontdoor_open_links failed ({rc}): a worker did not greet (died while starting?), the workers disagree on "
                                    f"frame rate / mode, or the ports are taken")
        self._h = h
        a, b = C.c_int32(0), C.c_int32(0)
        self.lib.vapx_frontdoor_ports(self._h, C.byref(a), C.byref(b))
        self.port_in, self.port_out = a.value, b.value

    counts = None      # (bound below: same as FrontDoor.counts)

    def close(self):
        if getattr(self, "_h", None):
            self.lib.vapx_frontdoor_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FrontDoor:
    """ONE port pair in front of N per-GPU front-ends (``vapx_frontdoor_*``): the reference's single ``port_num_in`` / ``port_num_out``
    (vap_main.py:338-366,470-471) for a whole node.  ``shards`` are ``NativeServer`` objects opened PASSIVE (``port_in=-1, port_out=-1``),
    one per engine / GPU.  Dialogue slots are global, ``g = local_slot * N + shard``: input connections take the lowest free one (GPUs
    fill evenly; a reconnecting dialogue returns to the GPU holding its state while its slot is the lowest free one), the k-th output
    connection hears the k-th dialogue."""

    def __init__(self, shards, port_in: int = 50007, port_out: int = 50008, bind_any: bool = False):
        self.lib = _engine.load_library()
        self.shards = list(shards)
        arr = (C.c_void_p * len(self.shards))(*[s._h.value for s in self.shards])
        h = C.c_void_p()
        rc = self.lib.vapx_frontdoor_open(arr, len(self.shards), port_in, port_out, int(bool(bind_any)), C.byref(h))
        if rc != 0:
            raise _engine.VapxError(f"vapx_frontdoor_open failed ({rc}): shards must be passive front-ends of one frame rate and mode, ports free")
        self._h = h
        a, b = C.c_int32(0), C.c_int32(0)
        self.lib.vapx_frontdoor_ports(self._h, C.byref(a), C.byref(b))
        self.port_in, self.port_out = a.value, b.value

    @staticmethod
    def owner_of(global_slot: int, n_shards: int):
        """(shard, local slot) of a global dialogue slot."""
        return global_slot % n_shards, global_slot // n_shards

    def counts(self) -> dict:
        a, b, c = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        self.lib.vapx_frontdoor_counts(self._h, C.byref(a), C.byref(b), C.byref(c))
        return {"accepted_in": a.value, "accepted_out": b.value, "refused": c.value}

    def stats(self, reset_latency_window: bool = False) -> list:
        return [s.stats(reset_latency_window) for s in self.shards]

    def close(self, close_shards: bool = True):
        if getattr(self, "_h", None):
            self.lib.vapx_frontdoor_close(self._h)
            self._h = None
        if close_shards:
            for s in self.shards:
                s.close()

    stop = close

    def __del__(self):
        try:
            self.close(close_shards=False)
        except Exception:
            pass


RemoteFrontDoor.counts = FrontDoor.counts
