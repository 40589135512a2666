"""ctypes binding of libvapx.so (include/vapx.h) — the only way Python reaches the HIP path.

There is deliberately NO fallback: if the shared library is missing, was built for another
architecture, or no gfx950 device is visible, construction raises.  PyTorch is used only as the
owner of device buffers / HIP streams whose raw pointers are handed to the C ABI.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvapx.so")

OUT_STRIDE = 784
OUT_P_NOW, OUT_P_FUTURE, OUT_VAD, OUT_AUX, OUT_NVALID, OUT_LOGITS, OUT_E = 0, 2, 4, 6, 10, 16, 272
OUT_VAD_LOGIT = 11
OUT_STATUS = 13
E_NUMERIC = -6
ABI_VERSION = 2
AUDIO_DEVICE, OUT_DEVICE, IDS_DEVICE = 1, 2, 4
MODE = {"vap": 0, "bc": 1, "nod": 2}

EXPORTS = ("vapx_abi_version", "vapx_blob_floats", "vapx_create", "vapx_destroy", "vapx_step",
           "vapx_attach_trunk", "vapx_join", "vapx_reset_stream", "vapx_get_state", "vapx_set_state", "vapx_encode_audio",
           "vapx_transformer", "vapx_peek", "vapx_gemm", "vapx_last_error", "vapx_profile_enable",
           "vapx_profile_read", "vapx_bad_slots", "vapx_host_alloc", "vapx_host_free", "vapx_reset_carry", "vapx_get_config",
           "vapx_ingest_open", "vapx_ingest_open_fn", "vapx_ingest_ports", "vapx_ingest_stats_read", "vapx_ingest_late_read", "vapx_ingest_close",
           "vapx_wire_decode_input", "vapx_wire_encode_result", "vapx_vap_head", "vapx_va_classifier", "vapx_softmax256",
           "vapx_aggregate", "vapx_aux_head", "vapx_frontdoor_open", "vapx_frontdoor_open_links", "vapx_ingest_attach_link", "vapx_frontdoor_ports", "vapx_frontdoor_counts", "vapx_frontdoor_close")
PROF_CLASSES = {0: "gemm_store", 1: "gemm_gelu", 2: "gemm_resid", 3: "gemm_resid_ln", 4: "gemm_cn_relu",
                5: "conv_tail", 6: "ffn_block", 7: "last_row", 8: "conv0", 9: "lstm", 10: "gather_ln", 11: "attention", 12: "head",
                13: "gemm_bias_ln_gelu", 14: "ffn_proj"}


class VapxError(RuntimeError):
    pass


class _Config(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("device_id", C.c_int32), ("frame_hz", C.c_int32),
                ("ctx_frames", C.c_int32), ("max_streams", C.c_int32), ("max_batch", C.c_int32),
                ("mode", C.c_int32), ("flags", C.c_int32)]


_lib = None


def load_library(path: Optional[str] = None):
    """dlopen libvapx.so and declare prototypes.  Import torch first when it is going to be used in
    the same process so both share one HIP runtime (same SONAME libamdhip64.so.7)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("VAPX_LIBRARY") or LIB_PATH     # VAPX_LIBRARY: the debug build with phase stamps (make trace)
    if not os.path.exists(p):
        raise VapxError(f"{p} not found — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                        f"(or `make -C vap-realtime_amd/csrc`). There is no CPU fallback.")
    try:
        import torch  # noqa: F401  (loads torch's bundled libamdhip64 first when torch is installed)
    except Exception:  # pragma: no cover
        pass
    lib = C.CDLL(p, mode=C.RTLD_GLOBAL)
    vp, i32, f32p, i32p = C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p
    lib.vapx_abi_version.restype = i32
    lib.vapx_blob_floats.restype = C.c_size_t
    lib.vapx_blob_floats.argtypes = [i32]
    lib.vapx_create.restype = i32
    lib.vapx_create.argtypes = [C.POINTER(_Config), f32p, C.c_size_t, C.POINTER(vp)]
    lib.vapx_destroy.restype = None
    lib.vapx_destroy.argtypes = [vp]
    lib.vapx_step.restype = i32
    lib.vapx_step.argtypes = [vp, i32, i32p, f32p, i32, f32p, i32, vp]
    lib.vapx_join.restype = i32
    lib.vapx_join.argtypes = [vp, vp]
    lib.vapx_attach_trunk.restype = i32
    lib.vapx_attach_trunk.argtypes = [vp, vp]
    lib.vapx_reset_stream.restype = i32
    lib.vapx_reset_stream.argtypes = [vp, i32]
    lib.vapx_get_state.restype = i32
    lib.vapx_get_state.argtypes = [vp, i32, f32p, C.POINTER(i32), f32p, f32p]
    lib.vapx_set_state.restype = i32
    lib.vapx_set_state.argtypes = [vp, i32, f32p, i32, f32p, f32p]
    lib.vapx_encode_audio.restype = i32
    lib.vapx_encode_audio.argtypes = [vp, i32, i32p, f32p, f32p, vp]
    lib.vapx_transformer.restype = i32
    lib.vapx_transformer.argtypes = [vp, i32, i32, f32p, f32p, f32p, f32p, i32, vp]
    lib.vapx_peek.restype = C.c_int64
    lib.vapx_peek.argtypes = [vp, C.c_char_p, f32p, C.c_size_t]
    lib.vapx_gemm.restype = i32
    lib.vapx_gemm.argtypes = [vp, i32, i32, i32, f32p, f32p, f32p, i32, f32p, f32p, f32p, f32p, f32p, i32]
    lib.vapx_profile_enable.restype = i32
    lib.vapx_profile_enable.argtypes = [vp, C.c_uint32]
    lib.vapx_profile_read.restype = i32
    lib.vapx_profile_read.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int64), i32]
    lib.vapx_last_error.restype = C.c_char_p
    lib.vapx_last_error.argtypes = [vp]
    lib.vapx_bad_slots.restype = i32
    lib.vapx_bad_slots.argtypes = [vp, i32p, i32]
    lib.vapx_host_alloc.restype = vp
    lib.vapx_host_alloc.argtypes = [C.c_size_t]
    lib.vapx_host_free.restype = None
    lib.vapx_host_free.argtypes = [vp]
    lib.vapx_reset_carry.restype = i32
    lib.vapx_reset_carry.argtypes = [vp, i32]
    lib.vapx_get_config.restype = i32
    lib.vapx_get_config.argtypes = [vp, C.POINTER(_Config)]
    lib.vapx_ingest_open.restype = i32
    lib.vapx_ingest_open.argtypes = [vp, vp, C.POINTER(vp)]
    lib.vapx_ingest_open_fn.restype = i32
    lib.vapx_ingest_open_fn.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp, C.POINTER(vp)]
    lib.vapx_ingest_ports.restype = i32
    lib.vapx_ingest_ports.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    lib.vapx_ingest_stats_read.restype = i32
    lib.vapx_ingest_stats_read.argtypes = [vp, vp, i32]
    lib.vapx_ingest_late_read.restype = i32
    lib.vapx_ingest_late_read.argtypes = [vp, vp, vp]
    lib.vapx_ingest_close.restype = None
    lib.vapx_ingest_close.argtypes = [vp]
    lib.vapx_frontdoor_open.restype = i32
    lib.vapx_frontdoor_open.argtypes = [vp, i32, i32, i32, i32, C.POINTER(vp)]
    lib.vapx_frontdoor_open_links.restype = i32
    lib.vapx_frontdoor_open_links.argtypes = [C.POINTER(i32), i32, i32, i32, i32, C.POINTER(vp)]
    lib.vapx_ingest_attach_link.restype = i32
    lib.vapx_ingest_attach_link.argtypes = [vp, i32]
    lib.vapx_frontdoor_ports.restype = i32
    lib.vapx_frontdoor_ports.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    lib.vapx_frontdoor_counts.restype = i32
    lib.vapx_frontdoor_counts.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.vapx_frontdoor_close.restype = None
    lib.vapx_frontdoor_close.argtypes = [vp]
    lib.vapx_vap_head.restype = i32
    lib.vapx_vap_head.argtypes = [vp, C.c_int64, f32p, f32p, vp]
    lib.vapx_va_classifier.restype = i32
    lib.vapx_va_classifier.argtypes = [vp, C.c_int64, f32p, f32p, vp]
    lib.vapx_aux_head.restype = i32
    lib.vapx_aux_head.argtypes = [vp, i32, C.c_int64, f32p, f32p, vp]
    lib.vapx_softmax256.restype = i32
    lib.vapx_softmax256.argtypes = [C.c_int64, f32p, f32p, vp]
    lib.vapx_aggregate.restype = i32
    lib.vapx_aggregate.argtypes = [C.c_int64, f32p, i32, i32, f32p, vp]
    lib.vapx_wire_decode_input.restype = C.c_int64
    lib.vapx_wire_decode_input.argtypes = [vp, C.c_size_t, C.c_double, vp, vp, vp, vp]
    lib.vapx_wire_encode_result.restype = C.c_int64
    lib.vapx_wire_encode_result.argtypes = [i32, C.c_double, vp, vp, i32, vp, vp, C.c_size_t]
    if path is None:
        _lib = lib
    return lib


def _np_ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class _PinnedOwner:
    """Frees a vapx_host_alloc block when the last numpy view of it is collected."""

    def __init__(self, lib, ptr):
        self.lib, self.ptr = lib, ptr

    def __del__(self):
        try:
            self.lib.vapx_host_free(self.ptr)
        except Exception:
            pass


def pinned_empty(shape, dtype=np.float32) -> np.ndarray:
    """numpy array in page-locked host memory (vapx_host_alloc): ``Engine.step`` DMAs straight from / into such arrays
    instead of staging through the engine's own pinned buffers."""
    lib = load_library()
    dt = np.dtype(dtype)
    nbytes = int(np.prod(shape)) * dt.itemsize
    ptr = lib.vapx_host_alloc(max(nbytes, 1))
    if not ptr:
        raise VapxError(f"vapx_host_alloc({nbytes}) failed")
    owner = _PinnedOwner(lib, ptr)
    buf = (C.c_char * max(nbytes, 1)).from_address(ptr)
    buf._owner = owner                                   # the ctypes buffer is the numpy base: keeps the block alive
    return np.frombuffer(buf, dtype=dt, count=int(np.prod(shape))).reshape(shape)


class Engine:
    """One engine = one GPU: device weights + state of ``max_streams`` dialogue streams.

    Mirrors ``VAPRealTime(vap_model, cpc_model, device, frame_rate, context_len_sec)``
    (rvap/vap_main/vap_main.py:192-247) with the weights given as a packed blob
    (``weights.pack_blob``)."""

    def __init__(self, blob: np.ndarray, frame_hz: int = 20, context_len_sec: float = 2.5,
                 max_streams: int = 1, max_batch: Optional[int] = None, mode: str = "vap", device_id: int = 0,
                 groups: int = 0, full_last_layer: bool = False, unfused_conv: bool = False,
                 materialize_x0: bool = False, unfused_last_row: bool = False, split_f16: bool = False,
                 unfused_proj: bool = False, split_qkv_in_ffn: bool = False):
        self.lib = load_library()
        self.frame_hz = frame_hz
        self.T = int(context_len_sec * frame_hz)           # vap_main.py:221
        self.hop = 16000 // frame_hz
        self.L = self.hop + 320                             # vap_main.py:230
        self.max_streams = max_streams
        self.max_batch = max_batch or max_streams
        self.mode = mode
        self.device_id = device_id
        blob = np.ascontiguousarray(blob, dtype=np.float32)
        flags = ((groups & 0xF) | (16 if full_last_layer else 0) | (32 if unfused_conv else 0)
                 | (64 if materialize_x0 else 0) | (256 if unfused_last_row else 0) | (512 if split_f16 else 0) | (1024 if unfused_proj else 0)
                 | (2048 if split_qkv_in_ffn else 0))
        cfg = _Config(C.sizeof(_Config), device_id, frame_hz, self.T, max_streams, self.max_batch, MODE[mode], flags)
        h = C.c_void_p()
        rc = self.lib.vapx_create(C.byref(cfg), _np_ptr(blob), blob.size, C.byref(h))
        if rc != 0:
            raise VapxError(f"vapx_create failed ({rc}): {self.lib.vapx_last_error(None).decode()}")
        self._h = h

    # -- lifecycle -------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self.lib.vapx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int, what: str):
        if rc < 0:
            raise VapxError(f"{what} failed ({rc}): {self.lib.vapx_last_error(self._h).decode()}")
        return rc

    # -- the step --------------------------------------------------------------------------------
    def step(self, audio: np.ndarray, stream_ids: Optional[Sequence[int]] = None, out: Optional[np.ndarray] = None,
             on_numeric: str = "raise") -> np.ndarray:
        """Host path.  audio: float [n,2,hop] (new samples; engine keeps the carry) or [n,2,hop+320]
        (complete frames as ``process_vap`` receives them).  Returns float32 [n, OUT_STRIDE] (``out`` if given, e.g. a
        ``pinned_empty`` block).  A stream with non-finite results (VAPX_E_NUMERIC) raises by default; with
        ``on_numeric="status"`` the block is returned — every other row is valid, the bad rows have column OUT_STATUS = 1
        and ``bad_slots()`` lists them."""
        audio = np.ascontiguousarray(audio, dtype=np.float32)
        n, two, spc = audio.shape
        assert two == 2
        ids = None if stream_ids is None else np.ascontiguousarray(stream_ids, dtype=np.int32)
        if ids is not None:
            assert ids.shape == (n,)
        if out is None:
            out = np.empty((n, OUT_STRIDE), dtype=np.float32)
        assert out.dtype == np.float32 and out.flags.c_contiguous and out.size >= n * OUT_STRIDE
        rc = self.lib.vapx_step(self._h, n, _np_ptr(ids), _np_ptr(audio), spc, _np_ptr(out), 0, None)
        if not (rc == E_NUMERIC and on_numeric == "status"):
            self._check(rc, "vapx_step")
        return out.reshape(-1, OUT_STRIDE)[:n]

    def bad_slots(self) -> list:
        """Batch slots of the latest host-path step whose results were not finite."""
        n = self.lib.vapx_bad_slots(self._h, None, 0)
        if n <= 0:
            return []
        buf = np.empty(n, np.int32)
        self.lib.vapx_bad_slots(self._h, _np_ptr(buf), n)
        return buf.tolist()

    def attach_trunk(self, leader: "Engine"):
        """Make this engine a follower of ``leader``: it shares the leader's CPC CNN + LSTM (vapx.h, vapx_attach_trunk)."""
        self._check(self.lib.vapx_attach_trunk(self._h, leader._h), "vapx_attach_trunk")
        self._leader = leader                                # keeps the leader alive as long as the follower

    def step_follow(self, n: int, out: Optional[np.ndarray] = None, on_numeric: str = "raise") -> np.ndarray:
        """Follower step on the host path: consumes the encoder output of the leader's latest ``step`` (``out`` / ``on_numeric``
        as in ``step``)."""
        if out is None:
            out = np.empty((n, OUT_STRIDE), dtype=np.float32)
        assert out.dtype == np.float32 and out.flags.c_contiguous and out.size >= n * OUT_STRIDE
        rc = self.lib.vapx_step(self._h, n, None, None, 0, _np_ptr(out), 0, None)
        if not (rc == E_NUMERIC and on_numeric == "status"):
            self._check(rc, "vapx_step")
        return out.reshape(-1, OUT_STRIDE)[:n]

    def step_follow_device(self, n: int, out_ptr: int, stream: int = 0):
        self._check(self.lib.vapx_step(self._h, n, None, None, 0, out_ptr, OUT_DEVICE, stream or None), "vapx_step")

    def join(self, stream: int = 0):
        self._check(self.lib.vapx_join(self._h, stream or None), "vapx_join")

    def step_device(self, n: int, audio_ptr: int, spc: int, out_ptr: int, ids_ptr: int = 0, stream: int = 0, defer_join: bool = False):
        """Device path: raw device pointers (e.g. ``tensor.data_ptr()``), work enqueued on ``stream``
        (a hipStream_t as int, 0 = default); returns immediately."""
        flags = AUDIO_DEVICE | OUT_DEVICE | (IDS_DEVICE if ids_ptr else 0) | (8 if defer_join else 0)
        self._check(self.lib.vapx_step(self._h, n, ids_ptr or None, audio_ptr, spc, out_ptr, flags, stream or None), "vapx_step")

    def reset_stream(self, sid: int):
        self._check(self.lib.vapx_reset_stream(self._h, sid), "vapx_reset_stream")

    def reset_carry(self, sid: int):
        """Zero only the 320-sample carry (what a reconnect does in the reference, vap_main.py:368-369)."""
        self._check(self.lib.vapx_reset_carry(self._h, sid), "vapx_reset_carry")

    def get_state(self, sid: int):
        ring = np.zeros((2, self.T, 256), np.float32)
        lstm = np.zeros((2, 2, 256), np.float32)
        carry = np.zeros((2, 320), np.float32)
        n = C.c_int32(0)
        if getattr(self, "_leader", None) is not None:       # follower: LSTM / carry live in the leader
            lstm = carry = None
        self._check(self.lib.vapx_get_state(self._h, sid, _np_ptr(ring), C.byref(n), _np_ptr(lstm), _np_ptr(carry)), "vapx_get_state")
        return {"ring": ring, "n_frames": n.value, "lstm": lstm, "carry": carry}

    def set_state(self, sid: int, state: dict):
        ring = np.ascontiguousarray(state["ring"], np.float32)
        lstm = None if state.get("lstm") is None else np.ascontiguousarray(state["lstm"], np.float32)
        carry = None if state.get("carry") is None else np.ascontiguousarray(state["carry"], np.float32)
        self._check(self.lib.vapx_set_state(self._h, sid, _np_ptr(ring), int(state["n_frames"]), _np_ptr(lstm), _np_ptr(carry)), "vapx_set_state")

    def profile_enable(self, classes=()):
        mask = 0
        for c in classes:
            mask |= 1 << int(c)
        self._check(self.lib.vapx_profile_enable(self._h, mask), "vapx_profile_enable")

    def profile_read(self) -> dict:
        """{class_name: (total_ms, launches)} since the last read (synchronises the device)."""
        n = len(PROF_CLASSES)
        ms = (C.c_double * n)()
        cnt = (C.c_int64 * n)()
        self._check(self.lib.vapx_profile_read(self._h, ms, cnt, n), "vapx_profile_read")
        return {PROF_CLASSES[i]: (ms[i], cnt[i]) for i in PROF_CLASSES if cnt[i]}

    def peek(self, name: str, shape) -> np.ndarray:
        buf = np.empty(int(np.prod(shape)), np.float32)
        got = self._check(self.lib.vapx_peek(self._h, name.encode(), _np_ptr(buf), buf.size), "vapx_peek")
        assert got == buf.size, (name, got, buf.size)
        return buf.reshape(shape)

    # -- stage-level (device pointers) -------------------------------------------------------------
    def encode_audio_device(self, n: int, frames_ptr: int, e_ptr: int, stream_ids=None, stream: int = 0):
        ids = None if stream_ids is None else np.ascontiguousarray(stream_ids, dtype=np.int32)
        self._check(self.lib.vapx_encode_audio(self._h, n, _np_ptr(ids), frames_ptr, e_ptr, stream or None), "vapx_encode_audio")

    def transformer_device(self, n: int, rows: int, x_ptr: int, o_ptr: int = 0, x12_ptr: int = 0, comb_ptr: int = 0,
                           stage: int = 0, stream: int = 0):
        self._check(self.lib.vapx_transformer(self._h, n, rows, x_ptr, o_ptr or None, x12_ptr or None, comb_ptr or None,
                                              stage, stream or None), "vapx_transformer")


class TrunkGroup:
    """Several weight sets (vap / bc / nod) served from ONE pass of the CPC CNN + LSTM per tick.

    The reference runs one process per model, each re-encoding the same audio with the same ``cpc_model``
    weights (vap_main.py:199-201, vap_bc_main.py, vap_nod_main.py).  Here ``blobs`` is ``{mode: blob}``; the first
    entry leads (runs the encoder), the others follow.  ``step`` returns ``{mode: out[n, OUT_STRIDE]}``."""

    def __init__(self, blobs: dict, frame_hz: int = 20, context_len_sec: float = 2.5, max_streams: int = 1,
                 max_batch: Optional[int] = None, device_id: int = 0, **engine_kw):
        self.modes = list(blobs)
        self.engines = {}
        for m in self.modes:                                   # engine_kw: groups / split_f16 / ... — the same for every weight set
            self.engines[m] = Engine(blobs[m], frame_hz, context_len_sec, max_streams, max_batch, m, device_id, **engine_kw)
        self.leader = self.engines[self.modes[0]]
        for m in self.modes[1:]:
            self.engines[m].attach_trunk(self.leader)
        self.hop, self.L, self.T = self.leader.hop, self.leader.L, self.leader.T

    def step(self, audio: np.ndarray, stream_ids: Optional[Sequence[int]] = None) -> dict:
        res = {self.modes[0]: self.leader.step(audio, stream_ids)}
        for m in self.modes[1:]:
            res[m] = self.engines[m].step_follow(len(audio))
        return res

    def step_device(self, n: int, audio_ptr: int, spc: int, out_ptrs: dict, ids_ptr: int = 0, stream: int = 0):
        self.leader.step_device(n, audio_ptr, spc, out_ptrs[self.modes[0]], ids_ptr, stream)
        for m in self.modes[1:]:
            self.engines[m].step_follow_device(n, out_ptrs[m], stream)

    def reset_stream(self, sid: int):
        self.leader.reset_stream(sid)                      # cascades to the followers

    def close(self):
        for m in reversed(self.modes):                     # followers before their leader
            self.engines[m].close()


def split_outputs(out: np.ndarray) -> dict:
    """Name the columns of a vapx_step output block."""
    return {
        "p_now": out[:, OUT_P_NOW:OUT_P_NOW + 2], "p_future": out[:, OUT_P_FUTURE:OUT_P_FUTURE + 2],
        "vad": out[:, OUT_VAD:OUT_VAD + 2], "aux": out[:, OUT_AUX:OUT_AUX + 4],
        "n": out[:, OUT_NVALID].astype(np.int32), "logits": out[:, OUT_LOGITS:OUT_LOGITS + 256],
        "vad_logit": out[:, OUT_VAD_LOGIT:OUT_VAD_LOGIT + 2],
        "status": out[:, OUT_STATUS].astype(np.int32),
        "e": out[:, OUT_E:OUT_E + 512].reshape(-1, 2, 256),
    }
