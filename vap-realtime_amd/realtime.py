"""Drop-in host objects mirroring the reference's realtime classes, backed by libvapx.

* ``VAPRealTime``   — same constructor, attributes and ``process_vap(x1, x2)`` as
  ``rvap/vap_main/vap_main.py:185-335`` (and the bc / nod twins), so the reference's server threads
  (``proc_serv_in`` :354-414, ``proc_serv_out_dist`` :416-457) run unchanged with this object.
* ``VapGPT``        — the model-attribute surface ``process_vap`` itself touches
  (``encode_audio``, ``ar_channel``, ``ar``, ``vap_head``, ``va_classifier``,
  ``objective.probs_next_speaker_aggregate``; SURVEY.md §8b level 1), torch CUDA tensors in/out.
* ``ManyStreamVAP`` — the many-stream front object (what the reference lacks: it serves exactly one
  stream per process, vap_main.py:354-366): S independent dialogue streams per GPU, one
  ``process(frames)`` call per tick.

No CPU fallback anywhere: constructing any of these without libvapx.so / a gfx950 GPU raises.
"""
from __future__ import annotations

import time
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import engine as _engine
from . import weights as _weights

BINS_P_NOW = [0, 1]       # vap_main.py:187
BINS_PFUTURE = [2, 3]     # vap_main.py:188


def _load_state_dicts(vap_model, cpc_model):
    """Accept paths (torch.load like vap_main.py:199 / encoder_components.py:372) or ready dicts."""
    from . import checkpoints
    return checkpoints.load_state_dicts(vap_model, cpc_model)


class VAPRealTime:
    """One stream, reference-compatible.  ``mode``: "vap" | "bc" | "nod"."""

    BINS_P_NOW = BINS_P_NOW
    BINS_PFUTURE = BINS_PFUTURE
    CALC_PROCESS_TIME_INTERVAL = 100

    def __init__(self, vap_model, cpc_model, device=None, frame_rate: int = 20, context_len_sec: float = 2.5,
                 mode: str = "vap", **engine_options):
        cpc_sd, vap_sd = _load_state_dicts(vap_model, cpc_model)
        self.mode = mode
        self.device = device
        dev_id = 0
        if device is not None and getattr(device, "index", None) is not None:
            dev_id = int(device.index)
        if device is not None and str(device).startswith("cpu"):
            raise _engine.VapxError("vap-realtime_amd has no CPU path; pass a cuda device (MI355X)")
        self.engine = _engine.Engine(_weights.pack_blob(cpc_sd, vap_sd, mode), frame_rate, context_len_sec,
                                     max_streams=1, mode=mode, device_id=dev_id, **engine_options)
        self.audio_contenxt_lim_sec = context_len_sec
        self.frame_rate = frame_rate
        self.audio_context_len = int(context_len_sec * frame_rate)
        self.sampling_rate = 16000
        self.frame_contxt_padding = 320
        self.audio_frame_size = self.sampling_rate // frame_rate + self.frame_contxt_padding
        self.current_x1_audio: list = []
        self.current_x2_audio: list = []
        self.result_p_now = 0.
        self.result_p_future = 0.
        self.result_last_time = -1
        self.result_vad = [0., 0.]
        self.result_p_bc_react = 0.
        self.result_p_bc_emo = 0.
        self.result_p_bc = 0.
        self.result_p_nod_short = 0.
        self.result_p_nod_long = 0.
        self.result_p_nod_long_p = 0.
        self.result_logits = None
        self.process_time_abs = -1
        self.list_process_time_context: List[float] = []
        self.last_interval_time = time.time()

    def process_vap(self, x1, x2):
        """x1, x2: list or ndarray of ``audio_frame_size`` samples (carry included), exactly what the
        reference's proc_serv_in / vap_offline hand over (vap_main.py:402-405, vap_offline.py:51-61)."""
        time_start = time.time()
        self.current_x1_audio = x1[self.frame_contxt_padding:]
        self.current_x2_audio = x2[self.frame_contxt_padding:]
        frame = np.stack([np.asarray(x1, dtype=np.float32), np.asarray(x2, dtype=np.float32)])[None]
        if frame.shape[2] != self.audio_frame_size:
            raise ValueError(f"expected {self.audio_frame_size} samples per channel, got {frame.shape[2]}")
        o = _engine.split_outputs(self.engine.step(frame))
        if self.mode == "vap":
            self.result_p_now = [float(v) for v in o["p_now"][0]]
            self.result_p_future = [float(v) for v in o["p_future"][0]]
            self.result_vad = [float(o["vad"][0, 0]), float(o["vad"][0, 1])]
            self.result_logits = o["logits"][0].copy()
        elif self.mode == "bc":
            self.result_p_bc_react = [float(o["aux"][0, 1])]
            self.result_p_bc_emo = [float(o["aux"][0, 2])]
        else:
            self.result_p_nod_short = [float(o["aux"][0, 1])]
            self.result_p_nod_long = [float(o["aux"][0, 2])]
            self.result_p_nod_long_p = [float(o["aux"][0, 3])]
            self.result_p_bc = o["logits"][0, :int(o["n"][0])].reshape(-1, 1).copy()   # all n rows (reference quirk)
        self.result_last_time = time.time()
        self.list_process_time_context.append(time.time() - time_start)
        if len(self.list_process_time_context) > self.CALC_PROCESS_TIME_INTERVAL:
            ave = float(np.average(self.list_process_time_context))
            fps = len(self.list_process_time_context) / (time.time() - self.last_interval_time)
            self.last_interval_time = time.time()
            print('[VAP] Average processing time: %.5f [sec], #process/sec: %.3f' % (ave, fps))
            self.list_process_time_context = []
        self.process_time_abs = time.time()

    def get_result(self) -> Dict:
        """Library-twin result dict: vap keys vap_realtime/model.py:189-194, bc :208-214, nod :226-240 (``Vap.get_result``
        switches on the loaded model's mode)."""
        r = {"t": self.result_last_time, "x1": self.current_x1_audio, "x2": self.current_x2_audio}
        if self.mode == "vap":
            r.update(p_now=self.result_p_now, p_future=self.result_p_future, vad=self.result_vad)
        elif self.mode == "bc":
            r.update(p_bc_react=self.result_p_bc_react, p_bc_emo=self.result_p_bc_emo)
        else:
            r.update(p_bc=self.result_p_bc, p_nod_short=self.result_p_nod_short, p_nod_long=self.result_p_nod_long,
                     p_nod_long_p=self.result_p_nod_long_p)
        return r


class Vap(VAPRealTime):
    """The library twin (``vap_realtime/model.py:15-257``): same constructor arguments, ``start_process()`` / ``worker()`` pulling
    160-sample chunks from two microphone-like objects (anything with ``get_audio_data()`` and ``start_process()``), ``process_vap``,
    and ``get_result()`` as a BLOCKING queue read of one result dict per processed frame (``:189-194,208-214,226-240,256-257``).
    The checkpoint is resolved with the reference's file naming (``vap_realtime/util.py:15-56``) in local directories / the local
    Hugging Face cache — there is no download path (``force_download`` is rejected)."""

    def __init__(self, mode, frame_rate, context_len_sec, language: str = "jp", mic1=None, mic2=None, num_channels: int = 2,
                 cpc_model: str = "~/.cache/cpc/60k_epoch4-d0f474de.pt", device="cuda", cache_dir: Optional[str] = None,
                 force_download: bool = False, search_dirs: Sequence[str] = (".", "asset"), **engine_options):
        import os
        import queue
        from . import checkpoints
        if force_download:
            raise _engine.VapxError("vap-realtime_amd never downloads checkpoints; place the files locally")
        if num_channels != 2:
            raise _engine.VapxError("the engine is built for the 2-channel models (vap_MC with other channel counts is out of scope)")
        sd = checkpoints.load_vap_model(mode, frame_rate, context_len_sec, language, search_dirs=list(search_dirs), cache_dir=cache_dir)
        dev = device
        if isinstance(device, str):
            import torch
            dev = torch.device(device)
        super().__init__(sd, os.path.expanduser(cpc_model), dev, frame_rate, context_len_sec,
                         mode="vap" if mode == "vap_MC" else mode, **engine_options)
        self.mic1, self.mic2 = mic1, mic2
        self.result_dict_queue = queue.Queue()

    def process_vap(self, x1, x2):
        super().process_vap(x1, x2)
        import copy
        r = VAPRealTime.get_result(self)
        r["t"] = time.time()
        self.result_dict_queue.put({k: copy.copy(v) for k, v in r.items()})

    def worker(self):
        """``vap_realtime/model.py:96-119``: accumulate microphone chunks, step on every complete frame, keep the 320-sample carry."""
        current_x1 = np.zeros(self.frame_contxt_padding)
        current_x2 = np.zeros(self.frame_contxt_padding)
        while not getattr(self, "_stop_worker", False):
            x1 = self.mic1.get_audio_data()
            x2 = self.mic2.get_audio_data()
            if x1 is None or x2 is None:          # (a finite test source ran dry; real microphones block instead)
                break
            current_x1 = np.concatenate([current_x1, x1])
            current_x2 = np.concatenate([current_x2, x2])
            if len(current_x1) < self.audio_frame_size:
                continue
            self.process_vap(current_x1, current_x2)
            current_x1 = current_x1[-self.frame_contxt_padding:]
            current_x2 = current_x2[-self.frame_contxt_padding:]

    def start_process(self):
        import threading
        self.mic1.start_process()
        self.mic2.start_process()
        self._worker_thread = threading.Thread(target=self.worker, daemon=True)
        self._worker_thread.start()

    def get_result(self):
        return self.result_dict_queue.get()


class ManyStreamVAP:
    """S independent streams on one GPU.  ``process(new_samples[, stream_ids])`` = one tick."""

    def __init__(self, cpc_sd, vap_sd, frame_rate: int = 20, context_len_sec: float = 2.5, n_streams: int = 256,
                 max_batch: Optional[int] = None, mode: str = "vap", device_id: int = 0, **engine_options):
        """``engine_options`` go to ``engine.Engine`` (e.g. ``split_f16=True``, ``groups=2``)."""
        self.engine = _engine.Engine(_weights.pack_blob(cpc_sd, vap_sd, mode), frame_rate, context_len_sec,
                                     max_streams=n_streams, max_batch=max_batch, mode=mode, device_id=device_id,
                                     **engine_options)
        self.n_streams = n_streams
        self.hop = 16000 // frame_rate
        self.mode = mode

    def process(self, new_samples: np.ndarray, stream_ids: Optional[Sequence[int]] = None,
                on_numeric: str = "raise") -> Dict[str, np.ndarray]:
        """new_samples float [n,2,hop] (or [n,2,hop+320] complete frames) -> dict of [n,...] arrays.  A stream with non-finite
        results raises ``VapxError`` (the engine's fail-loudly rule: an offline run must not write 'nan' rows for the rest of a
        file); a caller that handles the per-stream ``status`` column itself — the TCP front-end resets that dialogue and keeps
        serving the others — passes ``on_numeric="status"``."""
        res = _engine.split_outputs(self.engine.step(new_samples, stream_ids, on_numeric=on_numeric))
        if self.mode == "nod":      # p_bc of every window row (vap_nod_main.py:276 quirk) sits in the logits columns
            res["p_bc"] = [res["logits"][k, :int(res["n"][k])] for k in range(len(res["n"]))]
        return res

    def reset(self, stream_id: int):
        self.engine.reset_stream(stream_id)


# ------------------------------------------------------------------------------------------------
# level-1 surface: what VAPRealTime.process_vap calls on self.vap (vap_main.py:272-307)
# ------------------------------------------------------------------------------------------------
class _Objective:
    """``vap.objective``: only ``probs_next_speaker_aggregate`` is on the path (vap_main.py:297-307); a HIP kernel."""

    def __init__(self, lib):
        self.lib = lib

    def probs_next_speaker_aggregate(self, probs, from_bin: int = 0, to_bin: int = 3, scale_with_bins: bool = False):
        """objective.py:186-206 on a torch CUDA tensor [..., 256] -> [..., 2]."""
        import torch
        if scale_with_bins:
            raise NotImplementedError("scale_with_bins is never used by the realtime programs")
        p = probs.float().contiguous()
        rows = p.numel() // 256
        out = torch.empty(*p.shape[:-1], 2, device=p.device)
        rc = self.lib.vapx_aggregate(rows, p.data_ptr(), from_bin, to_bin, out.data_ptr(), torch.cuda.current_stream().cuda_stream or None)
        if rc != 0:
            raise _engine.VapxError(f"vapx_aggregate failed ({rc})")
        return out


_LOGITS_CLS = None


def _logits_cls():
    """torch.Tensor subclass returned by ``VapGPT.vap_head``: ``logits.softmax(dim=-1)`` (vap_main.py:295) runs the HIP row
    softmax; every other operation is the plain tensor's."""
    global _LOGITS_CLS
    if _LOGITS_CLS is None:
        import torch

        class VapLogits(torch.Tensor):
            def softmax(self, dim=-1, dtype=None):
                t = self.as_subclass(torch.Tensor)
                if dtype is not None or dim not in (-1, t.dim() - 1) or not t.is_cuda or t.shape[-1] != 256 or t.dtype != torch.float32:
                    return t.softmax(dim, dtype=dtype)
                t = t.contiguous()
                y = torch.empty_like(t)
                rc = _engine.load_library().vapx_softmax256(t.numel() // 256, t.data_ptr(), y.data_ptr(),
                                                            torch.cuda.current_stream().cuda_stream or None)
                if rc != 0:
                    raise _engine.VapxError(f"vapx_softmax256 failed ({rc})")
                return y

        _LOGITS_CLS = VapLogits
    return _LOGITS_CLS


class VapGPT:
    """``self.vap`` replacement: torch CUDA tensors in/out, HIP kernels inside (stage-level C ABI)."""

    def __init__(self, cpc_sd, vap_sd, frame_rate: int = 20, context_len_sec: float = 2.5, max_batch: int = 1,
                 mode: str = "vap", device_id: int = 0):
        import torch
        self._torch = torch
        self.engine = _engine.Engine(_weights.pack_blob(cpc_sd, vap_sd, mode), frame_rate, context_len_sec,
                                     max_streams=max_batch, mode=mode, device_id=device_id)
        self.device = torch.device("cuda", device_id)
        self.objective = _Objective(self.engine.lib)

    def to(self, device):
        return self

    def eval(self):
        return self

    def _stream(self):
        return self._torch.cuda.current_stream().cuda_stream

    def encode_audio(self, audio1, audio2):
        """[B,1,L] x2 -> ([B,1,256], [B,1,256]); stateful LSTM like EncoderCPC (vap_main.py:175-180)."""
        torch = self._torch
        B = audio1.shape[0]
        frames = torch.stack([audio1.reshape(B, -1), audio2.reshape(B, -1)], dim=1).float().contiguous()
        e = torch.empty(B, 2, 256, device=self.device)
        self.engine.encode_audio_device(B, frames.data_ptr(), e.data_ptr(), stream=self._stream())
        return e[:, 0:1].contiguous(), e[:, 1:2].contiguous()

    def ar_channel(self, x, attention: bool = False):
        """GPT.forward (1 self-attention layer), [B,n,256] -> {"x": [B,n,256]} (vap_main.py:285-286).  The two towers share
        weights and layer 0 has no cross-channel term, so a batch of B inputs rides as ceil(B/2) (stream, channel) pairs:
        no input is computed twice."""
        torch = self._torch
        B, n, _ = x.shape
        x = x.float()
        if B % 2:
            x = torch.cat([x, x[-1:]], dim=0)                         # odd batch: one padding row
        P = x.shape[0] // 2
        xin = x.reshape(P, 2, n, 256).contiguous()
        o = torch.empty(P, 2, n, 256, device=self.device)
        self.engine.transformer_device(P, n, xin.data_ptr(), o_ptr=o.data_ptr(), stage=1, stream=self._stream())
        return {"x": o.reshape(2 * P, n, 256)[:B].contiguous()}

    def ar(self, x1, x2, attention: bool = False):
        """GPTStereo.forward (3 self+cross layers + Combinator) (vap_main.py:287)."""
        torch = self._torch
        B, n, _ = x1.shape
        xin = torch.stack([x1, x2], dim=1).float().contiguous()
        x12 = torch.empty(B, 2, n, 256, device=self.device)
        comb = torch.empty(B, n, 256, device=self.device)
        self.engine.transformer_device(B, n, xin.data_ptr(), x12_ptr=x12.data_ptr(), comb_ptr=comb.data_ptr(), stage=2,
                                       stream=self._stream())
        return {"x": comb, "x1": x12[:, 0].contiguous(), "x2": x12[:, 1].contiguous()}

    def forward(self, waveform, attention: bool = False, lang_info: list = None):
        """The training module's call signature (train/model.py:292-319): ``waveform [B,2,N] -> {"logits": [B,n,256],
        "vad": [B,n,2]}`` (vad = classifier outputs BEFORE the sigmoid, as there).  Semantics are the REALTIME ones
        ("follow realtime"): the waveform is cut into hop-sized frames exactly like ``proc_serv_in`` does
        (zero carry before the first frame, vap_main.py:368-409), each frame sees the last T frames, the LSTM state
        runs through the whole call and starts from zero, VAD reads the ar_channel output (vap_main.py:292-293).
        n = N // hop.  Attention maps are not materialised by the fused kernels."""
        if attention:
            raise NotImplementedError("attention maps are never materialised by the fused attention block")
        torch = self._torch
        B, two, N = waveform.shape
        assert two == 2
        eng = self.engine
        if B > eng.max_batch:
            raise _engine.VapxError(f"batch {B} exceeds max_batch {eng.max_batch}")
        for b in range(B):
            eng.reset_stream(b)
        hop = eng.hop
        n = N // hop
        wav = waveform.to(self.device).float()
        out = torch.empty(B, _engine.OUT_STRIDE, device=self.device)
        logits = torch.empty(B, n, 256, device=self.device)
        vad = torch.empty(B, n, 2, device=self.device)
        for f in range(n):
            x = wav[:, :, f * hop:(f + 1) * hop].contiguous()
            eng.step_device(B, x.data_ptr(), hop, out.data_ptr(), stream=self._stream())
            logits[:, f] = out[:, _engine.OUT_LOGITS:_engine.OUT_LOGITS + 256]
            vad[:, f] = out[:, _engine.OUT_VAD_LOGIT:_engine.OUT_VAD_LOGIT + 2]
        return {"logits": logits, "vad": vad}

    __call__ = forward

    def vap_head(self, t):
        """Linear(256, 256) + bias on any [..., 256] tensor (vap_main.py:131,290): the engine's fp32-MFMA GEMM."""
        torch = self._torch
        x = t.float().contiguous()
        y = torch.empty_like(x)
        self.engine._check(self.engine.lib.vapx_vap_head(self.engine._h, x.numel() // 256, x.data_ptr(), y.data_ptr(), self._stream() or None),
                           "vapx_vap_head")
        return y.as_subclass(_logits_cls())

    def va_classifier(self, t):
        """Linear(256, 1) + bias -> [..., 1] (vap_main.py:142,292-293); the caller applies the sigmoid."""
        torch = self._torch
        x = t.float().contiguous()
        y = torch.empty(*x.shape[:-1], 1, device=x.device)
        self.engine._check(self.engine.lib.vapx_va_classifier(self.engine._h, x.numel() // 256, x.data_ptr(), y.data_ptr(), self._stream() or None),
                           "vapx_va_classifier")
        return y


# ------------------------------------------------------------------------------------------------
# level-3 surface: the functional step the reference wraps for ONNX export (tools/vap_static.py:235-304)
# ------------------------------------------------------------------------------------------------
class VAPRealTimeStatic:
    """``forward(x1_, x2_, e1_context, e2_context) -> (p_now[B,2], p_future[B,2], vad1[B,1], vad2[B,1], e1[B,1,256],
    e2[B,1,256])`` with the caller holding the embedding context (first call: zeros ``[B,1,256]``, afterwards the returned
    embeddings, at most T-1 rows).  Only the LSTM state lives inside, as in the reference (``encode_audio`` is stateful).
    Same constructor as the reference class (tools/vap_static.py:177); torch CUDA tensors in and out."""

    BINS_P_NOW = BINS_P_NOW
    BINS_PFUTURE = BINS_PFUTURE

    def __init__(self, vap_model, cpc_model, device=None, frame_rate: int = 20, context_len_sec: float = 2.5, max_batch: int = 1):
        import torch
        cpc_sd, vap_sd = _load_state_dicts(vap_model, cpc_model)
        if device is not None and str(device).startswith("cpu"):
            raise _engine.VapxError("vap-realtime_amd has no CPU path; pass a cuda device (MI355X)")
        dev_id = int(device.index) if device is not None and getattr(device, "index", None) is not None else 0
        self.vap_gpt = VapGPT(cpc_sd, vap_sd, frame_rate, context_len_sec, max_batch=max_batch, device_id=dev_id)
        self.device = torch.device("cuda", dev_id)
        self.frame_rate = frame_rate
        self.audio_contenxt_lim_sec = context_len_sec
        self.audio_context_len = int(context_len_sec * frame_rate)
        self.sampling_rate = 16000
        self.frame_contxt_padding = 320
        self.audio_frame_size = self.sampling_rate // frame_rate + self.frame_contxt_padding

    def forward(self, x1_, x2_, e1_context, e2_context):
        import torch
        vg, dev = self.vap_gpt, self.device
        e1, e2 = vg.encode_audio(x1_.to(dev), x2_.to(dev))
        x1 = torch.cat([e1_context.to(dev).float(), e1], dim=1)
        x2 = torch.cat([e2_context.to(dev).float(), e2], dim=1)
        if x1.shape[1] > self.audio_context_len:
            raise _engine.VapxError(f"context of {x1.shape[1] - 1} rows exceeds T-1 = {self.audio_context_len - 1}")
        o1, o2 = vg.ar_channel(x1), vg.ar_channel(x2)
        out = vg.ar(o1["x"], o2["x"])
        probs = vg.vap_head(out["x"]).softmax(dim=-1)
        p_now = vg.objective.probs_next_speaker_aggregate(probs, self.BINS_P_NOW[0], self.BINS_P_NOW[-1])
        p_future = vg.objective.probs_next_speaker_aggregate(probs, self.BINS_PFUTURE[0], self.BINS_PFUTURE[1])
        vad1 = vg.va_classifier(o1["x"]).sigmoid()[::, -1]
        vad2 = vg.va_classifier(o2["x"]).sigmoid()[::, -1]
        return p_now[:, -1, :], p_future[:, -1, :], vad1, vad2, e1, e2

    __call__ = forward
