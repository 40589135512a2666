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
from dataclasses import dataclass, field
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


BIN_TIMES: list = [0.2, 0.4, 0.6, 0.8]     # vap_main.py:33


@dataclass
class VapConfig:
    """Fields and defaults of the reference's ``VapConfig`` (vap_main.py:35-85; the bc / nod programs and the library twin carry the
    same class: vap_realtime/vap_models.py:19-69).  The engine is built for the one architecture every published checkpoint has;
    ``VapGPT(conf)`` rejects a ``conf`` that asks for another one (see ``_check_conf``)."""
    sample_rate: int = 16000
    frame_hz: int = 50
    bin_times: List[float] = field(default_factory=lambda: BIN_TIMES)
    encoder_type: str = "cpc"
    wav2vec_type: str = "mms"
    hubert_model: str = "hubert_jp"
    freeze_encoder: int = 1
    load_pretrained: int = 1
    only_feature_extraction: int = 0
    dim: int = 256
    channel_layers: int = 1
    cross_layers: int = 3
    num_heads: int = 4
    dropout: float = 0.1
    context_limit: int = -1
    context_limit_cpc_sec: float = -1
    lid_classify: int = 0
    lid_classify_num_class: int = 3
    lid_classify_adversarial: int = 0
    lang_cond: int = 0

    @staticmethod
    def add_argparse_args(parser, fields_added=[]):
        for k, v in VapConfig.__dataclass_fields__.items():
            if k == "bin_times":
                parser.add_argument(f"--vap_{k}", nargs="+", type=float, default=v.default_factory())
            else:
                parser.add_argument(f"--vap_{k}", type=v.type if callable(v.type) else {"int": int, "float": float, "str": str}[v.type], default=v.default)
            fields_added.append(k)
        return parser, fields_added

    @staticmethod
    def args_to_conf(args):
        return VapConfig(**{k.replace("vap_", ""): v for k, v in vars(args).items() if k.startswith("vap_")})


_FIXED_CONF = {"sample_rate": 16000, "encoder_type": "cpc", "dim": 256, "channel_layers": 1, "cross_layers": 3, "num_heads": 4,
               "context_limit": -1, "load_pretrained": 1}


def _check_conf(conf):
    """The HIP kernels are written for dim 256 / 4 heads / 1 + 3 layers / the CPC encoder / full causal attention: any other
    value of those fields is refused here, at construction, instead of computing something else.  ``frame_hz``, ``dropout`` (inference),
    ``freeze_encoder`` and the training-only multi-task switches do not touch the realtime path (the frame rate that counts is the one
    of the loaded downsample kernel, vap_main.py:203-212)."""
    for k, want in _FIXED_CONF.items():
        got = getattr(conf, k, want)
        if got != want:
            raise _engine.VapxError(f"VapConfig.{k} = {got!r}: libvapx implements {k} = {want!r} only (the published Realtime-VAP architecture)")
    if list(getattr(conf, "bin_times", BIN_TIMES)) != BIN_TIMES:
        raise _engine.VapxError(f"VapConfig.bin_times = {conf.bin_times!r}: the 256-class codebook of libvapx is the one of {BIN_TIMES}")


def _to_np(t) -> np.ndarray:
    if hasattr(t, "detach"):
        t = t.detach().cpu().float().numpy()
    return np.ascontiguousarray(np.asarray(t, dtype=np.float32))


class _Slot:
    """A parameter holder (``.weight`` / ``.bias``) that takes whatever the reference assigns to it — ``nn.Parameter``, tensor, ndarray
    (vap_main.py:204-212) — and tells the owning model that its device weights are stale."""

    def __init__(self, owner, **children):
        object.__setattr__(self, "_owner", owner)
        object.__setattr__(self, "weight", None)
        object.__setattr__(self, "bias", None)
        for k, v in children.items():
            object.__setattr__(self, k, v)

    def __setattr__(self, name, value):
        object.__setattr__(self, name, value)
        self._owner._weights_changed()


class _EncoderCPCProxy:
    """``vap.encoder1`` / ``vap.encoder2``: what ``VAPRealTime.__init__`` touches of ``EncoderCPC`` (encoder.py:6-47) — the
    ``downsample`` Sequential's entries ``[1]`` (Conv1d: ``.weight [256,256,K]``, ``.bias``) and ``[2].ln`` (``.weight``, ``.bias``)
    (encoder_components.py:496-511) — plus ``eval() / freeze() / unfreeze()``.  The CPC tensors themselves (``gEncoder.*`` / ``gAR.*``)
    sit in ``cpc_sd`` as loaded from the ``cpc_model`` file."""

    def __init__(self, owner, cpc_sd):
        self.sample_rate = 16000
        self.output_dim = self.dim = 256
        self.downsample_ratio = 320
        self.cpc_sd = cpc_sd
        self.downsample = [None, _Slot(owner), _Slot(owner, ln=_Slot(owner)), None, None]

    def eval(self):
        return self

    def freeze(self):
        pass

    def unfreeze(self):
        raise NotImplementedError("libvapx is an inference engine; there is nothing to train")

    def _downsample_tensors(self):
        d = self.downsample
        return {"encoder.downsample.1.weight": d[1].weight, "encoder.downsample.1.bias": d[1].bias,
                "encoder.downsample.2.ln.weight": d[2].ln.weight, "encoder.downsample.2.ln.bias": d[2].ln.bias}


class _IncompatibleKeys(tuple):
    """What ``nn.Module.load_state_dict`` returns: ``(missing_keys, unexpected_keys)``."""

    def __new__(cls, missing, unexpected):
        return super().__new__(cls, (missing, unexpected))

    missing_keys = property(lambda self: self[0])
    unexpected_keys = property(lambda self: self[1])


_T_CAPACITIES = (64, 256, 512)      # window capacities a lazily built level-1 engine grows through (kernel families of DESIGN §4)


class VapGPT:
    """``self.vap`` replacement (SURVEY §8b level 1) with the reference's CONSTRUCTION surface, so that ``VAPRealTime.__init__``
    (vap_main.py:194-215; library twin vap_realtime/model.py:25-49) runs textually unchanged with only the class names rebound:

        self.vap = VapGPT(VapConfig())                                   # :194-195   (nothing touches the GPU yet)
        self.vap.load_encoder(cpc_model=cpc_model)                       # :200       (CPC file -> encoder1 / encoder2 proxies)
        self.vap.load_state_dict(sd, strict=False)                       # :201       (``encoder.*`` keys ignored like there)
        self.vap.encoder1.downsample[1].weight = nn.Parameter(sd[...])   # :204-212   (eight assignments)
        self.vap.to(self.device); self.vap = self.vap.eval()             # :214-215

    The libvapx engine (device weights + LSTM state) is built at the first call that computes something: by then the downsample kernel
    says the frame rate (K = n_cpc = 100 / rate) and the call says the window length.  ``VAPRealTime`` never tells the model its
    ``context_len_sec``; the engine starts with room for 64 context rows and is rebuilt for 256 / 512 (carrying the LSTM state over) the
    first time a longer window arrives — pass ``context_frames=`` to size it up front.  Assigning a weight after the engine exists
    rebuilds it on the next call.  ``VapGPT.from_state_dicts(cpc_sd, vap_sd, ...)`` is the short form the tests and tools use.

    Torch CUDA tensors in and out; every computation is a HIP kernel behind the C ABI (no rocBLAS, no CPU path)."""

    _MODE = "vap"

    def __init__(self, conf: Optional[VapConfig] = None, *, max_batch: int = 1, context_frames: Optional[int] = None, **engine_options):
        if conf is None:
            conf = VapConfig()
        if not hasattr(conf, "dim"):
            raise TypeError("VapGPT(conf): conf must be a VapConfig; the state-dict form is VapGPT.from_state_dicts(cpc_sd, vap_sd, ...)")
        _check_conf(conf)
        import torch
        self._torch = torch
        self.conf = conf
        self.sample_rate = conf.sample_rate
        self.frame_hz = conf.frame_hz
        self.temp_elapse_time = []
        self._lib = _engine.load_library()            # fails loudly here when libvapx.so is missing
        self.objective = _Objective(self._lib)
        self._sd: Dict[str, object] = {}
        self._engine: Optional[_engine.Engine] = None
        self._stale = False
        self._max_batch = int(max_batch)
        self._ctx_hint = context_frames
        self._engine_options = engine_options
        self._device_id = 0
        self.device = None
        self.training = False

    # -- the reference's construction calls ------------------------------------------------------------------------------------
    @classmethod
    def from_state_dicts(cls, cpc_sd, vap_sd, frame_rate: Optional[int] = None, context_len_sec: float = 2.5, max_batch: int = 1,
                         mode: Optional[str] = None, device_id: int = 0, **engine_options):
        """Ready state dicts (reference key names) -> a model that is already on ``cuda:device_id`` with its engine built."""
        from . import checkpoints
        mode = mode or checkpoints.infer_mode(vap_sd)
        klass = {"vap": VapGPT, "bc": VapGPT_bc, "nod": VapGPT_nod}[mode]
        hz = frame_rate or checkpoints.infer_frame_rate(vap_sd)
        m = klass(VapConfig(), max_batch=max_batch, context_frames=max(1, int(context_len_sec * hz)), **engine_options)
        m.load_encoder(cpc_model=cpc_sd)
        m.load_state_dict(vap_sd, strict=False)
        for enc in (m.encoder1, m.encoder2):
            enc.downsample[1].weight = vap_sd["encoder.downsample.1.weight"]
            enc.downsample[1].bias = vap_sd["encoder.downsample.1.bias"]
            enc.downsample[2].ln.weight = vap_sd["encoder.downsample.2.ln.weight"]
            enc.downsample[2].ln.bias = vap_sd["encoder.downsample.2.ln.bias"]
        m.to(f"cuda:{device_id}").eval()
        m._build(max_batch, m._ctx_hint)
        return m

    def load_encoder(self, cpc_model):
        """vap_main.py:144-169: both channels' ``EncoderCPC`` from ONE ``cpc_model`` file (path as ``load_CPC`` takes it,
        encoder_components.py:372-399 — no download path here — or the loaded dict)."""
        from . import checkpoints
        if isinstance(cpc_model, (str, bytes)) or hasattr(cpc_model, "__fspath__"):
            import os
            if not os.path.isfile(cpc_model):
                raise FileNotFoundError(f"CPC checkpoint {cpc_model!r} not found (the reference would download it; libvapx never does)")
            cpc_model = checkpoints._torch_load(cpc_model)
        cpc_sd = cpc_model["weights"] if "weights" in cpc_model else cpc_model
        self.encoder1 = _EncoderCPCProxy(self, cpc_sd).eval()
        self.encoder2 = _EncoderCPCProxy(self, cpc_sd).eval()
        self._weights_changed()

    def _model_keys(self):
        """Parameter / buffer names the reference module owns (what ``load_state_dict`` matches against)."""
        keys = [n for n, _, _ in _weights._vap_spec(5, self._MODE) if not n.startswith("encoder.")]
        if hasattr(self, "encoder1"):
            for e in ("encoder1", "encoder2"):
                keys += [f"{e}.encoder.{n}" for n, _, _ in _weights._cpc_spec()]
                keys += [f"{e}.downsample.1.weight", f"{e}.downsample.1.bias", f"{e}.downsample.2.ln.weight", f"{e}.downsample.2.ln.bias"]
        return keys

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        """``nn.Module.load_state_dict``: takes the tensors this module owns, reports the rest.  The reference calls it with
        ``strict=False`` (vap_main.py:201) — the file's ``encoder.*`` keys match nothing (the module's encoders are ``encoder1`` /
        ``encoder2``) and are dropped; that is why the four downsample tensors are assigned by hand afterwards."""
        own = self._model_keys()
        missing = [k for k in own if k not in state_dict]
        unexpected = [k for k in state_dict if k not in own]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict for {type(self).__name__}: Missing key(s): {missing[:6]}..., "
                               f"Unexpected key(s): {unexpected[:6]}...")
        shapes = {n: s for n, s, _ in _weights._vap_spec(5, self._MODE)}
        for k in own:
            if k in state_dict and not k.startswith("encoder"):
                if tuple(state_dict[k].shape) != tuple(shapes[k]):
                    raise RuntimeError(f"size mismatch for {k}: copying a param with shape {tuple(state_dict[k].shape)}, the model has {tuple(shapes[k])}")
                self._sd[k] = state_dict[k]
        self._weights_changed()
        return _IncompatibleKeys(missing, unexpected)

    def state_dict(self):
        sd = dict(self._sd)
        if hasattr(self, "encoder1"):
            for e, enc in (("encoder1", self.encoder1), ("encoder2", self.encoder2)):
                sd.update({f"{e}.encoder.{k}": v for k, v in enc.cpc_sd.items()})
                sd.update({k.replace("encoder.", e + ".", 1): v for k, v in enc._downsample_tensors().items() if v is not None})
        return sd

    def to(self, device=None, *args, **kwargs):
        """``self.vap.to(self.device)`` (vap_main.py:214).  There is no CPU path: a cpu device raises."""
        if device is None:
            return self
        dev = self._torch.device(device) if not isinstance(device, self._torch.device) else device
        if dev.type != "cuda":
            raise _engine.VapxError(f"vap-realtime_amd has no CPU path; .to({device!r}) needs a cuda device (MI355X)")
        idx = 0 if dev.index is None else int(dev.index)
        if self._engine is not None and idx != self._device_id:
            self._stale = True
        self._device_id = idx
        self.device = self._torch.device("cuda", idx)
        return self

    def cuda(self, device=None):
        return self.to(f"cuda:{device or 0}")

    def eval(self):
        return self

    def train(self, mode: bool = True):
        if mode:
            raise NotImplementedError("libvapx is an inference engine; train() is not available")
        return self

    def _weights_changed(self):
        self._stale = True

    # -- the lazily built engine -------------------------------------------------------------------------------------------------
    def _gather_state_dicts(self):
        if not hasattr(self, "encoder1"):
            raise _engine.VapxError("VapGPT: load_encoder(cpc_model=...) was never called (vap_main.py:200)")
        d1, d2 = self.encoder1._downsample_tensors(), self.encoder2._downsample_tensors()
        for k in d1:
            if d1[k] is None or d2[k] is None:
                raise _engine.VapxError(f"VapGPT: encoder1/encoder2.{k[len('encoder.'):]} was never assigned — load_state_dict does not load the "
                                        f"downsample tensors, the caller assigns them (vap_main.py:203-212)")
            if not np.array_equal(_to_np(d1[k]), _to_np(d2[k])):
                raise _engine.VapxError(f"VapGPT: encoder1 and encoder2 differ in {k}; the engine serves both channels with one set "
                                        f"(the reference assigns the same tensors to both, vap_main.py:204-212)")
        if self.encoder1.cpc_sd is not self.encoder2.cpc_sd:
            raise _engine.VapxError("VapGPT: encoder1 and encoder2 must come from the same cpc_model (vap_main.py:147-160)")
        vap_sd = {k: _to_np(v) for k, v in self._sd.items()}
        vap_sd.update({k: _to_np(v) for k, v in d1.items()})
        cpc_sd = {k: _to_np(v) for k, v in self.encoder1.cpc_sd.items() if hasattr(v, "shape")}
        return cpc_sd, vap_sd

    def _build(self, batch: int, rows: Optional[int]):
        from . import checkpoints
        if self.device is None:
            self.to("cuda:0")
        cpc_sd, vap_sd = self._gather_state_dicts()
        hz = checkpoints.infer_frame_rate(vap_sd)
        checkpoints.validate(cpc_sd, vap_sd, hz, self._MODE)
        want = max(int(rows or 1), int(self._ctx_hint or 1), self._engine.T if self._engine is not None else 1)
        if self._ctx_hint and want == self._ctx_hint:
            T = want
        else:
            fits = [c for c in _T_CAPACITIES if c >= want]
            if not fits:
                raise _engine.VapxError(f"a window of {want} frames exceeds the engine's maximum of {_T_CAPACITIES[-1]}")
            T = fits[0]
        mb = max(int(batch), self._max_batch)
        old = self._engine
        saved = []
        if old is not None:
            for sid in range(old.max_streams):
                st = old.get_state(sid)
                saved.append((st["lstm"], st["carry"]))
            old.close()
        eng = _engine.Engine(_weights.pack_blob(cpc_sd, vap_sd, self._MODE), hz, (T + 0.5) / hz, max_streams=mb, mode=self._MODE,
                             device_id=self._device_id, **self._engine_options)
        assert eng.T == T, (eng.T, T)
        for sid, (lstm, carry) in enumerate(saved):          # the LSTM state is the model's only memory (encoder.py:27); the window is the caller's
            eng.set_state(sid, {"ring": np.zeros((2, T, 256), np.float32), "n_frames": 0, "lstm": lstm, "carry": carry})
        self._engine, self._max_batch, self._stale = eng, mb, False
        self.frame_hz = hz
        return eng

    @property
    def engine(self) -> "_engine.Engine":
        if self._engine is None or self._stale:
            self._build(self._max_batch, None)
        return self._engine

    def _engine_for(self, batch: int, rows: int = 1):
        if self._engine is None or self._stale or batch > self._engine.max_batch or rows > self._engine.T:
            self._build(batch, rows)
        return self._engine

    def _stream(self):
        return self._torch.cuda.current_stream().cuda_stream

    def encode_audio(self, audio1, audio2):
        """[B,1,L] x2 -> ([B,1,256], [B,1,256]); stateful LSTM like EncoderCPC (vap_main.py:175-180)."""
        torch = self._torch
        B = audio1.shape[0]
        eng = self._engine_for(B, 1)
        frames = torch.stack([audio1.reshape(B, -1), audio2.reshape(B, -1)], dim=1).to(self.device).float().contiguous()
        if frames.shape[2] != eng.L:
            raise ValueError(f"encode_audio: {frames.shape[2]} samples per channel, the loaded {eng.frame_hz} Hz model takes "
                             f"{eng.L} (= 16000 // rate + 320, vap_main.py:230)")
        e = torch.empty(B, 2, 256, device=self.device)
        eng.encode_audio_device(B, frames.data_ptr(), e.data_ptr(), stream=self._stream())
        return e[:, 0:1].contiguous(), e[:, 1:2].contiguous()

    def ar_channel(self, x, attention: bool = False):
        """GPT.forward (1 self-attention layer), [B,n,256] -> {"x": [B,n,256]} (vap_main.py:285-286).  The two towers share
        weights and layer 0 has no cross-channel term, so a batch of B inputs rides as ceil(B/2) (stream, channel) pairs:
        no input is computed twice."""
        torch = self._torch
        if attention:
            raise NotImplementedError("attention maps are never materialised by the fused attention block")
        B, n, _ = x.shape
        x = x.to(self.device).float()
        if B % 2:
            x = torch.cat([x, x[-1:]], dim=0)                         # odd batch: one padding row
        P = x.shape[0] // 2
        xin = x.reshape(P, 2, n, 256).contiguous()
        o = torch.empty(P, 2, n, 256, device=self.device)
        self._engine_for(P, n).transformer_device(P, n, xin.data_ptr(), o_ptr=o.data_ptr(), stage=1, stream=self._stream())
        return {"x": o.reshape(2 * P, n, 256)[:B].contiguous()}

    def ar(self, x1, x2, attention: bool = False):
        """GPTStereo.forward (3 self+cross layers + Combinator) (vap_main.py:287)."""
        torch = self._torch
        if attention:
            raise NotImplementedError("attention maps are never materialised by the fused attention block")
        B, n, _ = x1.shape
        xin = torch.stack([x1, x2], dim=1).to(self.device).float().contiguous()
        x12 = torch.empty(B, 2, n, 256, device=self.device)
        comb = torch.empty(B, n, 256, device=self.device)
        self._engine_for(B, n).transformer_device(B, n, xin.data_ptr(), x12_ptr=x12.data_ptr(), comb_ptr=comb.data_ptr(), stage=2,
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
        if not self._ctx_hint:
            raise _engine.VapxError("forward(waveform) slides the model's own window: construct with context_frames= "
                                    "(VapGPT.from_state_dicts takes context_len_sec)")
        eng = self._engine_for(B, self._ctx_hint)
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
        x = t.to(self.device).float().contiguous()
        y = torch.empty_like(x)
        eng = self.engine
        eng._check(eng.lib.vapx_vap_head(eng._h, x.numel() // 256, x.data_ptr(), y.data_ptr(), self._stream() or None), "vapx_vap_head")
        return y.as_subclass(_logits_cls())

    def va_classifier(self, t):
        """Linear(256, 1) + bias -> [..., 1] (vap_main.py:142,292-293); the caller applies the sigmoid."""
        torch = self._torch
        x = t.to(self.device).float().contiguous()
        y = torch.empty(*x.shape[:-1], 1, device=x.device)
        eng = self.engine
        eng._check(eng.lib.vapx_va_classifier(eng._h, x.numel() // 256, x.data_ptr(), y.data_ptr(), self._stream() or None), "vapx_va_classifier")
        return y

    def _aux_head(self, which: int, nout: int, t):
        torch = self._torch
        x = t.to(self.device).float().contiguous()
        y = torch.empty(*x.shape[:-1], nout, device=x.device)
        eng = self.engine
        eng._check(eng.lib.vapx_aux_head(eng._h, which, x.numel() // 256, x.data_ptr(), y.data_ptr(), self._stream() or None), "vapx_aux_head")
        return y


class VapGPT_bc(VapGPT):
    """``vap_realtime/vap_models.py:157-244`` (and rvap/vap_bc/vap_bc_main.py:88-137): VapGPT + ``bc_head = Linear(256, 3)``; the
    caller takes ``bc_head(out["x"]).softmax(-1)[:, -1, 1 | 2]`` (vap_realtime/model.py:197-200)."""

    _MODE = "bc"

    def vap_head(self, t):
        raise _engine.VapxError("the bc program never applies vap_head (vap_realtime/model.py:196-214); its weights are not on the device")

    def bc_head(self, t):
        """Linear(256, 3) + bias on any [..., 256] tensor -> [..., 3] (before the softmax)."""
        return self._aux_head(0, 3, t)


class VapGPT_nod(VapGPT):
    """``vap_realtime/vap_models.py:246-334`` (and rvap/vap_nod/vap_nod_main.py:88-138): VapGPT + ``nod_head = Linear(256, 4)`` +
    ``bc_head = Linear(256, 1)``; the caller takes ``bc_head(out["x"]).sigmoid()[-1]`` — every row of the window, the reference's
    quirk — and ``nod_head(out["x"]).softmax(-1)[:, -1, 1..3]`` (vap_realtime/model.py:217-224)."""

    _MODE = "nod"

    def vap_head(self, t):
        raise _engine.VapxError("the nod program never applies vap_head (vap_realtime/model.py:215-240); its weights are not on the device")

    def bc_head(self, t):
        """Linear(256, 1) + bias -> [..., 1] (before the sigmoid)."""
        return self._aux_head(0, 1, t)

    def nod_head(self, t):
        """Linear(256, 4) + bias -> [..., 4] (before the softmax)."""
        return self._aux_head(1, 4, t)


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
        self.vap_gpt = VapGPT.from_state_dicts(cpc_sd, vap_sd, frame_rate, context_len_sec, max_batch=max_batch, mode="vap", device_id=dev_id)
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
