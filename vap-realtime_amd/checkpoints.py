"""Checkpoint importer: real ``asset/*.pt`` / Hugging Face ``maai-kyoto/*`` files -> libvapx weights blob.

What the reference does with its two checkpoint files, and what this module restates:

* the CPC file is a dict with a ``"weights"`` entry holding ``gEncoder.*`` / ``gAR.*`` tensors
  (``rvap/vap_main/encoder_components.py:372-399``); the path is ``asset/cpc/60k_epoch4-d0f474de.pt`` by default
  (``rvap/vap_main/vap_main.py:479``);
* the VAP file is a flat state dict.  Its ``encoder.*`` keys are skipped (``vap_main.py:199-201``) except the four
  ``encoder.downsample.*`` tensors, which are assigned onto both encoders (``vap_main.py:203-212``);
* file names encode (mode, language, frame rate, context length) — ``vap_realtime/util.py:15-56`` for the Hugging
  Face repos and ``README.md`` / ``asset/vap`` for the local layout.

No network here or in production serving: ``find_checkpoint`` only searches local directories and the local
Hugging Face cache layout (``<cache>/models--<org>--<name>/snapshots/<rev>/<file>``) and raises
``FileNotFoundError`` naming every path it tried.  Nothing in this module touches the GPU.
"""
from __future__ import annotations

import glob
import os
from typing import Dict, Iterable, List, Optional, Tuple

import numpy as np

from . import weights as _weights

DEFAULT_CPC_FILE = "60k_epoch4-d0f474de.pt"          # vap_main.py:479

# (mode, language) -> (HF repo, file-name template); templates take {hz} and {ms}.  vap_realtime/util.py:4-56.
_CATALOG: Dict[Tuple[str, str], Tuple[str, str]] = {
    ("vap", "jp"): ("maai-kyoto/vap_jp", "vap_state_dict_jp_{hz}hz_{ms}msec.pt"),
    ("vap", "en"): ("maai-kyoto/vap_en", "vap_state_dict_eng_{hz}hz_{ms}msec.pt"),
    ("vap", "tri"): ("maai-kyoto/vap_tri", "vap_state_dict_tri_ecj_{hz}hz_{ms}msec.pt"),
    ("vap_MC", "jp"): ("maai-kyoto/vap_MC", "vap_state_dict_jp_{hz}hz_{ms}msec_MC.pt"),
    ("vap_MC", "en"): ("maai-kyoto/vap_MC", "vap_state_dict_en_{hz}hz_{ms}msec_MC.pt"),
    ("vap_MC", "tri"): ("maai-kyoto/vap_MC", "vap_state_dict_tri_{hz}hz_{ms}msec_MC.pt"),
    ("bc", "jp"): ("maai-kyoto/vap_bc_jp", "vap-bc_state_dict_erica_{hz}hz_{ms}msec.pt"),
    ("nod", "jp"): ("maai-kyoto/vap_nod_jp", "vap-nod_state_dict_erica_{hz}hz_{ms}msec.pt"),
}

# engine head set for each catalog mode ("vap_MC" checkpoints have the plain VAP architecture)
ENGINE_MODE = {"vap": "vap", "vap_MC": "vap", "bc": "bc", "nod": "nod"}


def checkpoint_name(mode: str, frame_rate: int, context_len_sec: float, language: str = "jp") -> Tuple[str, str]:
    """(hf_repo_id, file_name) the reference would fetch for this model (vap_realtime/util.py:15-56)."""
    if mode not in ENGINE_MODE:
        raise ValueError(f"Invalid mode: {mode}")
    key = (mode, language)
    if key not in _CATALOG:
        raise ValueError(f"Invalid language: {language}")
    repo, tmpl = _CATALOG[key]
    return repo, tmpl.format(hz=int(frame_rate), ms=int(context_len_sec * 1000))


def _candidate_paths(repo: str, fname: str, search_dirs: Iterable[str], cache_dir: Optional[str]) -> List[str]:
    out = []
    for d in search_dirs:
        out += [os.path.join(d, fname), os.path.join(d, "vap", fname), os.path.join(d, "asset", "vap", fname),
                os.path.join(d, "cpc", fname), os.path.join(d, "asset", "cpc", fname)]
    caches = [cache_dir] if cache_dir else []
    caches += [os.environ.get("HF_HUB_CACHE"), os.path.join(os.environ.get("HF_HOME", ""), "hub") if
               os.environ.get("HF_HOME") else None, os.path.expanduser("~/.cache/huggingface/hub")]
    for c in [c for c in caches if c]:
        out.append(os.path.join(c, "models--" + repo.replace("/", "--"), "snapshots", "*", fname))
    return out


def find_checkpoint(mode: str, frame_rate: int, context_len_sec: float, language: str = "jp",
                    search_dirs: Iterable[str] = (".",), cache_dir: Optional[str] = None) -> str:
    """Local path of the VAP state-dict file for (mode, rate, context, language)."""
    repo, fname = checkpoint_name(mode, frame_rate, context_len_sec, language)
    tried = _candidate_paths(repo, fname, search_dirs, cache_dir)
    for pat in tried:
        hits = sorted(glob.glob(pat)) if "*" in pat else ([pat] if os.path.isfile(pat) else [])
        if hits:
            return hits[-1]
    raise FileNotFoundError(f"{fname} ({repo}) not found locally; there is no download path. Tried: " + ", ".join(tried))


def find_cpc(search_dirs: Iterable[str] = (".",), fname: str = DEFAULT_CPC_FILE) -> str:
    tried = []
    for d in search_dirs:
        for p in (os.path.join(d, fname), os.path.join(d, "cpc", fname), os.path.join(d, "asset", "cpc", fname)):
            tried.append(p)
            if os.path.isfile(p):
                return p
    raise FileNotFoundError(f"CPC checkpoint {fname} not found; tried: " + ", ".join(tried))


def _np(t) -> np.ndarray:
    if hasattr(t, "detach"):
        t = t.detach().cpu().float().numpy()
    return np.ascontiguousarray(np.asarray(t, dtype=np.float32))


def _torch_load(path, allow_unsafe: bool = False):
    """``torch.load`` with the safe (weights-only) unpickler.  The stock CPC checkpoint carries an ``argparse.Namespace``
    next to its tensors (the reference loads it with a full unpickle, encoder_components.py:372-380): that one class is
    allow-listed.  Anything else the safe loader rejects is NOT retried with the arbitrary-code unpickler unless the
    caller opts in (``allow_unsafe=True`` or ``VAPX_ALLOW_UNSAFE_PICKLE=1``), and the opt-in is logged."""
    import argparse
    import pickle
    import sys
    import torch
    try:
        with torch.serialization.safe_globals([argparse.Namespace]):
            return torch.load(path, map_location="cpu", weights_only=True)
    except pickle.UnpicklingError as e:
        if not (allow_unsafe or os.environ.get("VAPX_ALLOW_UNSAFE_PICKLE") == "1"):
            raise RuntimeError(f"{path}: rejected by the weights-only unpickler ({e}). If the file is trusted, pass "
                               f"allow_unsafe=True or set VAPX_ALLOW_UNSAFE_PICKLE=1 to unpickle it in full.") from e
        print(f"[vapx] WARNING: loading {path} with the full (arbitrary-code) unpickler", file=sys.stderr)
        return torch.load(path, map_location="cpu", weights_only=False)


def load_state_dicts(vap_model, cpc_model, allow_unsafe: bool = False):
    """Paths (``torch.load``-ed like vap_main.py:199 and encoder_components.py:372) or ready dicts ->
    ``(cpc_sd, vap_sd)`` with the CPC ``"weights"`` wrapper removed."""
    vap_sd = _torch_load(vap_model, allow_unsafe) if isinstance(vap_model, (str, bytes, os.PathLike)) else vap_model
    cpc_sd = _torch_load(cpc_model, allow_unsafe) if isinstance(cpc_model, (str, bytes, os.PathLike)) else cpc_model
    if "weights" in cpc_sd:
        cpc_sd = cpc_sd["weights"]
    return cpc_sd, vap_sd


def infer_mode(vap_sd) -> str:
    """Head set of a VAP state dict from its keys (vap_bc_main.py:137, vap_nod_main.py:137-138)."""
    if "nod_head.weight" in vap_sd:
        return "nod"
    if "bc_head.weight" in vap_sd:
        return "bc"
    return "vap"


def infer_frame_rate(vap_sd) -> int:
    """Frame rate from the downsample kernel width K = n_cpc (train/encoder.py:33-42)."""
    K = int(vap_sd["encoder.downsample.1.weight"].shape[-1])
    for hz in (50, 20, 10, 5):
        if _weights.cpc_frames_for_rate(hz) == K:
            return hz
    raise ValueError(f"downsample kernel width {K} matches no supported frame rate")


def validate(cpc_sd, vap_sd, frame_rate: int, mode: str = "vap") -> None:
    """Raise ``KeyError`` / ``ValueError`` naming the first missing or mis-shaped tensor; extra keys are ignored
    exactly like ``load_state_dict(strict=False)`` does (vap_main.py:215)."""
    K = _weights.cpc_frames_for_rate(frame_rate)
    for name, shape, _ in _weights._cpc_spec():
        if name not in cpc_sd:
            raise KeyError(f"CPC checkpoint lacks {name}")
        if tuple(cpc_sd[name].shape) != tuple(shape):
            raise ValueError(f"CPC tensor {name}: shape {tuple(cpc_sd[name].shape)}, expected {tuple(shape)}")
    for name, shape, kind in _weights._vap_spec(K, mode):
        if kind in ("alibi", "codebook") and name not in vap_sd:
            continue                                      # constants the kernels hard-code; checked when present
        if name not in vap_sd:
            raise KeyError(f"VAP state dict lacks {name} (mode {mode}, {frame_rate} Hz)")
        if tuple(vap_sd[name].shape) != tuple(shape):
            raise ValueError(f"VAP tensor {name}: shape {tuple(vap_sd[name].shape)}, expected {tuple(shape)} "
                             f"(is this a {frame_rate} Hz checkpoint?)")


def import_checkpoints(vap_model, cpc_model, frame_rate: Optional[int] = None, mode: Optional[str] = None):
    """-> ``(blob, frame_rate, mode)``; rate and head set are inferred from the state dict when not given."""
    cpc_sd, vap_sd = load_state_dicts(vap_model, cpc_model)
    mode = mode or infer_mode(vap_sd)
    frame_rate = frame_rate or infer_frame_rate(vap_sd)
    validate(cpc_sd, vap_sd, frame_rate, mode)
    cpc_np = {k: _np(v) for k, v in cpc_sd.items()}
    vap_np = {k: _np(v) for k, v in vap_sd.items() if not hasattr(v, "dtype") or str(v.dtype) != "torch.bool"}
    return _weights.pack_blob(cpc_np, vap_np, mode), frame_rate, mode


def load_vap_model(mode: str, frame_rate: int, context_len_sec: float, language: str = "jp", device: str = "cpu",
                   cache_dir: Optional[str] = None, search_dirs: Iterable[str] = (".",)):
    """Same call as ``vap_realtime.util.load_vap_model`` (util.py:15-66) minus the download: returns the state
    dict found in the local directories / Hugging Face cache."""
    return _torch_load(find_checkpoint(mode, frame_rate, context_len_sec, language, search_dirs, cache_dir))
