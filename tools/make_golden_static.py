#!/usr/bin/env python3
"""Golden vectors of the reference's functional step ``VAPRealTimeStatic.forward`` (tools/vap_static.py:235-304; the module its
ONNX exporter wraps, SURVEY.md §8b level 3), produced by importing the UNMODIFIED reference — build container only.

Drives ``forward(x1_, x2_, e1_context, e2_context)`` exactly as its docstring prescribes: the first call gets a zero context
``[1,1,256]``, every later call the concatenation of the returned embeddings, trimmed to the last T-1 rows.  Records
(p_now, p_future, vad1, vad2, e1, e2) per frame into tests/golden/static20.npz (seeds and outputs only)."""
import contextlib
import io
import os
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path[:0] = [REF, os.path.join(REF, "rvap", "vap_main"), os.path.join(REF, "tools")]

SEED, HZ, CTX, STREAM, FRAMES = 12, 20, 2.5, 8, 56


def main():
    import torch
    from vap_realtime_amd import synth, weights as W
    with contextlib.redirect_stdout(io.StringIO()):
        import vap_static as ref
    cpc_sd, vap_sd = W.synthetic_weights(SEED, HZ, "vap")
    tmp = tempfile.mkdtemp(prefix="vapgold_")
    cpc_pt, vap_pt = os.path.join(tmp, "cpc.pt"), os.path.join(tmp, "vap.pt")
    torch.save({"weights": {k: torch.from_numpy(v.copy()) for k, v in cpc_sd.items()}}, cpc_pt)
    torch.save({k: torch.from_numpy(v.copy()) for k, v in vap_sd.items()}, vap_pt)
    with contextlib.redirect_stdout(io.StringIO()):
        m = ref.VAPRealTimeStatic(vap_pt, cpc_pt, torch.device("cpu"), HZ, CTX)
    hop, T = 16000 // HZ, int(CTX * HZ)
    L = hop + 320
    audio = synth.dialogue_batch([STREAM], hop * FRAMES + 320)            # [1,2,n]
    e1c = torch.zeros(1, 1, 256)
    e2c = torch.zeros(1, 1, 256)
    out = {k: [] for k in ("p_now", "p_future", "vad", "e")}
    for f in range(FRAMES):
        win = torch.from_numpy(audio[:, :, f * hop:f * hop + L].copy())  # offline framing (vap_offline.py:51-61)
        p_now, p_future, v1, v2, e1, e2 = m.forward(win[:, 0:1], win[:, 1:2], e1c, e2c)
        out["p_now"].append(p_now[0].numpy()); out["p_future"].append(p_future[0].numpy())
        out["vad"].append(np.array([float(v1), float(v2)], np.float32))
        out["e"].append(np.stack([e1[0, 0].numpy(), e2[0, 0].numpy()]))
        # caller-side context protocol of the docstring: first call zeros [1,1,256]; afterwards the embeddings, <= T-1 rows
        e1c = e1 if f == 0 else torch.cat([e1c, e1], dim=1)[:, -(T - 1):]
        e2c = e2 if f == 0 else torch.cat([e2c, e2], dim=1)[:, -(T - 1):]
    z = {k: np.stack(v).astype(np.float32) for k, v in out.items()}
    z.update({"meta.seed": np.array(SEED), "meta.frame_hz": np.array(HZ), "meta.ctx_sec": np.array(CTX),
              "meta.stream": np.array(STREAM), "meta.n_frames": np.array(FRAMES),
              "meta.weights_fp": W.weights_fingerprint(cpc_sd, vap_sd),
              "meta.audio_fp": np.array([audio.astype(np.float64).sum(), np.abs(audio.astype(np.float64)).sum()])})
    path = os.path.join(REPO, "tests", "golden", "static20.npz")
    np.savez_compressed(path, **z)
    print("wrote", path, os.path.getsize(path) // 1024, "KB; p_now range", z["p_now"].min(), z["p_now"].max())


if __name__ == "__main__":
    main()
