#!/usr/bin/env python3
"""Generate golden vectors by running the UNMODIFIED reference (imported from /root/reference).

Runs only where /root/reference exists (the build container).  Nothing of the reference is
copied: the script writes seeded synthetic checkpoints to a temp dir (the real ``asset/*.pt``
files are absent, see .MISSING_LARGE_BLOBS), instantiates the reference's own ``VAPRealTime``
(rvap/vap_main/vap_main.py:185-335, rvap/vap_bc/vap_bc_main.py, rvap/vap_nod/vap_nod_main.py),
drives ``process_vap`` frame by frame exactly like ``proc_serv_in`` (vap_main.py:368-409) or
``vap_offline.py:51-73`` do, and records outputs (+ a few intermediates through forward hooks)
into ``tests/golden/<case>.npz``.  Only inputs' seeds, checksums and outputs are stored.

Usage:  python tools/make_golden.py [case ...]      (each case runs in its own subprocess)
"""
from __future__ import annotations

import contextlib
import io
import os
import subprocess
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, REPO)

CASES = {
    # name: dict(mode, frame_hz, ctx_sec, stream_ids, n_frames, framing, seed, inter_frames)
    "vap20": dict(mode="vap", frame_hz=20, ctx=2.5, streams=[0], n_frames=58, framing="server", seed=0, inter=[0, 1, 52]),
    "vap50": dict(mode="vap", frame_hz=50, ctx=5.0, streams=[1], n_frames=256, framing="server", seed=1, inter=[0, 252], e_stride=16),
    "vap10": dict(mode="vap", frame_hz=10, ctx=5.0, streams=[2], n_frames=54, framing="server", seed=2, inter=[0, 52]),
    "offline20": dict(mode="vap", frame_hz=20, ctx=2.5, streams=[3], n_frames=12, framing="offline", seed=0, inter=[0]),
    "multi3": dict(mode="vap", frame_hz=20, ctx=2.5, streams=[10, 11, 12], n_frames=55, framing="server", seed=3, inter=[], e_stride=6),
    "bc20": dict(mode="bc", frame_hz=20, ctx=2.5, streams=[4], n_frames=56, framing="server", seed=4, inter=[]),
    "nod20": dict(mode="nod", frame_hz=20, ctx=2.5, streams=[5], n_frames=56, framing="server", seed=5, inter=[]),
    # three models on ONE cpc_model file (cpc_seed) and the same audio, as the reference deploys them side by side:
    # pins the shared-trunk serving path (vapx_attach_trunk)
    "trunk_vap20": dict(mode="vap", frame_hz=20, ctx=2.5, streams=[6, 7], n_frames=54, framing="server", seed=7, cpc_seed=6, inter=[], e_stride=6),
    "trunk_bc20": dict(mode="bc", frame_hz=20, ctx=2.5, streams=[6, 7], n_frames=54, framing="server", seed=8, cpc_seed=6, inter=[], e_stride=6),
    "trunk_nod20": dict(mode="nod", frame_hz=20, ctx=2.5, streams=[6, 7], n_frames=54, framing="server", seed=9, cpc_seed=6, inter=[], e_stride=6),
    # degenerate microphone input (synth.degenerate): one independent reference run per kind, on the dialogue of stream 50 + k.
    # This is where ChannelNorm divides by ~0 (encoder_components.py:64-66): pins the oracle AND the HIP path on exactly those inputs.
    "degenerate20": dict(mode="vap", frame_hz=20, ctx=2.5, streams=[50, 51, 52, 53, 54, 55], n_frames=56, framing="server", seed=23,
                         inter=[], e_stride=8, kinds=["silence", "tiny", "full_scale_square", "loud", "dc_offset", "one_channel_dead"]),
    # one NaN / Inf SAMPLE (frame 3): what the unmodified reference does with it (torch.relu and ChannelNorm propagate it, the LSTM
    # state keeps it for good) next to a clean run of the same model
    "poison20": dict(mode="vap", frame_hz=20, ctx=2.5, streams=[60, 61, 62], n_frames=10, framing="server", seed=24,
                     inter=[], e_stride=1, kinds=["clean", "nan_sample", "inf_sample"]),
    # T = 200: the longest published window (20 Hz x 10 s, README.md:381 vap-nod_state_dict_erica_20hz_10000msec) for vap and nod
    "vap20_10s": dict(mode="vap", frame_hz=20, ctx=10.0, streams=[70], n_frames=204, framing="server", seed=25, inter=[0, 202], e_stride=17),
    "nod20_10s": dict(mode="nod", frame_hz=20, ctx=10.0, streams=[71], n_frames=203, framing="server", seed=26, inter=[], e_stride=29),
}
ROW_SUBSET_AT = 8  # intermediates with more rows than this keep rows [0, n//3, n-1] only


def _rows(n):
    return list(range(n)) if n <= ROW_SUBSET_AT else [0, n // 3, n - 1]


def run_case(name: str) -> None:
    import torch
    from vap_realtime_amd import weights as W, synth

    cfg = CASES[name]
    mode, hz = cfg["mode"], cfg["frame_hz"]
    sys.path[:0] = [REF, os.path.join(REF, "rvap", "vap_main")]
    if mode == "vap":
        import vap_main as ref
    elif mode == "bc":
        import rvap.vap_bc.vap_bc_main as ref
    else:
        import rvap.vap_nod.vap_nod_main as ref

    cpc_sd, vap_sd = W.synthetic_weights(cfg["seed"], hz, mode)
    if "cpc_seed" in cfg:
        cpc_sd = W.synthetic_weights(cfg["cpc_seed"], hz, "vap")[0]
    tmp = tempfile.mkdtemp(prefix="vapgold_")
    cpc_pt, vap_pt = os.path.join(tmp, "cpc.pt"), os.path.join(tmp, "vap.pt")
    torch.save({"weights": {k: torch.from_numpy(v.copy()) for k, v in cpc_sd.items()}}, cpc_pt)
    torch.save({k: torch.from_numpy(v.copy()) for k, v in vap_sd.items()}, vap_pt)

    hop = 16000 // hz
    L = hop + 320
    F_, S = cfg["n_frames"], len(cfg["streams"])
    n_samp = hop * F_ + 320
    audio = synth.dialogue_batch(cfg["streams"], n_samp)            # [S,2,n]
    if "kinds" in cfg:
        audio = np.stack([synth.degenerate(audio[i], k) for i, k in enumerate(cfg["kinds"])])

    res = {k: [] for k in ("p_now", "p_future", "vad", "logits", "e")}
    aux = {}
    inter = {}
    for si in range(S):
        with contextlib.redirect_stdout(io.StringIO()):
            rt = ref.VAPRealTime(vap_pt, cpc_pt, torch.device("cpu"), hz, cfg["ctx"])
        vap = rt.vap
        cap = {}
        hooks = []

        def hk(key, fn=lambda o: o):
            def _h(mod, inp, out):
                cap.setdefault(key, []).append(fn(out))
            return _h
        hooks.append(vap.vap_head.register_forward_hook(hk("logits", lambda o: o[0, -1].clone())))
        hooks.append(vap.encoder1.register_forward_hook(hk("e1", lambda o: o[0, 0].clone())))
        hooks.append(vap.encoder2.register_forward_hook(hk("e2", lambda o: o[0, 0].clone())))
        hooks.append(vap.encoder1.encoder.gEncoder.register_forward_hook(hk("cnn4_1", lambda o: o[0].clone())))
        hooks.append(vap.encoder1.encoder.gAR.register_forward_hook(hk("lstm_1", lambda o: o[0].clone())))
        hooks.append(vap.encoder2.encoder.gAR.register_forward_hook(hk("lstm_2", lambda o: o[0].clone())))
        hooks.append(vap.ar_channel.register_forward_hook(hk("o", lambda o: o["x"][0].clone())))
        for l in range(3):
            hooks.append(vap.ar.layers[l].register_forward_hook(
                hk(f"stereo{l}", lambda o: torch.stack([o[0][0], o[1][0]]).clone())))
        hooks.append(vap.ar.combinator.register_forward_hook(hk("comb", lambda o: o[0].clone())))

        carry = np.zeros((2, 320), dtype=np.float64)
        per = {k: [] for k in res}
        per_aux = {}
        for f in range(F_):
            cap.clear()
            if cfg["framing"] == "server":
                # proc_serv_in: float64 samples accumulate on the carry (vap_main.py:391-409)
                new = audio[si, :, f * hop:(f + 1) * hop].astype(np.float64)
                buf = np.concatenate([carry, new], axis=1)
                carry = buf[:, -320:]
            else:
                # vap_offline.py:51-61 — sliding window over the raw signal, hop = frame - 320
                buf = audio[si, :, f * hop:f * hop + L].astype(np.float64)
            with contextlib.redirect_stdout(io.StringIO()):
                rt.process_vap(buf[0].copy(), buf[1].copy())
            if "logits" in cap:     # bc / nod process_vap never evaluates vap_head
                per["logits"].append(cap["logits"][0].numpy())
            if f % cfg.get("e_stride", 1) == 0:
                per["e"].append(torch.stack([cap["e1"][0], cap["e2"][0]]).numpy())
            per["vad"].append(np.array([float(rt.result_vad[0]), float(rt.result_vad[1])], np.float32)
                              if mode == "vap" else np.zeros(2, np.float32))
            if mode == "vap":
                per["p_now"].append(np.array(rt.result_p_now, np.float32))
                per["p_future"].append(np.array(rt.result_p_future, np.float32))
            elif mode == "bc":
                per_aux.setdefault("p_bc_react", []).append(np.float32(rt.result_p_bc_react[0]))
                per_aux.setdefault("p_bc_emo", []).append(np.float32(rt.result_p_bc_emo[0]))
            else:
                per_aux.setdefault("p_nod_short", []).append(np.float32(rt.result_p_nod_short[0]))
                per_aux.setdefault("p_nod_long", []).append(np.float32(rt.result_p_nod_long[0]))
                per_aux.setdefault("p_nod_long_p", []).append(np.float32(rt.result_p_nod_long_p[0]))
                pbc = rt.result_p_bc.numpy().reshape(-1)                       # all n rows (quirk)
                row = np.full(int(cfg["ctx"] * hz), np.nan, np.float32); row[:pbc.size] = pbc
                per_aux.setdefault("p_bc", []).append(row)
            if si == 0 and f in cfg["inter"]:
                inter[f"f{f}.cnn4"] = cap["cnn4_1"][0].numpy()                   # ch-1 CNN out [256,P4]
                inter[f"f{f}.lstm_out"] = np.stack([cap["lstm_1"][0].numpy(), cap["lstm_2"][0].numpy()])
                o = torch.stack(cap["o"]).numpy()                                  # [2,n,256]
                rows = _rows(o.shape[1])
                inter[f"f{f}.rows"] = np.array(rows, np.int32)
                inter[f"f{f}.o"] = o[:, rows]
                for l in range(3):
                    inter[f"f{f}.stereo{l}"] = cap[f"stereo{l}"][0].numpy()[:, rows]
                inter[f"f{f}.comb"] = cap["comb"][0].numpy()[rows]
        for h in hooks:
            h.remove()
        for k in res:
            if per[k]:
                res[k].append(np.stack(per[k]))
        for k, v in per_aux.items():
            aux.setdefault(k, []).append(np.stack(v))

    out = {k: np.stack(v, axis=1).astype(np.float32) for k, v in res.items() if v}   # [F,S,...]
    out.update({k: np.stack(v, axis=1).astype(np.float32) for k, v in aux.items()})
    out.update({f"inter.{k}": v.astype(np.float32) if v.dtype.kind == "f" else v for k, v in inter.items()})
    out["meta.e_stride"] = np.array(cfg.get("e_stride", 1))
    out["meta.mode"] = np.array(mode)
    out["meta.frame_hz"] = np.array(hz)
    out["meta.ctx_sec"] = np.array(cfg["ctx"])
    out["meta.streams"] = np.array(cfg["streams"], np.int64)
    out["meta.n_frames"] = np.array(F_)
    out["meta.framing"] = np.array(cfg["framing"])
    out["meta.seed"] = np.array(cfg["seed"])
    out["meta.cpc_seed"] = np.array(cfg.get("cpc_seed", cfg["seed"]))
    out["meta.weights_fp"] = W.weights_fingerprint(cpc_sd, vap_sd)
    fin = np.where(np.isfinite(audio), audio, 0.0).astype(np.float64)          # (poison20 holds one NaN and one Inf sample)
    out["meta.audio_fp"] = np.array([fin.sum(), np.abs(fin).sum()])
    if "kinds" in cfg:
        out["meta.kinds"] = np.array(cfg["kinds"])
    path = os.path.join(REPO, "tests", "golden", f"{name}.npz")
    np.savez_compressed(path, **out)
    print(f"{name}: wrote {path} ({os.path.getsize(path) / 1024:.0f} KB); "
          + (f"logits range [{out['logits'].min():.2f}, {out['logits'].max():.2f}]" if "logits" in out else ""))


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "--one":
        run_case(sys.argv[2])
    else:
        names = sys.argv[1:] or list(CASES)
        for n in names:
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "--one", n])
