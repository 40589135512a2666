"""Offline batch driver: many stereo recordings through one engine, output
identical in format to the reference's ``rvap/vap_main/vap_offline.py`` (``time_sec,p_now(0=left),
p_now(1=right),p_future(0=left),p_future(1=right)``, one row per VAP frame) so that
``output/offline_prediction_visualizer`` keeps working.

Framing follows ``vap_offline.py:47-61``: a window of ``frame = 16000//rate + 320`` samples slides by
``frame - 320``; the first window starts at sample 0 (its 320 "carry" samples are real audio, not zeros);
``time = (i + frame) / 16000``.  Here ALL dialogues advance in lock-step, one ``vapx_step`` per frame
index, dialogues that have run out of audio simply drop out of the batch (ragged stream ids).
"""
from __future__ import annotations

import wave
from typing import Dict, List, Sequence, Tuple

import numpy as np

SR = 16000
HEADER = "time_sec,p_now(0=left),p_now(1=right),p_future(0=left),p_future(1=right)\n"


def read_wav_mono(path: str) -> np.ndarray:
    """16-bit / 32-bit PCM or float wav -> float32 in [-1, 1] (what ``sf.read(dtype='float32')`` yields)."""
    with wave.open(path, "rb") as w:
        if w.getframerate() != SR:
            raise ValueError(f"{path}: expected {SR} Hz, got {w.getframerate()}")
        n, ch, sw = w.getnframes(), w.getnchannels(), w.getsampwidth()
        raw = w.readframes(n)
    if sw == 2:
        x = np.frombuffer(raw, "<i2").astype(np.float32) / 32768.0
    elif sw == 4:
        x = np.frombuffer(raw, "<i4").astype(np.float32) / 2147483648.0
    else:
        raise ValueError(f"{path}: unsupported sample width {sw}")
    return x.reshape(-1, ch)[:, 0].copy()


def frame_starts(n_samples: int, frame: int) -> range:
    """Start offsets of the windows ``vap_offline.py:51-54`` processes."""
    shift = frame - 320
    last = n_samples - frame
    return range(0, last + 1, shift) if last >= 0 else range(0)


def run_offline(vap, dialogues: Sequence[Tuple[np.ndarray, np.ndarray]], on_numeric: str = "raise") -> List[List[Dict]]:
    """``vap``: object with ``.hop`` and ``.process(frames[n,2,hop+320], stream_ids) -> dict`` (``ManyStreamVAP``).
    ``dialogues``: (left, right) float32 arrays.  Returns per dialogue the list of
    ``{"t", "p_now", "p_future"}`` rows.

    ``on_numeric``: a recording with a NaN / Inf sample poisons its LSTM state for good.  The reference keeps going and writes
    ``nan`` rows for the rest of that file (``vap_offline.py:62-73`` has no check); the default here is to FAIL (``"raise"``: a batch
    job should not fill files with ``nan`` silently).  ``"reference"`` reproduces the reference: the poisoned dialogue's rows are
    ``nan`` from that frame on, every other dialogue of the batch is unaffected (streams are independent)."""
    if on_numeric not in ("raise", "reference"):
        raise ValueError("on_numeric must be 'raise' or 'reference'")
    frame = vap.hop + 320
    starts = [frame_starts(min(len(l), len(r)), frame) for l, r in dialogues]
    n_frames = [len(s) for s in starts]
    results: List[List[Dict]] = [[] for _ in dialogues]
    for f in range(max(n_frames, default=0)):
        ids = [d for d in range(len(dialogues)) if f < n_frames[d]]
        batch = np.empty((len(ids), 2, frame), np.float32)
        for k, d in enumerate(ids):
            i = starts[d][f]
            batch[k, 0] = dialogues[d][0][i:i + frame]
            batch[k, 1] = dialogues[d][1][i:i + frame]
        out = (vap.process(batch, np.asarray(ids, np.int32)) if on_numeric == "raise"
               else vap.process(batch, np.asarray(ids, np.int32), on_numeric="status"))
        for k, d in enumerate(ids):
            results[d].append({"t": float(starts[d][f] + frame) / SR,
                               "p_now": [float(v) for v in out["p_now"][k]],
                               "p_future": [float(v) for v in out["p_future"][k]]})
    return results


def write_csv(path: str, rows: List[Dict]) -> None:
    """Same text as ``vap_offline.py:76-86`` (``str()`` of Python floats)."""
    with open(path, "w") as f:
        f.write(HEADER)
        for r in rows:
            f.write(str(r["t"]) + "," + str(r["p_now"][0]) + "," + str(r["p_now"][1]) + ","
                    + str(r["p_future"][0]) + "," + str(r["p_future"][1]) + "\n")


def main(argv=None):
    import argparse
    import torch
    from . import realtime
    ap = argparse.ArgumentParser(description="Many-dialogue offline VAP (reference: rvap/vap_main/vap_offline.py)")
    ap.add_argument("--vap_model", required=True)
    ap.add_argument("--cpc_model", required=True)
    ap.add_argument("--pairs", nargs="+", required=True, help="left.wav:right.wav[:out.txt] ...")
    ap.add_argument("--vap_process_rate", type=int, default=20)
    ap.add_argument("--context_len_sec", type=float, default=5)
    ap.add_argument("--on_numeric", choices=["raise", "reference"], default="raise",
                    help="a NaN / Inf sample: fail (default) or, like the reference, keep writing nan rows for that file")
    args = ap.parse_args(argv)
    cpc_sd, vap_sd = realtime._load_state_dicts(args.vap_model, args.cpc_model)
    specs = [p.split(":") for p in args.pairs]
    dialogues = [(read_wav_mono(s[0]), read_wav_mono(s[1])) for s in specs]
    vap = realtime.ManyStreamVAP(cpc_sd, vap_sd, args.vap_process_rate, args.context_len_sec, n_streams=len(dialogues))
    for k, rows in enumerate(run_offline(vap, dialogues, on_numeric=args.on_numeric)):
        out = specs[k][2] if len(specs[k]) > 2 else f"output_offline_{k}.txt"
        write_csv(out, rows)
        print("Generated output file: ", out)


if __name__ == "__main__":
    main()
