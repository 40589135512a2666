"""Synthetic two-speaker dialogue audio.

The reference ships no WAVs (``input/wav_sample/*.wav`` are listed in ``.MISSING_LARGE_BLOBS``),
so parity tests and the bench drive the path with seeded synthetic audio: per stream two
speakers alternate talk-spurts (on/off Markov chain, anti-correlated), voiced segments are a
harmonic stack with 4 Hz amplitude modulation, silence is low-level noise.  float32 in [-1, 1],
16 kHz.  Deterministic in ``1000 + stream_id``.
"""
from __future__ import annotations

import numpy as np

SR = 16000


def dialogue(stream_id: int, n_samples: int, seed_base: int = 1000) -> np.ndarray:
    """Return float32 ``[2, n_samples]`` for one stream."""
    rng = np.random.default_rng(seed_base + int(stream_id))
    seg = 1600  # 100 ms decision grid for the talk-spurt chain
    n_seg = (n_samples + seg - 1) // seg
    p_on_off = seg / (1.5 * SR)   # mean on 1.5 s
    p_off_on = seg / (1.0 * SR)   # mean off 1.0 s
    state = np.zeros((2, n_seg), dtype=bool)
    s = [bool(rng.integers(0, 2)), False]
    s[1] = not s[0]
    for i in range(n_seg):
        for c in (0, 1):
            u = rng.random()
            other = s[1 - c]
            if s[c]:
                if u < p_on_off:
                    s[c] = False
            else:
                p = p_off_on * (0.2 if other else 1.8)  # anti-correlated turn taking
                if u < p:
                    s[c] = True
        state[0, i], state[1, i] = s
    t = np.arange(n_samples, dtype=np.float64) / SR
    out = np.empty((2, n_samples), dtype=np.float32)
    for c in (0, 1):
        f0 = rng.uniform(100.0, 250.0)
        amps = rng.uniform(0.2, 1.0, size=8) / np.arange(1, 9)
        phases = rng.uniform(0, 2 * np.pi, size=8)
        voiced = np.zeros(n_samples, dtype=np.float64)
        for k in range(8):
            voiced += amps[k] * np.sin(2 * np.pi * (k + 1) * f0 * t + phases[k])
        voiced *= 0.2 / np.max(np.abs(voiced))
        voiced *= 0.6 + 0.4 * np.sin(2 * np.pi * 4.0 * t + rng.uniform(0, 2 * np.pi))
        gate = np.repeat(state[c], seg)[:n_samples].astype(np.float64)
        # 10 ms raised-cosine smoothing of the on/off gate
        k = np.hanning(321); k /= k.sum()
        gate = np.convolve(gate, k, mode="full")[160:160 + n_samples]   # == mode "same", also for clips shorter than the kernel
        noise = 1e-3 * rng.standard_normal(n_samples)
        out[c] = (gate * voiced + noise).astype(np.float32)
    return out


def dialogue_batch(stream_ids, n_samples: int, seed_base: int = 1000) -> np.ndarray:
    """float32 ``[S, 2, n_samples]``."""
    return np.stack([dialogue(s, n_samples, seed_base) for s in stream_ids], axis=0)


def noise_batch(n_streams: int, n_samples: int, seed: int = 7, scale: float = 0.1) -> np.ndarray:
    """Cheap white-noise audio for throughput runs where generating dialogue for thousands of
    streams would dominate set-up time; same shape/dtype as ``dialogue_batch``."""
    rng = np.random.default_rng(seed)
    return (scale * rng.standard_normal((n_streams, 2, n_samples), dtype=np.float32)).astype(np.float32)


# Inputs a live microphone path really produces, as transforms of a seeded dialogue (the degenerate-audio goldens of
# tools/make_golden.py and their parity tests share this one definition).  "nan_sample" / "inf_sample" poison ONE sample of
# channel 0 (index 3 * 800 + 17: frame 3 at 20 Hz): the reference propagates it into the LSTM state for good.
DEGENERATE_KINDS = ("silence", "tiny", "full_scale_square", "loud", "dc_offset", "one_channel_dead")
POISON_KINDS = ("clean", "nan_sample", "inf_sample")
POISON_INDEX = 3 * 800 + 17


def degenerate(base: np.ndarray, kind: str) -> np.ndarray:
    """base float32 [2, n] -> float32 [2, n] of the given kind."""
    n = base.shape[-1]
    t = np.arange(n)
    if kind == "clean":
        return base.copy()
    if kind == "silence":
        return np.zeros_like(base)
    if kind == "tiny":
        return (base * np.float32(1e-30)).astype(np.float32)
    if kind == "full_scale_square":
        return np.broadcast_to(np.where((t // 40) % 2 == 0, 1.0, -1.0).astype(np.float32), base.shape).copy()
    if kind == "loud":
        return (base * np.float32(3e3)).astype(np.float32)
    if kind == "dc_offset":
        return (base + np.float32(0.75)).astype(np.float32)
    if kind == "one_channel_dead":
        out = base.copy()
        out[1] = 0.0
        return out
    if kind in ("nan_sample", "inf_sample"):
        out = base.copy()
        out[0, POISON_INDEX] = np.nan if kind == "nan_sample" else np.inf
        return out
    raise ValueError(kind)
