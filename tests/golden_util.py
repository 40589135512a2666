"""Helpers shared by the parity tests: load a golden case and rebuild its seeded inputs."""
import os

import numpy as np

from vap_realtime_amd import synth, weights as W

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Case:
    def __init__(self, name):
        self.name = name
        self.z = np.load(os.path.join(GOLDEN, f"{name}.npz"))
        z = self.z
        self.mode = str(z["meta.mode"])
        self.frame_hz = int(z["meta.frame_hz"])
        self.ctx_sec = float(z["meta.ctx_sec"])
        self.streams = [int(s) for s in z["meta.streams"]]
        self.n_frames = int(z["meta.n_frames"])
        self.framing = str(z["meta.framing"])
        self.seed = int(z["meta.seed"])
        self.hop = 16000 // self.frame_hz
        self.L = self.hop + 320
        self.T = int(self.ctx_sec * self.frame_hz)
        self.cpc_sd, self.vap_sd = W.synthetic_weights(self.seed, self.frame_hz, self.mode)
        if "meta.cpc_seed" in z and int(z["meta.cpc_seed"]) != self.seed:      # models sharing one cpc_model file
            self.cpc_sd = W.synthetic_weights(int(z["meta.cpc_seed"]), self.frame_hz, "vap")[0]
        fp = W.weights_fingerprint(self.cpc_sd, self.vap_sd)
        assert np.array_equal(fp, z["meta.weights_fp"]), "seeded weights differ from the ones the golden was made with"
        self.audio = synth.dialogue_batch(self.streams, self.hop * self.n_frames + 320)
        self.kinds = [str(k) for k in z["meta.kinds"]] if "meta.kinds" in z.files else None
        if self.kinds:                                   # degenerate / poisoned microphone input, one kind per stream
            self.audio = np.stack([synth.degenerate(self.audio[i], k) for i, k in enumerate(self.kinds)])
        fin = np.where(np.isfinite(self.audio), self.audio, 0.0).astype(np.float64)
        afp = np.array([fin.sum(), np.abs(fin).sum()])
        assert np.allclose(afp, z["meta.audio_fp"], rtol=0, atol=1e-9), "seeded audio differs from the golden's"

    def new_samples(self, f):
        """[S,2,hop] new samples of frame f (server framing)."""
        return self.audio[:, :, f * self.hop:(f + 1) * self.hop]

    def window(self, f):
        """[S,2,L] offline framing window of frame f (vap_offline.py:51-61)."""
        return self.audio[:, :, f * self.hop:f * self.hop + self.L]
