"""ORACLE — CPU restatement of the reference's streaming VAP step.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; the product path (``vap-realtime_amd/``) never does and fails loudly when its HIP
library is missing.

What it restates (all citations relative to /root/reference):
  * ``VAPRealTime.process_vap``                      rvap/vap_main/vap_main.py:249-335
  * ``EncoderCPC.forward``                            rvap/vap_main/encoder.py:58-80
  * ``CPCEncoder.forward`` / ``ChannelNorm.forward``  rvap/vap_main/encoder_components.py:37-104
  * ``CPCAR.forward`` (LSTM, keepHidden)              rvap/vap_main/encoder_components.py:107-159
  * downsample ``get_cnn_layer``                      rvap/vap_main/encoder_components.py:496-511
  * ``MultiHeadAttentionAlibi`` / ``TransformerLayer`` / ``TransformerStereoLayer`` /
    ``GPT`` / ``GPTStereo`` / ``Combinator``          rvap/vap_main/modules.py:24-464
  * ``ObjectiveVAP.probs_next_speaker_aggregate``     rvap/vap_main/objective.py:186-206
  * bc / nod heads                                    rvap/vap_bc/vap_bc_main.py:272-277,
                                                      rvap/vap_nod/vap_nod_main.py:273-279

Arithmetic is fp32 on PyTorch-CPU (the same ATen kernels the reference runs on), written as a
pure function of explicit per-stream state and batched over S independent streams.

Pinning: the reference has no tests or golden vectors for this path and its checkpoints/WAVs
are absent, so the oracle is pinned against outputs of the *imported,
unmodified reference itself* run in the build container on seeded synthetic weights and audio:
``tools/make_golden.py`` writes ``tests/golden/*.npz`` and ``tests/test_oracle_golden.py``
checks this file against them (<= 2e-6 abs).  Real-checkpoint parity is unpinned (assets absent).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

DIM = 256
HEADS = 4
PAD = 320  # frame_contxt_padding, vap_main.py:224


def _t(x, dtype=torch.float32) -> torch.Tensor:
    if isinstance(x, torch.Tensor):
        return x.detach().to(dtype).cpu()
    return torch.from_numpy(np.ascontiguousarray(np.asarray(x, dtype=np.float32))).to(dtype)


@dataclass
class OracleState:
    """Per-stream state the reference keeps as Python attributes: embedding context lists
    (vap_main.py:243-244), LSTM (h, c) (encoder_components.py:148-153)."""
    n_streams: int
    ctx_len: int
    ring: List[torch.Tensor] = field(default_factory=list)      # list (<=T) of [S,2,256]
    h: Optional[torch.Tensor] = None                             # [S,2,256]
    c: Optional[torch.Tensor] = None

    def clone(self) -> "OracleState":
        return OracleState(self.n_streams, self.ctx_len, [r.clone() for r in self.ring],
                           None if self.h is None else self.h.clone(),
                           None if self.c is None else self.c.clone())


class VapOracle:
    def __init__(self, cpc_sd: Dict[str, np.ndarray], vap_sd: Dict[str, np.ndarray],
                 frame_hz: int = 20, context_len_sec: float = 2.5, mode: str = "vap", dtype=torch.float32):
        # dtype = torch.float64 turns the restatement into a ground truth for rounding-error comparisons between the
        # HIP arithmetic variants (tests/test_split_precision_gpu.py); the reference itself computes in float32
        self.dtype = dtype
        self.w = {k: _t(v, dtype) for k, v in cpc_sd.items()}
        self.v = {k: _t(v, dtype) for k, v in vap_sd.items()}
        self.frame_hz = frame_hz
        self.T = int(context_len_sec * frame_hz)                 # vap_main.py:221
        self.hop = 16000 // frame_hz
        self.L = self.hop + PAD                                  # vap_main.py:230
        self.mode = mode
        # aggregation tables, objective.py:93-110,141-143,196-201: states[i,c,b] = bit(4c+b)
        idx = torch.arange(256)
        bits = ((idx[:, None] >> torch.arange(8)[None, :]) & 1).to(dtype).view(256, 2, 4)
        self.abp_now = bits[:, :, 0:2].sum(-1)                   # BINS_P_NOW = [0,1]  vap_main.py:187
        self.abp_fut = bits[:, :, 2:4].sum(-1)                   # BINS_PFUTURE = [2,3]

    def new_state(self, n_streams: int) -> OracleState:
        return OracleState(n_streams, self.T)

    # ---- encoder ------------------------------------------------------------------------------
    def cnn(self, x: torch.Tensor, collect: Optional[dict] = None) -> torch.Tensor:
        """x [B,1,L] -> [B,256,P4]; encoder_components.py:98-104 with ChannelNorm 64-70."""
        w = self.w
        for i, (s, p) in enumerate(((5, 3), (4, 2), (2, 1), (2, 1), (2, 1))):
            x = F.conv1d(x, w[f"gEncoder.conv{i}.weight"], w[f"gEncoder.conv{i}.bias"], stride=s, padding=p)
            mean = x.mean(dim=1, keepdim=True)
            var = x.var(dim=1, keepdim=True)                     # unbiased (N-1) — :65
            x = (x - mean) * torch.rsqrt(var + 1e-5)
            x = x * w[f"gEncoder.batchNorm{i}.weight"] + w[f"gEncoder.batchNorm{i}.bias"]
            x = F.relu(x)
            if collect is not None:
                collect[f"cnn{i}"] = x
        return x

    def lstm(self, z: torch.Tensor, h: torch.Tensor, c: torch.Tensor):
        """z [B,n,256]; gate order i,f,g,o; both biases (torch nn.LSTM semantics)."""
        w = self.w
        wih, whh = w["gAR.baseNet.weight_ih_l0"], w["gAR.baseNet.weight_hh_l0"]
        b = w["gAR.baseNet.bias_ih_l0"] + w["gAR.baseNet.bias_hh_l0"]
        outs = []
        for t in range(z.shape[1]):
            g = z[:, t] @ wih.T + h @ whh.T + b
            i, f, gg, o = g.split(DIM, dim=1)
            c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
            h = torch.sigmoid(o) * torch.tanh(c)
            outs.append(h)
        return torch.stack(outs, dim=1), h, c

    def downsample(self, y: torch.Tensor) -> torch.Tensor:
        """y [B,K,256] -> [B,256]: Conv1d(256,256,K) single output + LN + exact GELU."""
        v = self.v
        wd = v["encoder.downsample.1.weight"]                    # [256,256,K]
        out = torch.einsum("bkc,ock->bo", y, wd) + v["encoder.downsample.1.bias"]
        out = F.layer_norm(out, (DIM,), v["encoder.downsample.2.ln.weight"], v["encoder.downsample.2.ln.bias"], 1e-5)
        return F.gelu(out)

    def encode(self, audio: torch.Tensor, st: OracleState, collect: Optional[dict] = None) -> torch.Tensor:
        """audio [S,2,L] -> e [S,2,256]; the two channels share weights but own (h,c)
        (vap_main.py:144-169: encoder1/encoder2 built from the same checkpoint)."""
        S = audio.shape[0]
        x = audio.reshape(S * 2, 1, self.L)
        z = self.cnn(x, collect)                                  # [S*2,256,P4]
        z = z.transpose(1, 2)[:, 1:-1, :]                         # encoder.py:75-76
        if st.h is None:
            st.h = torch.zeros(S, 2, DIM, dtype=self.dtype)
            st.c = torch.zeros(S, 2, DIM, dtype=self.dtype)
        y, h, c = self.lstm(z, st.h.reshape(S * 2, DIM), st.c.reshape(S * 2, DIM))
        st.h, st.c = h.reshape(S, 2, DIM), c.reshape(S, 2, DIM)
        e = self.downsample(y).reshape(S, 2, DIM)
        if collect is not None:
            collect["z"] = z.reshape(S, 2, -1, DIM)
            collect["lstm_out"] = y.reshape(S, 2, -1, DIM)
            collect["e"] = e
        return e

    # ---- transformer --------------------------------------------------------------------------
    def _mha(self, pre: str, q_in: torch.Tensor, kv_in: torch.Tensor) -> torch.Tensor:
        """modules.py:82-110 + ALiBi mask 162-212.  [B,n,256]."""
        v = self.v
        B, n, _ = q_in.shape
        q = (q_in @ v[f"{pre}.query.weight"].T).view(B, n, HEADS, 64).transpose(1, 2)
        k = (kv_in @ v[f"{pre}.key.weight"].T).view(B, n, HEADS, 64).transpose(1, 2)
        val = (kv_in @ v[f"{pre}.value.weight"].T).view(B, n, HEADS, 64).transpose(1, 2)
        att = torch.einsum("bhid,bhjd->bhij", q, k) * (1.0 / math.sqrt(DIM))   # scale 1/16, :52
        m = v[f"{pre}.m"].view(1, HEADS, 1, 1)
        j = torch.arange(n, dtype=self.dtype).view(1, 1, 1, n)
        causal = torch.full((n, n), float("-inf")).triu(1)
        att = att + (m * j + causal)
        att = att.softmax(dim=-1)
        y = (att @ val).transpose(1, 2).reshape(B, n, DIM)
        return y @ v[f"{pre}.proj.weight"].T

    def _ln(self, x, name):
        return F.layer_norm(x, (DIM,), self.v[f"{name}.weight"], self.v[f"{name}.bias"], 1e-5)

    def layer(self, pre: str, x: torch.Tensor, src: Optional[torch.Tensor]) -> torch.Tensor:
        """TransformerLayer.forward, modules.py:257-286 (dropout off in eval)."""
        v = self.v
        z = self._ln(x, f"{pre}.ln_self_attn")
        x = x + self._mha(f"{pre}.mha", z, z)
        if src is not None:
            z = self._ln(x, f"{pre}.ln_src_attn")
            x = x + self._mha(f"{pre}.mha_cross", z, src)         # src NOT normalised, :276-283
        z = self._ln(x, f"{pre}.ln_ffnetwork")
        x = x + F.gelu(z @ v[f"{pre}.ffnetwork.0.weight"].T) @ v[f"{pre}.ffnetwork.3.weight"].T
        return x

    def transformer(self, x1: torch.Tensor, x2: torch.Tensor, collect: Optional[dict] = None):
        """x1,x2 [S,n,256] -> (o1,o2,a,b,h)."""
        o1 = self.layer("ar_channel.layers.0", x1, None)          # vap_main.py:285-286
        o2 = self.layer("ar_channel.layers.0", x2, None)
        a, b = o1, o2
        if collect is not None:
            collect["o"] = torch.stack([o1, o2], 1)
        for l in range(3):                                        # GPTStereo.forward, modules.py:395-412
            a, b = self.layer(f"ar.layers.{l}", a, b), self.layer(f"ar.layers.{l}", b, a)
            if collect is not None:
                collect[f"stereo{l}"] = torch.stack([a, b], 1)
        v = self.v
        ha = F.gelu(self._ln(a @ v["ar.combinator.h0_a.weight"].T, "ar.combinator.ln"))
        hb = F.gelu(self._ln(b @ v["ar.combinator.h0_b.weight"].T, "ar.combinator.ln"))
        h = ha + hb                                               # modules.py:449-464
        if collect is not None:
            collect["comb"] = h
        return o1, o2, a, b, h

    # ---- full step ----------------------------------------------------------------------------
    def step(self, audio, st: OracleState, collect: Optional[dict] = None) -> Dict[str, np.ndarray]:
        """One VAP frame for S streams.  audio: float [S,2,L] (carry + new samples, exactly what
        ``process_vap`` receives as x1/x2).  Returns numpy arrays."""
        audio = _t(audio, self.dtype)
        with torch.no_grad():
            e = self.encode(audio, st, collect)
            st.ring.append(e)
            if len(st.ring) > self.T:
                st.ring = st.ring[-self.T:]                       # vap_main.py:277-280
            X = torch.stack(st.ring, dim=2)                       # [S,2,n,256]
            o1, o2, a, b, h = self.transformer(X[:, 0], X[:, 1], collect)
            v = self.v
            out: Dict[str, np.ndarray] = {}
            out["vad"] = torch.stack([
                torch.sigmoid(o1[:, -1] @ v["va_classifier.weight"].T + v["va_classifier.bias"])[:, 0],
                torch.sigmoid(o2[:, -1] @ v["va_classifier.weight"].T + v["va_classifier.bias"])[:, 0],
            ], dim=1).numpy()                                     # vap_main.py:292-293,313-314
            if "vap_head.weight" in v:
                logits = h[:, -1] @ v["vap_head.weight"].T + v["vap_head.bias"]
                probs = logits.softmax(dim=-1)
                pn = probs @ self.abp_now
                pf = probs @ self.abp_fut
                pn = pn / (pn.sum(-1, keepdim=True) + 1e-5)        # objective.py:203-205
                pf = pf / (pf.sum(-1, keepdim=True) + 1e-5)
                out["logits"] = logits.numpy()
                out["p_now"] = pn.numpy()
                out["p_future"] = pf.numpy()
            if self.mode == "bc":                                 # vap_bc_main.py:272-277
                bc = (h[:, -1] @ v["bc_head.weight"].T + v["bc_head.bias"]).softmax(-1)
                out["p_bc_react"] = bc[:, 1].numpy()
                out["p_bc_emo"] = bc[:, 2].numpy()
            elif self.mode == "nod":                              # vap_nod_main.py:273-279
                nod = (h[:, -1] @ v["nod_head.weight"].T + v["nod_head.bias"]).softmax(-1)
                out["p_nod_short"] = nod[:, 1].numpy()
                out["p_nod_long"] = nod[:, 2].numpy()
                out["p_nod_long_p"] = nod[:, 3].numpy()
                # quirk: `p_bc.sigmoid()[-1]` indexes the batch dim -> all n rows are emitted
                out["p_bc"] = torch.sigmoid(h @ v["bc_head.weight"].T + v["bc_head.bias"])[..., 0].numpy()  # [S,n]
            out["e"] = e.numpy()
        return out


class ServerFramer:
    """Carry logic of ``proc_serv_in`` (vap_main.py:368-409): frames are [carry(320) | new hop
    samples]; the carry starts as zeros and becomes the last 320 samples of each frame."""

    def __init__(self, n_streams: int, hop: int):
        self.carry = np.zeros((n_streams, 2, PAD), dtype=np.float32)
        self.hop = hop

    def frame(self, new: np.ndarray) -> np.ndarray:
        buf = np.concatenate([self.carry, np.asarray(new, dtype=np.float32)], axis=2)
        self.carry = buf[:, :, -PAD:].copy()
        return buf
