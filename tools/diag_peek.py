"""Which layer / rows go wrong?  A victim engine steps next to an aggressor PROCESS; after every tick the layer outputs
(o, stereo0, stereo1) are peeked and compared row by row with a run of the same engine that had the GPU to itself.
usage: diag_peek.py S TICKS aggr=<proc|mfmaf16|none>"""
import os
import subprocess
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from vap_realtime_amd import engine, synth, weights as W

S = int(sys.argv[1]); F_ = int(sys.argv[2])
opts = dict(a.split("=") for a in sys.argv[3:] if "=" in a)
aggr = opts.get("aggr", "proc")
T = 50
cpc, vap = W.synthetic_weights(3, 20)
blob = W.pack_blob(cpc, vap)
NF = 16
audio = torch.from_numpy(np.concatenate([synth.dialogue_batch(list(range(64)), 800 * NF)] * ((S + 63) // 64))[:S]).cuda()
frames = [audio[:, :, k * 800:(k + 1) * 800].contiguous() for k in range(NF)]
NAMES = ("o", "stereo0", "stereo1")


def run(with_aggr):
    eng = engine.Engine(blob, 20, 2.5, max_streams=S)
    out = torch.zeros(S, engine.OUT_STRIDE, device="cuda")
    child = None
    if with_aggr and aggr != "none":
        here = __file__.rsplit("/", 1)[0]
        cmd = [sys.executable, here + "/diag_conc.py", str(S), "0", "aggr=child"] if aggr == "proc" else [here + "/mfma_aggr", aggr[4:], "120"]
        child = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True)
        child.stdout.readline(); time.sleep(1.0)
    recs = []
    for t in range(F_):
        eng.step_device(S, frames[t % NF].data_ptr(), 800, out.data_ptr(), stream=0)
        torch.cuda.synchronize()
        rec = {k: eng.peek(k, (S, 2, T, 256)).copy() for k in NAMES}
        rec["out"] = out.cpu().numpy().copy()
        recs.append(rec)
    if child is not None:
        child.kill()
    eng.close()
    return recs


ref = run(False)
got = run(True)
nbad = 0
for t in range(F_):
    n = min(t + 1, T)
    line = []
    anybad = False
    for k in NAMES:
        d = np.abs(got[t][k][:, :, :n] - ref[t][k][:, :, :n]).max(axis=-1)       # [S,2,n]
        bad = np.argwhere(~(d <= 1e-4))
        if len(bad):
            anybad = True
            streams = sorted(set(bad[:, 0].tolist()))
            rows = sorted(set(bad[:, 2].tolist()))
            ex = bad[0].tolist()
            cols = np.nonzero(~(np.abs(got[t][k][ex[0], ex[1], ex[2]] - ref[t][k][ex[0], ex[1], ex[2]]) <= 1e-4))[0]
            line.append(f"{k}: {len(bad)} bad rows in {len(streams)} streams (rows {rows[0]}..{rows[-1]}, {len(rows)} distinct; e.g. stream {ex[0]} ch {ex[1]} row {ex[2]} "
                        f"cols {cols[:4].tolist()}..{cols[-1] if len(cols) else -1} n={len(cols)})")
        else:
            line.append(f"{k}: ok")
    do = np.abs(got[t]["out"][:, :272] - ref[t]["out"][:, :272]).max(axis=1)
    nb = int((~(do <= 1e-4)).sum())
    if anybad or nb:
        nbad += 1
        if nbad <= 8:
            print(f"tick {t} (n={n}): out bad streams {nb} | " + " | ".join(line))
print(f"bad ticks {nbad} of {F_}")
