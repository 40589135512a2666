"""Co-running diagnostic: a victim engine steps while an aggressor keeps the GPU busy — another engine on the null stream of
this process, an engine in ANOTHER PROCESS, or the register-only matrix-core burner tools/mfma_aggr — and must stay
bit-identical to a run of the same engine that had the GPU to itself (DESIGN.md "co-running f16 / bf16 MFMA").
usage: diag_conc.py S TICKS victim=<one|grp|grpdefer> aggr=<none|fp32|split|unfconv|unfproj|fp32other|proc|procfp32|proctorch|
       mfmaf16|mfmabf16|mfmaf32> [hz=20 ctx=2.5 emode=vap vflags=flag,flag] [pertick] [sync]"""
import subprocess
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from vap_realtime_amd import engine, synth, weights as W

S = int(sys.argv[1])
F_ = int(sys.argv[2])
opts = dict(a.split("=") for a in sys.argv[3:] if "=" in a)
victim = opts.get("victim", "one")
aggr = opts.get("aggr", "split")
HZ = int(opts.get("hz", 20)); CTX = float(opts.get("ctx", 2.5 if HZ == 20 else 5.0)); EMODE = opts.get("emode", "vap")
HOP = 16000 // HZ
cpc, vap = W.synthetic_weights(3, HZ, mode=EMODE) if EMODE != "vap" else W.synthetic_weights(3, HZ)
blob = W.pack_blob(cpc, vap, EMODE) if EMODE != "vap" else W.pack_blob(cpc, vap)
NF = 16
audio = torch.from_numpy(np.concatenate([synth.dialogue_batch(list(range(64)), HOP * NF)] * ((S + 63) // 64))[:S]).cuda()
frames = [audio[:, :, k * HOP:(k + 1) * HOP].contiguous() for k in range(NF)]
vflags = {k: True for k in opts.get("vflags", "").split(",") if k}

if aggr == "childtorch":
    a = torch.randn(4096, 4096, device="cuda"); print("child running", flush=True)
    while True:
        for _ in range(20):
            b = a @ a
        torch.cuda.synchronize()
if aggr.startswith("child"):       # we ARE the aggressor process: loop an engine until killed
    e = engine.Engine(blob, HZ, CTX, max_streams=S, mode=EMODE, split_f16=(aggr == "child"))
    o = torch.zeros(S, engine.OUT_STRIDE, device="cuda")
    print("child running", flush=True)
    t = 0
    while True:
        e.step_device(S, frames[t % NF].data_ptr(), HOP, o.data_ptr(), stream=0)
        t += 1
        if t % 8 == 0:
            torch.cuda.synchronize()

ref_eng = engine.Engine(blob, HZ, CTX, max_streams=S, mode=EMODE, **vflags)
tmp = torch.zeros(S, engine.OUT_STRIDE, device="cuda")
ref = {}
for t in range(F_):
    ref_eng.step_device(S, frames[t % NF].data_ptr(), HOP, tmp.data_ptr(), stream=0)
    torch.cuda.synchronize(); ref[t] = tmp.clone()
torch.cuda.synchronize()

vic = engine.Engine(blob, HZ, CTX, max_streams=S, mode=EMODE, groups=(1 if victim == "one" else 2), **vflags)
kw = {"none": None, "fp32": {}, "fp32other": {}, "split": {"split_f16": True}, "unfconv": {"unfused_conv": True},
      "unfproj": {"unfused_proj": True}, "proc": None, "procfp32": None, "proctorch": None}.get(aggr)
agg = engine.Engine(blob, HZ, CTX, max_streams=S, mode=EMODE, **kw) if kw is not None else None
child = None
if aggr.startswith("mfma"):      # tools/mfma_aggr <f16|f32|bf16> in another process
    child = subprocess.Popen([__file__.rsplit("/", 1)[0] + "/mfma_aggr", aggr[4:], "600"], stdout=subprocess.PIPE, text=True)
    child.stdout.readline()
    time.sleep(0.5)
elif aggr.startswith("proc"):
    import os
    cenv = {k: v for k, v in os.environ.items() if k not in ("VAPX_FORCE_LONG", "VAPX_FFN_TILE")}
    child = subprocess.Popen([sys.executable, __file__, str(S), "0", "aggr=child" + aggr[4:]] + [a for a in sys.argv[3:] if a.split("=")[0] in ("hz", "ctx", "emode")], stdout=subprocess.PIPE, text=True, env=cenv)
    child.stdout.readline()
    time.sleep(1.0)
vs = torch.cuda.Stream()
ov = torch.zeros(S, engine.OUT_STRIDE, device="cuda")
ovs = [torch.zeros(S, engine.OUT_STRIDE, device="cuda") for _ in range(F_)] if "pertick" in sys.argv else None
oa = torch.zeros(S, engine.OUT_STRIDE, device="cuda")
for t in range(F_):
    if ovs is not None:
        ov = ovs[t]
    vic.step_device(S, frames[t % NF].data_ptr(), HOP, ov.data_ptr(), stream=vs.cuda_stream, defer_join=(victim == "grpdefer"))
    if agg is not None:
        agg.step_device(S, frames[(t + (5 if aggr == "fp32other" else 0)) % NF].data_ptr(), HOP, oa.data_ptr(), stream=0)
    if "sync" in sys.argv:
        torch.cuda.synchronize()
    if t % 25 == 24:
        vic.join(vs.cuda_stream); torch.cuda.synchronize()
        d = (ov[:, :272] - ref[t][:, :272]).abs().max(dim=1).values
        print(f"victim={victim} aggr={aggr} tick {t}: victim vs ref bad {int((d > 0).sum())} max {float(d.max()):.3g}", end="")
        if agg is not None and aggr != "fp32other":
            print(f"   aggressor vs ref max {float((oa[:, :6] - ref[t][:, :6]).abs().max()):.3g}", end="")
        print()
vic.join(vs.cuda_stream)
torch.cuda.synchronize()
if child is not None:
    child.kill()
if ovs is not None:
    nbad = 0
    for t in range(F_):
        d = (ovs[t] - ref[t]).abs()
        rows = torch.nonzero(~(d.max(dim=1).values <= 0)).flatten().tolist()
        if rows:
            nbad += 1
            cols = torch.nonzero(~(d.max(dim=0).values <= 0)).flatten().tolist()
            seg = {"p_now..aux(0-15)": [c for c in cols if c < 16], "logits": len([c for c in cols if 16 <= c < 272]), "e": len([c for c in cols if c >= 272])}
            if nbad <= 6:
                print(f"tick {t}: {len(rows)} bad streams, first {rows[:6]} last {rows[-3:]} in group0 {sum(r < S // 2 for r in rows)}; cols {seg}; max {float(d.max()):.3g}")
    print(f"bad ticks: {nbad} of {F_}")
