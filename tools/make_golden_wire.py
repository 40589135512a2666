#!/usr/bin/env python3
"""Golden byte vectors of the reference wire codec (rvap/common/util.py), produced by importing the
unmodified reference.  Output: tests/golden/wire.npz (inputs + expected bytes as uint8 arrays)."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference")
import rvap.common.util as util  # noqa: E402

rng = np.random.default_rng(99)
x1 = rng.standard_normal(160)
x2 = rng.standard_normal(160)
out = {"in.x1": x1, "in.x2": x2, "in.bytes": np.frombuffer(util.conv_2floatarray_2_bytearray(x1, x2), np.uint8)}
a1, a2 = util.conv_bytearray_2_2floatarray(bytes(out["in.bytes"]))
assert np.array_equal(a1, x1) and np.array_equal(a2, x2)

n = 800
res = {"t": 1727481600.123456, "x1": rng.standard_normal(n).tolist(), "x2": rng.standard_normal(n).tolist(),
       "p_now": [0.25, 0.75], "p_future": [0.6, 0.4], "vad": [0.9, 0.1]}
out["vap.t"] = np.array(res["t"]); out["vap.x1"] = np.array(res["x1"]); out["vap.x2"] = np.array(res["x2"])
out["vap.p_now"] = np.array(res["p_now"]); out["vap.p_future"] = np.array(res["p_future"]); out["vap.vad"] = np.array(res["vad"])
out["vap.bytes"] = np.frombuffer(util.conv_vapresult_2_bytearray(res), np.uint8)
bc = dict(res, p_bc_react=[0.3], p_bc_emo=[0.05])
out["bc.p_bc_react"] = np.array(bc["p_bc_react"]); out["bc.p_bc_emo"] = np.array(bc["p_bc_emo"])
out["bc.bytes"] = np.frombuffer(util.conv_vapresult_2_bytearray_bc(bc), np.uint8)
nod = dict(res, p_bc=rng.random(50).tolist(), p_nod_short=[0.2], p_nod_long=[0.1], p_nod_long_p=[0.05])
for k in ("p_bc", "p_nod_short", "p_nod_long", "p_nod_long_p"):
    out["nod." + k] = np.array(nod[k])
out["nod.bytes"] = np.frombuffer(util.conv_vapresult_2_bytearray_nod(nod), np.uint8)
np.savez_compressed(os.path.join(REPO, "tests", "golden", "wire.npz"), **out)
print({k: v.shape for k, v in out.items() if k.endswith("bytes")})
