"""Nothing the engine consumes is stale: with VAPX_POISON_SCRATCH every scratch buffer is refilled with NaN bit patterns before each
step and the context rings start as NaNs, so a kernel that reads a row, a padding lane or a window slot which no kernel of the same
tick (or no earlier frame) wrote turns the outputs non-finite.  The goldens must still be met."""
import os

import numpy as np
import pytest

from golden_util import Case

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.mark.parametrize("name,keys,kw", [
    ("multi3", ("p_now", "p_future", "vad", "logits"), {}),
    ("vap50", ("p_now", "p_future", "vad", "logits"), {}),                       # long-window chain, window fills and slides
    ("vap50", ("p_now", "p_future", "vad", "logits"), {"unfused_proj": True}),   # GEMM-chain variant
    ("nod20", ("e",), {}),                                                         # full last layer + all-rows combinator
    ("multi3", ("p_now", "p_future", "vad", "logits"), {"split_f16": True}),
    ("multi3", ("p_now", "p_future", "vad", "logits"), {"unfused_last_row": True, "unfused_conv": True, "materialize_x0": True}),
])
def test_goldens_with_poisoned_scratch(name, keys, kw):
    from vap_realtime_amd import engine, weights as W
    c = Case(name)
    os.environ["VAPX_POISON_SCRATCH"] = "1"
    try:
        eng = engine.Engine(W.pack_blob(c.cpc_sd, c.vap_sd, c.mode), c.frame_hz, c.ctx_sec, max_streams=len(c.streams) + 2, mode=c.mode, **kw)
    finally:
        del os.environ["VAPX_POISON_SCRATCH"]
    ids = np.arange(len(c.streams), dtype=np.int32)[::-1].copy() + 1            # not the identity mapping, slot 0 stays unused
    order = ids - 1
    worst = 0.0
    for f in range(c.n_frames):
        raw = eng.step(np.ascontiguousarray(c.new_samples(f)[order]), ids)
        assert np.isfinite(raw).all(), f"{name} frame {f}: non-finite outputs with poisoned scratch"
        o = engine.split_outputs(raw)
        for k in keys:
            want = c.z[k][f][order]
            worst = max(worst, float(np.abs(o[k].reshape(want.shape) - want).max()))
    eng.close()
    assert worst <= TOL, (name, kw, worst)


def test_no_out_of_bounds_writes_around_any_engine_buffer():
    """VAPX_GUARD_ZONES: 4 KiB canary zones on both sides of every device allocation of the engine; after full-window runs of the
    short-window, long-window and nod paths at a batch that is not a multiple of any tile size, no canary byte has changed."""
    import subprocess
    import sys
    code = r'''
import sys, numpy as np
sys.path.insert(0, ".")
from vap_realtime_amd import engine, synth, weights as W
total = 0
for hz, ctx, mode, S in ((20, 2.5, "vap", 77), (50, 5.0, "vap", 21), (20, 2.5, "nod", 33), (10, 5.0, "bc", 19), (20, 2.5, "vap", 1101)):
    cpc, vap = W.synthetic_weights(3, hz, mode=mode)
    for kw in ({}, {"groups": 2}, {"split_f16": True}, {"unfused_last_row": True, "unfused_conv": True, "materialize_x0": True}):
        if S > 1000 and kw:
            continue
        eng = engine.Engine(W.pack_blob(cpc, vap, mode), hz, ctx, max_streams=S, mode=mode, **kw)
        hop = 16000 // hz
        audio = synth.dialogue_batch(list(range(S)), hop * 4)
        for t in range(int(ctx * hz) + 3):
            out = eng.step(np.ascontiguousarray(audio[:, :, (t % 4) * hop:(t % 4 + 1) * hop]))
        assert np.isfinite(out).all()
        v = eng.peek("guard_violations", (1,))[0]
        assert v >= 0, "guard zones not enabled"
        total += int(v)
        assert v == 0, (hz, mode, S, kw, v)
        eng.close()
print("guard zones intact:", total)
'''
    env = dict(os.environ, VAPX_GUARD_ZONES="1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=env,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "guard zones intact: 0" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_create_destroy_does_not_leak_device_memory():
    """40 engines of every flavour created, stepped and destroyed: the free device memory afterwards is what it was before
    (events, streams, pinned staging and the trunk followers' dropped buffers included)."""
    import torch
    from vap_realtime_amd import engine, synth, weights as W

    def cycle(k):
        mode = ("vap", "bc", "nod")[k % 3]
        hz = (20, 50, 10)[k % 3]
        cpc, vap = W.synthetic_weights(3, hz, mode=mode)
        eng = engine.Engine(W.pack_blob(cpc, vap, mode), hz, 2.5, max_streams=64, mode=mode, groups=(k % 2) * 2, split_f16=bool(k % 4 == 3))
        hop = 16000 // hz
        audio = synth.dialogue_batch(list(range(64)), hop)
        for _ in range(3):
            eng.step(audio)
        eng.close()

    for k in range(12):                     # warm the allocator / code objects: every (mode, rate, groups, split) combination of the cycle once
        cycle(k)
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for k in range(40):
        cycle(k)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 32 << 20, f"device memory shrank by {(free0 - free1) >> 20} MiB over 40 create/destroy cycles"


def test_out_of_memory_at_create_is_an_error_code_and_leaks_nothing():
    """4 M stream slots need 400 GB of context rings: vapx_create must come back with VAPX_E_NOMEM (-4 family, not a crash) after releasing
    whatever it had already allocated, and a normal engine must still work afterwards."""
    import torch
    from vap_realtime_amd import engine, synth, weights as W
    cpc, vap = W.synthetic_weights(3, 20)
    blob = W.pack_blob(cpc, vap)
    for _ in range(2):                       # warm the runtime (code objects, its own pools) with a normal create + a failed one
        engine.Engine(blob, 20, 2.5, max_streams=8).close()
        with pytest.raises(engine.VapxError):
            engine.Engine(blob, 20, 2.5, max_streams=4_000_000, max_batch=64)
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(3):
        with pytest.raises(engine.VapxError) as ei:
            engine.Engine(blob, 20, 2.5, max_streams=4_000_000, max_batch=64)
        assert "memory" in str(ei.value).lower(), str(ei.value)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 64 << 20, f"{(free0 - free1) >> 20} MiB lost after three failed creates"
    eng = engine.Engine(blob, 20, 2.5, max_streams=8)
    out = eng.step(synth.dialogue_batch(list(range(8)), 800))
    assert np.isfinite(out).all()
    eng.close()


def test_engines_stepped_from_concurrent_host_threads():
    """One handle per thread is the contract (vapx.h): four threads, each creating, stepping and destroying its own engine at the same
    time (ctypes drops the GIL inside every call), must produce what the same four engines produce one after the other."""
    import threading
    from vap_realtime_amd import engine, synth, weights as W
    hz_of = (20, 50, 10, 20)
    blobs, audios = [], []
    for k, hz in enumerate(hz_of):
        cpc, vap = W.synthetic_weights(40 + k, hz)
        blobs.append(W.pack_blob(cpc, vap))
        audios.append(synth.dialogue_batch(list(range(48)), (16000 // hz) * 6))

    def run(k, sink):
        hz = hz_of[k]
        hop = 16000 // hz
        eng = engine.Engine(blobs[k], hz, 2.5, max_streams=48, groups=(k % 2) * 2)
        outs = [eng.step(np.ascontiguousarray(audios[k][:, :, (t % 6) * hop:(t % 6 + 1) * hop])).copy() for t in range(40)]
        eng.close()
        sink[k] = np.stack(outs)

    solo, conc = {}, {}
    for k in range(4):
        run(k, solo)
    errors = []

    def guarded(k):
        try:
            run(k, conc)
        except Exception as e:          # noqa: BLE001 - reported below
            errors.append((k, repr(e)))
    threads = [threading.Thread(target=guarded, args=(k,)) for k in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for k in range(4):
        assert np.array_equal(solo[k], conc[k]), f"engine {k} differs when stepped next to three other threads"


def test_device_path_is_hip_graph_capturable():
    """The all-device vapx_step does nothing a stream capture forbids (no synchronisation, no allocation, no host staging): it can be
    captured into a HIP graph once and replayed every tick, with the same bits as launching it.  (Measured: no speed-up — a tick is bound
    by kernel latency, not by launches — but a caller that builds graphs around the engine can include it.)"""
    import torch
    from vap_realtime_amd import engine, synth, weights as W
    cpc, vap = W.synthetic_weights(3, 20)
    blob = W.pack_blob(cpc, vap)
    S = 24
    a, b = engine.Engine(blob, 20, 2.5, max_streams=S), engine.Engine(blob, 20, 2.5, max_streams=S)
    audio = torch.from_numpy(synth.dialogue_batch(list(range(S)), 800 * 8)).cuda()
    frames = [audio[:, :, k * 800:(k + 1) * 800].contiguous() for k in range(8)]
    cur = torch.zeros_like(frames[0])
    oa, ob = torch.zeros(S, engine.OUT_STRIDE, device="cuda"), torch.zeros(S, engine.OUT_STRIDE, device="cuda")
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for t in range(3):                                   # first-use paths (kernel attributes) run outside the capture
            cur.copy_(frames[t])
            b.step_device(S, cur.data_ptr(), 800, ob.data_ptr(), stream=side.cuda_stream)
            a.step_device(S, frames[t].data_ptr(), 800, oa.data_ptr(), stream=side.cuda_stream)
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            b.step_device(S, cur.data_ptr(), 800, ob.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for t in range(3, 60):                                   # through the window fill and into the sliding regime
        cur.copy_(frames[t % 8])
        g.replay()
        a.step_device(S, frames[t % 8].data_ptr(), 800, oa.data_ptr(), stream=0)
        torch.cuda.synchronize()
        assert torch.equal(oa, ob), f"replayed tick {t} differs from the launched one"
    del g
    a.close(); b.close()
