"""CPU-side checks: weight packing, blob layout shared with the C side, C-ABI symbol coverage,
loud failure without a GPU, synthetic-input determinism, stream sharding incl. a 2-process gloo run."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from vap_realtime_amd import engine, sharding, synth, weights as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_geometry_per_rate():
    assert [W.cpc_frames_for_rate(hz) for hz in (50, 20, 10, 5)] == [2, 5, 10, 20]


def test_blob_layout_and_pack_inverse():
    cpc, vap = W.synthetic_weights(3, 20, "nod")
    blob = W.pack_blob(cpc, vap, "nod")
    lay = W.blob_layout(5)
    assert blob.size == lay["__total__"][0]
    assert all(off % 64 == 0 for off, _ in lay.values())
    get = lambda k: blob[lay[k][0]:lay[k][0] + lay[k][1]]
    # conv weights are [cout][tap][cin]
    w1 = get("conv1.w").reshape(256, 8, 256)
    np.testing.assert_array_equal(w1.transpose(0, 2, 1), cpc["gEncoder.conv1.weight"])
    # QKV stacking order is query, key, value
    wqkv = get("L2.wqkv").reshape(768, 256)
    np.testing.assert_array_equal(wqkv[256:512], vap["ar.layers.1.mha.key.weight"])
    # LSTM: undo fragment-major packing + gate permutation
    wf = get("lstm.whh").reshape(8, 16, 8, 4, 16, 4)           # [w][kc][ns][kq][l15][u]
    whh_perm = wf.transpose(0, 2, 4, 1, 3, 5).reshape(1024, 256)
    perm = W._lstm_perm()
    np.testing.assert_array_equal(whh_perm, vap_or(cpc, "gAR.baseNet.weight_hh_l0")[perm])
    np.testing.assert_array_equal(get("lstm.b"), (cpc["gAR.baseNet.bias_ih_l0"] + cpc["gAR.baseNet.bias_hh_l0"])[perm])
    # heads: transposed copies and nod aux rows
    np.testing.assert_array_equal(get("head.wT").reshape(256, 256).T, vap["vap_head.weight"])
    aux = get("aux.w").reshape(8, 256)
    np.testing.assert_array_equal(aux[0:4], vap["nod_head.weight"])
    np.testing.assert_array_equal(aux[4], vap["bc_head.weight"][0])


def vap_or(sd, k):
    return sd[k]


def test_alibi_slopes_guard():
    cpc, vap = W.synthetic_weights(0, 20)
    assert np.allclose(W.alibi_slopes(), [0.25, 0.0625, 0.015625, 0.00390625])
    vap["ar.layers.0.mha.m"] = vap["ar.layers.0.mha.m"] * 2
    with pytest.raises(ValueError):
        W.pack_blob(cpc, vap)


def test_layout_header_matches_python():
    hdr = open(os.path.join(ROOT, "vap-realtime_amd", "csrc", "vapx_layout.h")).read()
    for K in (2, 5, 10, 20):
        lay = W.blob_layout(K)
        block = hdr[hdr.index(f"kLayoutK{K}[]"):]
        block = block[:block.index("};")]
        entries = re.findall(r'\{"([^"]+)", (\d+)u, (\d+)u\}', block)
        assert [(n, int(o), int(c)) for n, o, c in entries] == [(n, o, c) for n, (o, c) in lay.items()]


def test_cabi_library_exports_every_declared_symbol():
    lib = engine.load_library()
    hdr = open(os.path.join(ROOT, "include", "vapx.h")).read()
    declared = set(re.findall(r"\b(vapx_[a-z_0-9]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(lib, name), f"libvapx.so does not export {name}"
    assert set(engine.EXPORTS) == declared
    assert lib.vapx_abi_version() == 2
    for hz, K in ((50, 2), (20, 5), (10, 10), (5, 20)):
        assert lib.vapx_blob_floats(hz) == W.blob_layout(K)["__total__"][0]
    assert lib.vapx_blob_floats(7) == 0


def test_product_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    cpc, vap = W.synthetic_weights(0, 20)
    with pytest.raises(engine.VapxError, match="no HIP device|device"):
        engine.Engine(W.pack_blob(cpc, vap), 20, 2.5)
    from vap_realtime_amd.realtime import VAPRealTime
    with pytest.raises(engine.VapxError):
        VAPRealTime(vap, {"weights": cpc}, torch.device("cpu"), 20, 2.5)


def test_bad_config_rejected_before_touching_a_device():
    lib = engine.load_library()
    cfg = engine._Config(C.sizeof(engine._Config), 0, 7, 50, 1, 1, 0, 0)        # unsupported rate
    h = C.c_void_p()
    blob = np.zeros(4, np.float32)
    rc = lib.vapx_create(C.byref(cfg), blob.ctypes.data_as(C.c_void_p), blob.size, C.byref(h))
    assert rc == -1 and b"frame_hz" in lib.vapx_last_error(None)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "vap-realtime_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in re.sub(r'""".*?"""', "", src, flags=re.S), f"{fn} references the oracle"


def test_synth_is_deterministic_and_bounded():
    a = synth.dialogue(7, 16000)
    b = synth.dialogue(7, 16000)
    np.testing.assert_array_equal(a, b)
    assert a.dtype == np.float32 and a.shape == (2, 16000) and np.abs(a).max() <= 1.0
    assert np.abs(synth.dialogue(8, 16000) - a).max() > 1e-3


def test_sharding_partitions_exactly():
    for n, w in ((32768, 8), (1000, 3), (5, 8), (256, 1)):
        shards = [sharding.shard_streams(n, w, r) for r in range(w)]
        flat = [i for s in shards for i in s]
        assert flat == list(range(n))
        for r, s in enumerate(shards):
            for i in s[:3] + s[-3:]:
                assert sharding.owner_of(i, n, w) == r
                assert s[sharding.local_slot(i, n, w)] == i


def test_two_rank_gloo_sharding_and_timing_reduction(tmp_path):
    """The N>1 bench path (one process per GPU, barrier, max over ranks) on CPU with gloo, world 2."""
    script = tmp_path / "w.py"
    script.write_text(
        "import sys; sys.path.insert(0, %r)\n"
        "from vap_realtime_amd import dist_util, sharding\n"
        "rank, _, world = dist_util.env_rank()\n"
        "dist = dist_util.init('gloo')\n"
        "mine = sharding.shard_streams(512, world, rank)\n"
        "dist_util.barrier(dist)\n"
        "allids = dist_util.gather_ints(dist, mine)\n"
        "assert sorted(i for l in allids for i in l) == list(range(512))\n"
        "t = dist_util.max_over_ranks(dist, 1.0 + rank)\n"
        "assert t == float(world)\n"
        "dist_util.barrier(dist); dist.destroy_process_group()\n"
        "print('rank', rank, 'ok')\n" % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=120)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"rank {r} ok" in o, o


def test_bench_gpus_flag_really_spawns_one_rank_per_gpu():
    """`python bench.py --gpus 2` as the driver invokes it (no WORLD_SIZE in the environment) must create TWO ranks that
    rendezvous, shard the streams disjointly and reduce over both ranks.  --rendezvous-only stops before any GPU work
    (gloo), everything before that point is the real launch path."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rendezvous-only"], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    line = [l for l in out.stdout.decode().splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    assert r["n_gpus"] == 2 and len(set(r["pids"])) == 2
    assert r["shards"] == [[0, 4095, 4096], [4096, 8191, 4096]]      # 4096 streams per rank, disjoint, contiguous
    assert r["max_over_ranks"] == 0.002                               # the reduction saw rank 1's value


def test_bench_gpus_4_launch_path_ranks_meet_and_one_front_door_serves_four_shards():
    """The SCALE path kept warm without a GPU: `python bench.py --gpus 4` as the driver invokes it -> four ranks rendezvous over
    torch.distributed (gloo here; "nccl" = RCCL on the GPU node, where the same all-reduce fills `ranks_seen` of the record), shard
    4 x 4096 streams contiguously, and rank 0 serves 8 dialogues through ONE front door over four passive per-GPU front-ends
    (stand-in step functions): dialogue k is answered by shard k mod 4, slot k div 4 (vapx_frontdoor_*, vap_main.py:338-366)."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--rendezvous-only"], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    r = json.loads([l for l in out.stdout.decode().splitlines() if l.startswith("{")][-1])
    assert r["n_gpus"] == 4 and len(set(r["pids"])) == 4
    assert r["shards"] == [[4096 * k, 4096 * (k + 1) - 1, 4096] for k in range(4)]
    assert r["ranks_seen"] == {"backend": "gloo", "n": 4}
    fd = r["front_door"]
    assert fd["ok"] and fd["shards"] == 4 and fd["dialogues"] == 8 and fd["owners"] == [[k % 4, k // 4] for k in range(8)]
    assert fd["counts"]["accepted_in"] == 8 and fd["counts"]["accepted_out"] == 8 and fd["counts"]["refused"] == 0


def test_bench_flop_accounting_matches_the_survey_counts():
    """bench.py's per-kernel MAC table: the executed work it prices is below the reference's dense count by the exact savings
    (last-layer pruning, absorbed K/V, cached layer-0 Q|K|V), for both the fused (T <= 64) and the long-window kernel chains."""
    sys.path.insert(0, ROOT)
    import bench
    for (hz, T), dense in bench.GFLOP_PER_STREAM_FRAME.items():
        ex = 2.0 * sum(bench.macs_per_stream_frame(hz, T).values()) / 1e9
        assert 0.65 * dense < ex < 0.80 * dense, (hz, T, ex, dense)
    assert bench.attention_executed_fraction(250) == 36 / 64 and bench.attention_executed_fraction(50) == 0.75


def test_serving_precision_plan_follows_the_measured_rates():
    """``serve --precision auto`` = ``capacity.plan``.  Row (g) of the scope table at the benchmark's own loads: 4096 dialogues of
    the 20 Hz / 2.5 s model fit the fp32 path; C5 (bc + nod on one trunk) needs the split path; C3 (50 Hz / 5 s) fits neither at 4096 and
    the plan says what does fit."""
    from vap_realtime_amd import capacity as C
    p = C.plan(4096, 20, 2.5, "vap")
    assert p["precision"] == "fp32" and p["ok"] and 0.4 < p["fp32"]["busy"] < 0.55
    p = C.plan(4096, 20, 2.5, "bc+nod")
    assert p["precision"] == "split" and p["ok"] and p["fp32"]["busy"] > 0.85 and p["split"]["busy"] < 0.6 and "fp32 path would be" in p["reason"]
    assert 2800 <= p["fp32"]["max_streams"] <= 4000          # bench.py's paced search accepted 2856-3072 (p99 <= 9 ms AND <= 85 % busy)
    p = C.plan(4096, 50, 5.0, "vap")
    assert not p["ok"] and p["precision"] == "split" and "NEITHER" in p["reason"]
    assert 500 <= p["fp32"]["max_streams"] <= 800 and 1200 <= p["split"]["max_streams"] <= 1800      # measured: 520-576 / 1216-1296
    assert C.plan(500, 50, 5.0, "vap")["precision"] == "fp32" and C.plan(1200, 50, 5.0, "vap")["precision"] == "split"
    # the work model is the one bench.py prices its roofline with (one definition)
    import bench
    assert bench.model_macs is C.model_macs and bench.macs_per_stream_frame is C.macs_per_stream_frame
    assert C.executed_gflop_per_stream_frame(50, 250, "vap") == pytest.approx(2.763, rel=2e-3)      # 119.2 TF / 43 146 frames/s
    assert C.executed_gflop_per_stream_frame(20, 50, "vap") == pytest.approx(0.671, rel=2e-2)


def test_serve_auto_refuses_a_load_no_path_can_hold(capsys):
    """No GPU is touched: the plan is made before any engine exists.  4096 dialogues of the 50 Hz / 5 s model per GPU: start-up fails loudly."""
    from vap_realtime_amd import serve
    rc = serve.main(["--synthetic-weights", "0", "--streams", "4096", "--vap_process_rate", "50", "--context_len_sec", "5.0", "--port_num_in", "0",
                     "--port_num_out", "0"])
    err = capsys.readouterr().err
    assert rc == 1 and "NEITHER path holds" in err and "--allow-overload" in err
