"""bench.py's LAST stdout line is what the driver parses: it must be one strict-JSON object of < 4 KB carrying the contract keys.
Round 3 printed a 30.8 KB line, the driver's 8 KB tail started in the middle of it and `BENCH_r03.parsed` was null.  The canned
record is that very line (profiles/r03z_bench_driver_cmd.json, the driver's command on the round-3 tree)."""
import copy
import json
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
            "config", "roofline", "cpu_baseline")


def canned():
    with open(os.path.join(ROOT, "profiles", "r03z_bench_driver_cmd.json")) as f:
        return json.load(f)


def strict(text):
    def no_const(c):
        raise ValueError(f"non-strict JSON constant {c}")
    return json.loads(text, parse_constant=no_const)


def test_compact_line_is_small_strict_json_with_the_contract_keys():
    full = canned()
    assert len(json.dumps(full)) > 20000                       # the record that could not be parsed
    text = bench.compact_line(full, "gpurun_out/bench_full.json")
    assert "\n" not in text and len(text.encode()) < 4096, len(text)
    line = strict(text)
    for k in CONTRACT:
        assert k in line, k
    assert line["value"] == pytest.approx(full["value"], rel=1e-6)
    assert line["ms_per_step"] == pytest.approx(full["ms_per_step"], rel=1e-5)
    assert line["config"]["workload"].startswith("c3:") and "model" not in line["config"]
    roof = line["roofline"]
    assert roof["bound"] == "mfma" and roof["unit"] == "TFLOP/s" and roof["frac"] == pytest.approx(roof["achieved"] / roof["peak"], rel=2e-3)
    assert roof["traffic"] is None or roof["traffic"] > 0
    cb = line["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and cb["sample"]
    assert line["parity_gate"]["ok"] is True and line["parity_gate"]["worst_abs"] <= 1e-4
    assert set(line["configs"]) == {"c2", "s4096_20hz", "c5"}
    for rec in line["configs"].values():
        assert rec["value"] > 0 and 0 < rec["frac"] <= 1.0 and rec["streams_at_10ms"] > 0 and rec["split"]["value"] > rec["value"]
    assert line["split_f16"]["value"] > line["value"]
    assert line["front_end"]["streams"] == 4096


def test_compact_line_survives_non_finite_numbers_missing_legs_and_many_ranks():
    full = canned()
    full["n_gpus"] = 8
    full["per_rank"] = {"value_min": 41000.123456, "value_max": 44000.98765}
    full["roofline"]["traffic"] = None
    full["configs"]["c5"]["split_f16"] = {"error": "bench.py: PARITY GATE FAILED for c5_split_f16: " + "x" * 5000}
    full["configs"]["c2"]["paced_latency"]["p99_ms"] = float("nan")
    full["front_end"] = {"error": "TimeoutExpired: " + "y" * 3000}
    del full["cpu_baseline_multiprocess"]
    text = bench.compact_line(bench._jsonable(full), "bench_full.json")
    assert len(text.encode()) < 4096
    line = strict(text)
    assert line["n_gpus"] == 8 and line["per_rank"]["value_min"] < line["per_rank"]["value_max"]
    assert line["configs"]["c2"]["p99_ms"] is None
    assert "error" in line["configs"]["c5"]["split"] and "error" in line["front_end"]
    # a minimal record (--no-latency --no-cpu-baseline --configs ''): the contract keys that exist still come out
    mini = {k: copy.deepcopy(v) for k, v in canned().items() if k not in ("configs", "split_f16", "paced_latency", "front_end", "cpu_baseline",
                                                                          "cpu_baseline_multiprocess", "concurrent_streams_at_10ms")}
    line = strict(bench.compact_line(mini))
    assert line["value"] > 0 and "configs" not in line and "cpu_baseline" not in line


def test_no_kernel_class_is_booked_twice():
    """Round 3 booked EPI_BIAS_LN_GELU (epilogue id 5) under class 5 = conv_tail, and the C5 line printed 294 TF for `conv_tail` on
    the fp32 path (peak 157.3).  The class table, the header and the engine's enum must agree and must be collision-free."""
    from vap_realtime_amd import engine
    names = list(engine.PROF_CLASSES.values())
    assert len(set(names)) == len(names) and engine.PROF_CLASSES[5] == "conv_tail" and engine.PROF_CLASSES[13] == "gemm_bias_ln_gelu"
    src = open(os.path.join(ROOT, "vap-realtime_amd", "csrc", "engine.hip")).read()
    m = re.search(r"CLS_COUNT = (\d+)", src)
    assert m and int(m.group(1)) == len(names)
    hdr = open(os.path.join(ROOT, "include", "vapx.h")).read()
    assert f"#define VAPX_PROF_CLASSES {len(names)}" in hdr
    assert "ProfScope ps(h, gemm_class(epi), st);" in src and "ProfScope ps(h, epi, st);" not in src
    # every class the MAC model prices exists as a profile class (or is the derived conv_tail split)
    for c in bench.model_macs(20, 50, "nod", leader=False):
        assert c in names, c


def test_roofline_bookkeeping_describes_one_set_of_launches():
    """Round 4's line divided the HBM bytes of ONE mode-1 launch by the mean time of three mode-1 and two mode-2 launches.  Now the mode-2
    launches are a profile class of their own, `traffic` is the class mean over exactly the launches the HIP events cover, the tick total
    and its ratio to the algorithmic bytes ride in the line, and `executed_tflops` counts the attention tiles that run."""
    from vap_realtime_amd import engine
    assert engine.PROF_CLASSES[14] == "ffn_proj"
    D = 256
    for hz, T, mode, n_ffn, n_proj in ((50, 250, "vap", 3, 7), (20, 100, "nod", 4, 10)):
        m = bench.model_macs(hz, T, mode)
        assert m["ffn_proj"] == 2 * T * D * (n_proj - n_ffn) * D
        assert m["ffn_block"] == 2 * T * D * (n_ffn * 2 * 768 + (n_ffn - 1) * (768 + 512) + n_ffn * D)
    assert bench.model_macs(20, 50, "vap")["ffn_proj"] == 0      # short windows: the projections ride in the fused attention block
    # the C3 tick of the committed PMC passes: class means and the tick total
    cls, tick, _src = bench.load_traffic("4096x50hz_T250", "ffn_block")
    cls2, _, _ = bench.load_traffic("4096x50hz_T250", "ffn_proj")
    assert 15e9 < cls < 25e9 and 6e9 < cls2 < 10e9 and 100e9 < tick < 160e9
    ratio = tick / (bench.ALGO_BYTES_PER_STREAM_FRAME[(50, 250)] * 4096)
    assert 40 < ratio < 80
    full = canned()
    full["roofline"]["traffic_ratio"] = ratio
    full["roofline"]["tick_traffic"] = tick
    line = strict(bench.compact_line(full))
    assert line["roofline"]["traffic_ratio"] == pytest.approx(ratio, rel=1e-2) and len(json.dumps(line)) < 4096
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert '"executed_tflops": value * exec_gflop_causal / 1e3' in src
    # round 5: the board's watts and shader clock during the timed steps, and the roofline fraction at that clock, ride in the line
    full["board"] = {"watts_median": 1325.0, "watts_max": 1337.0, "sclk_mhz_median": 2321.0, "sclk_mhz_min": 2313.0, "samples": 31}
    full["roofline"]["frac_at_sclk"] = full["roofline"]["frac"] * bench.NOMINAL_SCLK_MHZ / 2321.0
    if isinstance(full.get("split_f16"), dict) and "value" in full["split_f16"]:
        full["split_f16"]["board"] = {"watts_median": 1372.0, "sclk_mhz_median": 1957.0}
    line = strict(bench.compact_line(full))
    assert line["board"] == {"watts": 1325.0, "sclk_mhz": 2321.0}
    assert line["roofline"]["frac_at_sclk"] == pytest.approx(full["roofline"]["frac"] * 2400.0 / 2321.0, rel=1e-3)
    if "split_f16" in line and "value" in line["split_f16"]:
        assert line["split_f16"]["watts"] == 1372.0 and line["split_f16"]["sclk_mhz"] == 1957.0
    assert len(json.dumps(line)) < 4096


def test_the_timed_number_carries_its_spread_and_the_traffic_its_provenance():
    """(a) `roofline.traffic*` are read from committed PMC passes — the line must say which tree those were taken at and
    whether the kernels timed now are that tree's; (b) the one driver-timed number rests on K ticks — their min / median / p95 ride along."""
    from vap_realtime_amd import provenance
    now = provenance.kernel_source_hash()
    assert len(now) == 16 and now == provenance.kernel_source_hash()
    committed = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    for key, entry in committed.items():                         # every workload's passes name their tree
        if not key.startswith("_"):
            assert entry.get("_source", {}).get("csrc_sha") and entry["_source"].get("git"), f"{key}: PMC passes without provenance stamp"
    _, _, src = bench.load_traffic("4096x50hz_T250", "ffn_block")
    assert src["file"] == "profiles/pmc_traffic.json" and src["csrc_sha_now"] == now and src["git"]
    assert src["stale"] == (src["csrc_sha"] != now) and isinstance(src["stale"], bool)
    _, _, none = bench.load_traffic("no such workload", "ffn_block")
    assert none["stale"] is None and none["csrc_sha"] is None    # no passes at all is not "fresh"
    full = canned()
    full["roofline"]["traffic_source"] = src
    full["ms_per_step_spread"] = {"min": 93.1234567, "median": 94.5, "p95": 96.25, "max": 97.0, "n": 20}
    line = strict(bench.compact_line(full))
    assert line["roofline"]["traffic_source"] == {"file": src["file"], "git": src["git"], "csrc_sha": src["csrc_sha"], "stale": src["stale"]}
    assert (line["ms_per_step_min"], line["ms_per_step_median"], line["ms_per_step_p95"]) == (93.123, 94.5, 96.25)
    assert len(json.dumps(line)) < 4096
    text = open(os.path.join(ROOT, "bench.py")).read()
    assert "tick_ev[i + 1].record(tstream)" in text and '"traffic_source": traffic_src' in text     # measured inside the timed region / stamped per record


def test_bench_main_prints_only_the_compact_line():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert src.count("print(compact_line(") == 1
    assert "print(json.dumps(result))" not in src


def test_board_watch_reads_hwmon_power_and_clock(tmp_path):
    """bench.BoardWatch against a stand-in sysfs tree: microwatts / hertz in, watts / MHz medians out; no hwmon files -> no record."""
    import time
    hw = tmp_path / "card3" / "device" / "hwmon" / "hwmon7"
    hw.mkdir(parents=True)
    (hw / "power1_input").write_text("1325000000\n")
    (hw / "freq1_input").write_text("2321000000\n")
    w = bench.BoardWatch(period=0.005, root=str(tmp_path))
    with w:
        time.sleep(0.15)
    rec = w.record()
    assert rec and rec["watts_median"] == 1325.0 and rec["sclk_mhz_median"] == 2321.0 and rec["samples"] >= 3
    empty = bench.BoardWatch(root=str(tmp_path / "nothing"))
    with empty:
        pass
    assert empty.record() is None
