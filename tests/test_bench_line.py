"""The bench contract's LAST stdout line must stay parseable by the driver: round 3's 26 KB line was not (BENCH_r03.parsed = null).
`bench.compact_line` derives the line from the full record; here it is fed canned records (the round-3 full line committed under
profiles/, and a synthetic worst case with every optional block present and long strings) and must stay under bench.LINE_LIMIT."""
import copy
import json
import os

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config")


def _canned():
    return json.load(open(os.path.join(ROOT, "profiles", "r03_run22_bench_full.json")))


def test_round3_record_compacts_under_limit():
    full = _canned()
    assert len(json.dumps(full)) > 20000            # the record that broke the driver's parser
    line = bench.compact_line(full, "gpurun_out/bench_detail.json")
    s = json.dumps(line)
    assert len(s) < bench.LINE_LIMIT, len(s)
    back = json.loads(s)
    for k in CONTRACT:
        assert k in back, k
    assert back["config"]["workload"].startswith("full FGT forward")
    r = back["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "kernel_ms_per_step", "sustained"):
        assert k in r, k
    assert r["bound"] in ("mfma", "hbm") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    cb = back["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert {x["kind"] for x in back["rooflines"]} >= {"conv", "attn_temporal", "attn_spatial", "layernorm", "fold"}
    assert set(back["fp32_exact"]) >= {"value", "ms_per_step", "roofline_frac"}
    c4 = back["c4"]
    assert c4["pipeline_frames_per_s"] > 0 and "raft_864x480" in c4["stages"] and "ms_per_pair" in c4["stages"]["raft_864x480"]
    assert "dropped" not in back


def test_worst_case_record_still_under_limit():
    full = _canned()
    full["metric"] = full["metric"] + " " + "x" * 200
    full["config"]["sharding"] = "y" * 2000
    full["cpu_baseline"]["sample"] = "z" * 5000
    full["strong_error"] = "e" * 300
    full["weak_scaling_clip_per_rank"] = {"value": 1.0, "unit": "frames/s", "ms_per_step": 1.0, "scaling": "weak", "output_checksum": 1.0}
    full["strong_scaling_ideal"] = {"window_phase_speedup_bound": 7.1, "frame_phase_speedup_bound": 8.0, "note": "n" * 1000}
    full["phases_ms"] = {f"rank{i}": {"encode": 1.0, "gather_wait": 1.0, "windows": 1.0, "exchange": 1.0, "blend": 1.0} for i in range(8)}
    big = copy.deepcopy(full["rooflines"])
    full["rooflines"] = big + big                     # twice as many kernels as today
    line = bench.compact_line(full, "gpurun_out/bench_detail.json")
    s = json.dumps(line)
    assert len(s) < bench.LINE_LIMIT, len(s)
    back = json.loads(s)
    for k in CONTRACT + ("roofline", "cpu_baseline"):
        assert k in back, k


def test_c4_error_and_missing_blocks():
    full = _canned()
    full["c4"] = {"error": "RuntimeError: " + "q" * 1000, "trace": "t" * 1200}
    for k in ("fp32_exact", "f16", "parity_vs_cpu_oracle", "cpu_baseline", "rooflines", "roofline"):
        full.pop(k, None)
    line = bench.compact_line(full)
    assert len(json.dumps(line)) < bench.LINE_LIMIT
    assert "error" in line["c4"] and "detail" not in line


def test_partial_record_still_yields_the_contract_keys():
    """a record an optional block cannot digest (a rooflines entry without its time, a c4 stage of an unexpected shape) must not
    cost the final line — the contract keys, `roofline` and `cpu_baseline` still go out, with the reason."""
    full = _canned()
    del full["rooflines"][0]["kernel_ms_per_step"]
    line = bench.compact_line(full, "gpurun_out/bench_detail.json")
    s = json.dumps(line)
    assert len(s) < bench.LINE_LIMIT
    for k in CONTRACT + ("roofline", "cpu_baseline", "compact_error"):
        assert k in line, k
    full = _canned()
    full["c4"]["stages"]["raft_864x480"] = 3.0              # not a dict
    line = bench.compact_line(full)
    assert "compact_error" in line and all(k in line for k in CONTRACT)


def test_oversized_non_droppable_blocks_fall_back_to_the_contract_keys():
    full = _canned()
    full["config"] = {"workload": "w" * 150, **{f"k{i}": "v" * 190 for i in range(40)}}      # 40 x 200 bytes of config survive every pop
    full["pipeline_sharded"] = {f"s{i}": "p" * 100 for i in range(30)}
    line = bench.compact_line(full, "gpurun_out/bench_detail.json")
    assert len(json.dumps(line)) < bench.LINE_LIMIT
    for k in CONTRACT:
        assert k in line, k
    assert "dropped" in line


def test_graphs_policy_selects_replay_for_sharded_runs_and_launch_bound_hosts():
    """at N > 1 a rank's share of the step (13-20 ms of kernels) is shorter than the time a slow host needs to enqueue its ~600
    launches: bench.py --gpus N must replay hipGraphs by default; at N = 1 only when the probe step says the host is the bound."""
    g = bench.graphs_enabled
    assert g("auto", 8) and g("auto", 2) and g("auto", 2, 0.001, 1.0)
    assert not g("auto", 1) and not g("auto", 1, 0.009, 0.100)            # the builder's boxes: 9 of 100 ms
    assert g("auto", 1, 0.067, 0.102)                                      # the driver's round-4 box: 67 of 102 ms
    assert g("on", 1) and not g("off", 8)


def test_compact_line_carries_the_exchange_modes_and_launch_probe():
    full = _canned()
    full["strong_scaling_modes"] = {"allgather": {"value": 3000.0, "ms_per_step": 26.6, "host_enqueue_ms_per_step": 4.0, "output_checksum": 1.0},
                                    "a2a": {"error": "a2a: RuntimeError: " + "x" * 250}}
    full["host_launch_us_probe"] = 14.2
    line = bench.compact_line(full)
    assert line["strong_scaling_modes"]["allgather"]["value"] == 3000.0 and line["host_launch_us_probe"] == 14.2
    assert len(json.dumps(line)) < bench.LINE_LIMIT
