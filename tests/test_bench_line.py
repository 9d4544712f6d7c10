"""bench.py's ONE stdout line must stay something the driver can read (round 4's line grew to 20.8 KB and the driver
recorded `parsed: null`).  The line is built from canned full records -- the committed record of round 4's own run, and a worst case with
long names everywhere and an N = 8 `multi_gpu` part -- and must be valid JSON below 4 KB that carries the headline, `roofline` and
`cpu_baseline`."""
import copy
import json
import os

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def canned():
    with open(os.path.join(ROOT, "profiles", "r04_bench.json")) as f:
        return json.load(f)


def check(line):
    assert "\n" not in line
    assert len(line) < 4096, len(line)
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "dtype", "data", "config"):
        assert k in d, k
    assert "workload" in d["config"] and "model" not in d["config"]
    return d


def test_line_from_the_round_4_record_is_short_and_carries_roofline_and_cpu_baseline():
    full = canned()
    assert len(json.dumps(full)) > 15000          # the record that broke the driver's parser
    d = check(bench.compact_line(full))
    r = d["roofline"]
    assert r["frac"] == full["roofline"]["frac"] and r["bound"] == "mfma" and r["unit"] == "TFLOP/s"
    assert abs(r["achieved"] / r["peak"] - r["frac"]) < 1e-3
    assert r["kernel_ms"] + r["init_kernel_ms"] > 0 and len(r.get("note", "")) <= 120
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and len(c["sample"]) <= 200
    b = d["baseline_config_rooflines"]
    assert len(b) == len(full["baseline_config_rooflines"])
    for v in b.values():
        assert set(v) <= {"frac", "ms_per_step", "kernel_ms", "bound"} and "frac" in v and "ms_per_step" in v
    assert "other_configs" not in d and "step_stats" not in d and "timing_protocol" not in d


def test_line_stays_short_with_eight_ranks_and_long_strings():
    full = canned()
    full["n_gpus"] = 8
    full["scaling"] = "strong"
    full["config"]["parallelism"] = "row-split x8 " + "x" * 500
    full["roofline"]["note"] = "n" * 1000
    full["roofline"]["kernel"] = "k" * 300 + " (" + "d" * 300 + ")"
    full["cpu_baseline"]["sample"] = "s" * 2000
    side = {"workload": "w" * 200, "ms_per_step": 0.12345, "gflops": 1.0, "compute_only_ms": 0.1, "exchange_only_ms": 0.05,
            "overlap_gain_ms": 0.01, "exchange": "push_fused", "chunks": 4, "ranks": 8, "shard_bytes": 1, "bytes_in_per_gpu": 7,
            "exchange_GBs_in_per_gpu": 100.0}
    full["other_configs"] = {name: copy.deepcopy(side) for name in ("weak_4096_rows_per_gpu", "config5_vocab512", "headline_rccl_allgather",
                                                                    "headline_push")}
    full["multi_gpu"] = {"compute_only_ms": 0.06, "exchange_only_ms": 0.055, "overlap_gain_ms": 0.0, "exchange": "push_fused",
                         "exchange_choice": "c" * 300, "chunks": 1, "ranks_observed": 8, "backend": "nccl", "shard_bytes": 8388608,
                         "bytes_in_per_gpu": 58720256, "exchange_GBs_in_per_gpu": 1000.0, "devices_shared_by_ranks": False}
    d = check(bench.compact_line(full))
    assert d["multi_gpu"]["ranks_observed"] == 8 and d["multi_gpu"]["other"]["config5_vocab512"]["ms_per_step"] == 0.12345
    assert d["roofline"]["frac"] and d["cpu_baseline"]["value"]


def test_optional_parts_are_dropped_rather_than_the_limit_broken():
    full = canned()
    full["baseline_config_rooflines"] = {f"config_{i}_" + "n" * 60: {"frac": 0.1, "ms_per_step": 0.1, "kernel_ms": 0.1, "bound": "mfma"}
                                         for i in range(80)}
    d = check(bench.compact_line(full))
    assert "dropped" in d["baseline_config_rooflines"] and d["roofline"]["frac"] and d["cpu_baseline"]["value"]
