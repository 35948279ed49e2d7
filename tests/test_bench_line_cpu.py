"""The result line of bench.py: ONE compact JSON object the driver can parse (round 5's line had grown to 28 KB and BENCH_r05.parsed came back null).

CPU test: bench.compact() applied to a recorded FULL object of a real MI355X run (tests/golden/bench_full_r06.json, written by bench.py itself as
bench_detail.json) must give a strict-JSON line under 8 KB carrying every field the contract and the round-5 review name."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _strict(line):
    def no_const(name):
        raise ValueError("non-finite number %s in the result line" % name)
    return json.loads(line, parse_constant=no_const)


def test_compact_line_of_a_recorded_run_is_small_strict_and_complete():
    import bench
    full = json.load(open(os.path.join(ROOT, "tests", "golden", "bench_full_r06.json")))
    line = json.dumps(bench.compact(full, ["bench_detail.json"]), separators=(",", ":"))
    assert "\n" not in line and len(line) < 8192, len(line)
    r = _strict(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in r, k
    assert r["config"]["workload"] and "model" not in r["config"]
    assert abs(r["value"] * r["ms_per_step"] * 1e-3 * r["steps"] - r["config"]["frames_per_gpu"] * r["n_gpus"]) / (r["config"]["frames_per_gpu"] * r["n_gpus"]) < 1e-3
    R = r["roofline"]
    assert R["bound"] == "hbm" and R["unit"] == "GB/s" and R["peak"] == 8000.0
    assert 0 < R["frac"] <= 1 and abs(R["frac"] - R["achieved"] / R["peak"]) < 1e-3
    assert R["traffic"] and abs(R["traffic"] / (R["avg_launch_ms"] * 1e-3) / 1e9 - R["achieved"]) / R["achieved"] < 1e-2      # frac IS traffic / time / peak
    assert R["traffic"] >= R["model_bytes_per_launch"] * 0.99                                                                    # counter bytes cannot be below the kernel's own lower bound
    assert 0 < R["batch1_frac"] <= 1 and R["algorithmic_bytes_per_launch"] > 0
    cb = r["cpu_baseline"]
    assert cb["value"] > 0 and cb["cores"] >= 1 and cb["kind"] in ("port", "reference") and cb["sample"]
    assert r["parity"]["keys_equal"] is True and r["parity"]["voxels_bit_equal"] is True
    assert r["icp"]["iters_per_s"] > 0 and r["icp"]["in_tolerance_iters_per_s"] > 0 and r["icp"]["mode"]
    assert r["multi_gpu"]["ranks"] == r["n_gpus"]


def test_compact_line_drops_prose_and_rounds_numbers():
    import bench
    full = json.load(open(os.path.join(ROOT, "tests", "golden", "bench_full_r06.json")))
    c = bench.compact(full)
    flat = json.dumps(c)
    assert "note" not in c["roofline"] and "evidence" not in flat
    assert all(len(v) < 400 for v in _strings(c)), [v for v in _strings(c) if len(v) >= 400]


def _strings(o):
    if isinstance(o, str):
        yield o
    elif isinstance(o, dict):
        for v in o.values():
            yield from _strings(v)
    elif isinstance(o, list):
        for v in o:
            yield from _strings(v)


def test_compact_line_of_a_two_rank_run_carries_the_merge():
    """The N > 1 shape (recorded from `bench.py --gpus 2` with the library merge through the multi-process RCCL double): multi_gpu says which code merged, on how many
    ranks, with which algorithm and what crossed the wire; the ICP / tracking sections (one-GPU measurements) are absent; value = all ranks' frames / max-over-ranks time."""
    import bench
    full = json.load(open(os.path.join(ROOT, "tests", "golden", "bench_full_r06_n2.json")))
    line = json.dumps(bench.compact(full, ["bench_detail.json"]), separators=(",", ":"))
    assert len(line) < 8192
    r = _strict(line)
    mg = r["multi_gpu"]
    assert r["n_gpus"] == 2 and mg["ranks"] == 2 and mg["merge_impl"] == "cabi" and mg["rccl_ranks"] == 2 and mg["merge_algorithm"] == "owner" and mg["merge_fallback"] is None
    assert mg["union_blocks"] == r["per_frame"]["final_blocks_rank0"] and len(mg["per_rank"]) == 2 and len(mg["wire_bytes_sent_per_rank"]) == 2
    assert all(p["merge_ms"] > 0 and p["fusion_ms"] > 0 for p in mg["per_rank"])
    assert "icp" not in r and "tracking" not in r and "cpu_baseline" not in r
    assert abs(r["value"] * r["ms_per_step"] * 1e-3 * r["steps"] - 2 * r["config"]["frames_per_gpu"]) / (2 * r["config"]["frames_per_gpu"]) < 1e-3
    assert 0 < r["roofline"]["frac"] <= 1
