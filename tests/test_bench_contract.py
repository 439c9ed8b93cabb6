"""The contract line of bench.py stays small enough for the driver's 8 000-byte stdout window.

Built from a recorded full bench object (profiles/r05_bench.json, 26 KB: the line the driver could not parse) and from an
inflated N > 1 shaped object; no GPU."""
import copy
import io
import json
import os

import pytest

from benchlib.contract import LIMIT, compact_line, secondary_tuples, stderr_table

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEAD = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
        "config", "roofline", "cpu_baseline", "parity")


def recorded():
    with open(os.path.join(ROOT, "profiles", "r05_bench.json")) as f:
        return json.load(f)


def driver_view(stdout_text):
    """what the driver does: keep the last 8 000 bytes of stdout, parse the last line"""
    tail = stdout_text.encode()[-8000:].decode(errors="replace")
    return json.loads(tail.strip().splitlines()[-1])


def test_recorded_line_fits_and_keeps_the_head():
    full = recorded()
    assert len(json.dumps(full)) > 20000          # the object that overflowed the window in round 5
    line, shed = compact_line(copy.deepcopy(full))
    assert len(line) + 1 <= LIMIT <= 8000 - 1500
    got = driver_view("RCCL version banner\n" * 400 + line + "\n")
    for k in HEAD:
        assert k in got, k
    assert got["value"] == full["value"] and got["ms_per_step"] == full["ms_per_step"]
    assert got["steps"] == full["steps"] and got["warmup"] == full["warmup"] and got["n_gpus"] == 1
    rf = got["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and rf["unit"] == "GB/s"
    assert abs(rf["frac"] - full["roofline"]["frac"]) < 1e-5 and abs(rf["achieved"] / rf["peak"] - rf["frac"]) < 1e-5
    assert abs(rf["traffic"] - full["roofline"]["traffic"]) / rf["traffic"] < 1e-5
    cb = got["cpu_baseline"]
    assert cb["kind"] == "reference" and cb["cores"] == 1 and cb["value"] > 1e6 and cb["sample"]
    assert got["parity"]["bit_exact"] is True
    # one compact tuple per secondary; every secondary of the full object is there
    sec = got["secondary"]
    for name in full["secondary"]:
        if name == "orders":
            assert "orders.zipf_c2.exact" in sec and "orders.generator_c5.window" in sec
        else:
            assert name in sec, name
            assert abs(sec[name]["value"] - full["secondary"][name]["value"]) / full["secondary"][name]["value"] < 1e-5
    assert sec["pairwise_k128"]["parity"] is True and abs(sec["pairwise_k128"]["frac"] - 0.4551) < 1e-3
    assert not shed


def test_oversized_objects_shed_secondaries_never_the_head():
    full = recorded()
    # an N > 1 shaped line: exchange, allreduce_step, transports, and far too many secondaries
    full["n_gpus"] = 8
    full["exchange"] = {"step": "stratified", "backend": "nccl", "windows": 64, "preflight_us": {"x" * 40 + str(i): 1.0 for i in range(200)},
                        "ladder_rung": 0, "ladder": "stratified ring over RCCL point-to-point" * 5}
    full["allreduce_step"] = {"value": 1.0e10, "unit": "instances/s", "ms_per_step": 10.0, "rmse_minus_sequential": 3e-5, "backend": "nccl",
                              "measured_as": "secondary.allreduce_minibatch of this run " * 10, "per_rank_ms": {str(r): [1.0] * 100 for r in range(8)}}
    full["per_rank_ms"] = {str(r): {"enqueue_" + str(i): 0.1 for i in range(50)} for r in range(8)}
    for i in range(300):
        full["secondary"]["extra_%03d" % i] = copy.deepcopy(full["secondary"]["pairwise_k128"])
    full["secondary"]["single_process_handle"] = {"p2p": copy.deepcopy(full["secondary"]["pairwise_k128"]), "rccl": {"error": "x" * 900}}
    line, shed = compact_line(full)
    assert len(line) + 1 <= LIMIT
    got = driver_view(line + "\n")
    for k in HEAD:
        assert k in got, k
    assert got["n_gpus"] == 8 and got["roofline"]["frac"] > 0 and got["cpu_baseline"]["value"] > 0
    assert shed and got["dropped_to_fit"]
    assert got["allreduce_step"]["value"] == 1.0e10          # the optional head members outlive the secondaries


def test_error_only_run_still_prints_a_head():
    line, _ = compact_line({"metric": "m", "value": 0.0, "unit": "instances/s", "n_gpus": 2, "steps": 0, "warmup": 0, "ms_per_step": None,
                            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                            "config": {"workload": "w"}, "roofline": None, "cpu_baseline": None, "parity": None, "secondary_error": "boom " * 500})
    got = json.loads(line)
    assert got["value"] == 0.0 and "secondary_error" in got and len(line) < LIMIT


def test_tuples_and_table():
    full = recorded()
    t = secondary_tuples(full["secondary"])
    assert set(t["svdpp_k128"]) >= {"value", "frac", "cpu", "parity"}
    assert t["orders.generator_c5.exact"]["cpu"] > 1e6
    txt = stderr_table(full)
    assert "pairwise_k128" in txt and len(txt.splitlines()) == len(t)


def test_bench_emit_uses_the_compact_line():
    """bench.py's one print of the line goes through compact_line (no other json.dumps(out) on stdout)"""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "compact_line(out)" in src and "print(json.dumps(out)" not in src
    multi = open(os.path.join(ROOT, "benchlib", "multi.py")).read()
    assert "compact_line" in multi        # the ladder's last-resort line too
