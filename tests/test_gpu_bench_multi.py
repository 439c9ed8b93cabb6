"""bench.py --gpus N end to end on the ONE-GPU test box (SVDF_BENCH_SHARE_GPU=1: every rank on GPU 0, exchange through gloo): the whole
N > 1 flow of one driver command -- rendezvous + checked preflight under the watchdog, the stratified ring as the main line, the all-reduce
window-minibatch step and the single-process amd:gpus handle as secondaries on the same data -- ends in ONE JSON line with the fields the
scaling record is read from, inside the accuracy contract |dRMSE| <= 1e-4.  Replaces svd_feature.cpp:220-248,
272-283 (the reference's round loop) on N ranks."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, extra_env=None, timeout=1500):
    """-> (the FULL object bench.py wrote to its side file, stderr); the stdout line itself is checked here: one line, <= 6 000 bytes, the
    contract head present, value / ms_per_step equal to the full object's (benchlib/contract.py)"""
    import tempfile
    side = os.path.join(tempfile.mkdtemp(prefix="svdf_bench_side_"), "bench_secondary.json")
    env = dict(os.environ, SVDF_BENCH_SHARE_GPU="1", SVDF_BENCH_SIDE_FILE=side)
    for k in ("SVDF_BENCH_ATTEMPT", "SVDF_BENCH_FALLBACK_LOG", "RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    env.update(extra_env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line expected:\n" + p.stdout[-2000:]
    assert p.stdout.rstrip("\n").splitlines()[-1] == lines[0], "the contract line is the LAST line of stdout"
    assert len(lines[0]) + 1 <= 6000, len(lines[0])
    short = json.loads(p.stdout.encode()[-8000:].decode().strip().splitlines()[-1])    # the driver's view: the last 8 000 bytes
    with open(side) as f:
        full = json.load(f)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "roofline", "cpu_baseline", "parity"):
        assert key in short, key
    assert short["value"] == full["value"] and short["ms_per_step"] == full["ms_per_step"] and short["n_gpus"] == full["n_gpus"]
    if "exchange" in full:
        assert short["exchange"]["step"] == full["exchange"]["step"] and short["exchange"]["ladder_rung"] == full["exchange"]["ladder_rung"]
        assert abs(short["rmse_minus_sequential"] - full["rmse_minus_sequential"]) < 1e-9
    if isinstance(full.get("allreduce_step"), dict) and "value" in full["allreduce_step"]:
        assert abs(short["allreduce_step"]["value"] / full["allreduce_step"]["value"] - 1) < 1e-5
    for name, r in (full.get("secondary") or {}).items():
        if isinstance(r, dict) and "value" in r:
            assert abs(short["secondary"][name]["value"] / r["value"] - 1) < 1e-5, name
    return full, p.stderr


def test_two_ranks_one_command_yields_every_multi_gpu_number():
    # a replica of configs[2] at its density (100 ratings per user, 1000 per item): at 5 M ratings over the full 1 M x 100 K id space
    # (5 per user) the RMSE moves by 2e-4 under ANY reordering of the file, which says nothing about the exchange
    line, err = _bench(["--gpus", "2", "--users", "50000", "--items", "5000", "--ratings", "5000000", "--steps", "2", "--multi-secondary", "all"])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "roofline", "cpu_baseline", "exchange", "phase_ms", "per_rank_ms", "roofline_aggregate", "model_ms",
                "rmse_test_after_run", "rmse_sequential_reference", "rmse_minus_sequential", "secondary"):
        assert key in line, key
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["value"] > 0 and line["unit"] == "instances/s"
    x = line["exchange"]
    assert x["step"] == "stratified" and x["ladder_rung"] == 0 and x["fallback"] is None and x["backend"] == "gloo"
    assert set(("allreduce_4KB_us", "allreduce_13MB_fp16_us", "handoff_4KB_us", "handoff_1.6MB_us")) <= set(x["preflight_us"])
    assert abs(line["rmse_minus_sequential"]) <= 1e-4
    pr = line["per_rank_ms"]
    assert 0 < pr["min"] <= pr["max"] and abs(pr["max"] - line["ms_per_step"]) < 1e-6
    assert pr["instances_min"] + pr["instances_max"] == 5000000
    ra = line["roofline_aggregate"]
    assert ra["peak"] == 16000.0 and 0 < ra["frac"] < 1
    assert line["model_ms"]["compute_share_ms"] > 0
    # north_star's step on the same data, same command
    ar = line["secondary"]["allreduce_minibatch"]
    assert ar["exchange"]["step"] == "minibatch" and ar["value"] > 0 and abs(ar["rmse_minus_sequential"]) <= 1e-4
    assert set(("compute", "pack", "allreduce", "unpack")) <= set(ar["phase_ms"])
    # north_star's step sits at the TOP level of every N > 1 line, with the ranks' own clocks and the world size the group reports
    ns = line["allreduce_step"]
    assert ns["value"] == ar["value"] and ns["n_gpus"] == 2 and ns["world_size_reported"] == 2 and ns["measured_as"].startswith("secondary.allreduce_minibatch")
    assert ns["per_rank_ms"]["max"] > 0 and abs(ns["rmse_minus_sequential"]) <= 1e-4 and x["world_size_reported"] == 2
    # the same two steps with the direct exchange between the processes (IPC-mapped buffers; on this box the ranks share the device)
    for key, step in (("allreduce_minibatch_ipc", "minibatch"), ("stratified_ipc", "stratified")):
        ip = line["secondary"][key]
        assert ip["exchange"]["step"] == step and ip["exchange"]["transport"].startswith("ipc") and ip["value"] > 0
        assert abs(ip["rmse_minus_sequential"]) <= 1e-4
    assert x["contributions"] == "bf16"
    # one C-ABI handle over both (here: virtual) ranks: the peer-pointer exchange runs; RCCL refuses ranks that share a device, and says so
    sp = line["secondary"]["single_process_handle"]
    assert sp["p2p"]["value"] > 0 and sp["p2p"]["exchange_path"] == "p2p" and sp["p2p"]["exchanges"] > 0
    assert abs(sp["p2p"]["rmse_test_after_run"] - ar["rmse_sequential_reference"]) < 5e-3   # 4 passes against the secondary's 5: same ball park
    assert "error" in sp["rccl"] or sp["rccl"]["value"] > 0


@pytest.mark.parametrize("hook", ["SVDF_BENCH_TEST_FAIL_PREFLIGHT", "SVDF_BENCH_TEST_HANG_PREFLIGHT"])
def test_a_broken_ring_preflight_falls_back_to_the_all_reduce_step_and_says_so(hook):
    line, err = _bench(["--gpus", "2", "--users", "10000", "--items", "1000", "--ratings", "1000000", "--steps", "1", "--no-multi-secondary",
                        "--no-cpu-baseline", "--preflight-timeout", "45"], {hook: "0"})
    x = line["exchange"]
    assert x["ladder_rung"] == 1 and x["step"] == "minibatch"
    assert line["allreduce_step"]["measured_as"] == "the main line of this run" and line["allreduce_step"]["value"] == line["value"]
    assert x["fallback"] and x["fallback"][0]["attempt"] == 0
    assert line["value"] > 0 and abs(line["rmse_minus_sequential"]) <= 1e-4


def test_the_preflight_timings_pick_the_main_schedule_and_the_other_one_becomes_a_secondary():
    """--exchange auto on real devices decides ring vs all-reduce step from the preflight (bench.choose_schedule); gloo's host-staged hand-over is so slow
    that the test hook makes the all-reduce step the main line here -- the ring is then measured as secondary.stratified_ring"""
    line, err = _bench(["--gpus", "2", "--users", "10000", "--items", "1000", "--ratings", "1000000", "--steps", "1", "--no-cpu-baseline"],
                       {"SVDF_BENCH_TEST_AUTO_SCHEDULE": "1"})
    x = line["exchange"]
    assert x["ladder_rung"] == 0 and x["fallback"] is None
    c = x["schedule_choice"]
    assert c["pick"] == x["step"] and c["est_stratified_ms"] > 0 and c["est_allreduce_step_ms"] > 0
    other = "stratified_ring" if c["pick"] == "minibatch" else "allreduce_minibatch"
    assert line["secondary"][other]["value"] > 0 and abs(line["secondary"][other]["rmse_minus_sequential"]) <= 1e-4
    assert abs(line["rmse_minus_sequential"]) <= 1e-4


def test_a_transport_that_cannot_be_opened_skips_its_secondary_only():
    """two ranks on ONE device: ncclCommInitRank refuses the duplicate GPU on every rank -- the ranks agree through the store, the native secondaries
    carry the error, the IPC secondaries and the line itself are unaffected"""
    line, err = _bench(["--gpus", "2", "--users", "10000", "--items", "1000", "--ratings", "1000000", "--steps", "1", "--no-cpu-baseline",
                        "--secondary-timeout", "150", "--multi-secondary", "all"], {"SVDF_BENCH_TEST_NATIVE_ON_GLOO": "1"})
    sec = line["secondary"]
    for key in ("stratified_native", "allreduce_minibatch_native"):
        assert "error" in sec[key] and "native" in sec[key]["error"], sec[key]
    for key in ("allreduce_minibatch_ipc", "stratified_ipc"):
        assert sec[key]["value"] > 0 and abs(sec[key]["rmse_minus_sequential"]) <= 1e-4
    assert line["value"] > 0 and "secondary_error" not in line
