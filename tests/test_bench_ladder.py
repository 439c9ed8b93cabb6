"""bench.py --gpus N must always end in ONE JSON line: rendezvous + preflight run under a watchdog, and a rung that
fails or hangs is left through os.execv to the next one (RCCL ring -> RCCL all-reduce only -> gloo).  Here: the ladder itself on the CPU
with two gloo processes under torch.distributed.run -- an injected failure and an injected hang on one rank both end one rung lower on
every rank, with the reasons in the line; plus the model figures the N > 1 line carries."""
import json
import os
import socket
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(mode, world=2):
    env = dict(os.environ)
    env.pop(bench.ATTEMPT_ENV, None)
    env.pop(bench.FALLBACK_ENV, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_port()), os.path.join(ROOT, "tests", "ladder_helper.py"), mode]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=240)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0]), p.stderr


def test_no_failure_stays_on_the_first_rung():
    line, _ = _run("ok")
    assert line["rung"] == 0 and line["fallback"] == [] and line["sum"] == 3.0


def test_a_failing_rank_takes_every_rank_one_rung_down():
    line, err = _run("fail")
    assert line["rung"] == 1 and line["sum"] == 3.0
    assert line["fallback"] and all(f["attempt"] == 0 for f in line["fallback"])
    assert "preflight" in line["fallback"][0]["reason"]


def test_a_hanging_rank_is_left_by_the_watchdog():
    line, err = _run("hang")
    assert line["rung"] == 1 and line["sum"] == 3.0
    assert "WATCHDOG" in err   # the hanging rank's stacks were dumped before it re-executed


def test_watchdog_fires_only_when_armed_and_late():
    fired = []
    wd = bench.Watchdog(lambda m: None)
    wd.arm(0.2, "fast phase", lambda reason: fired.append(reason))
    wd.disarm()
    time.sleep(0.4)
    assert fired == []
    wd.arm(0.1, "slow phase", lambda reason: fired.append(reason))
    time.sleep(0.5)
    assert len(fired) == 1 and "slow phase" in fired[0]


def test_model_figures_of_the_line():
    # ring all-reduce of 13 MB at 8 ranks ~ 0.18 ms per window, 32 windows; 6c share 2.98 ms
    m = bench.model_ms("basicmf", 8, "minibatch", 100_000_000, 100_000, 64, 32, 1, 0, 23.5)
    assert m["compute_share_ms"] == 2.66 and 0.17 < m["allreduce_ms_per_window"] < 0.19 and 8.2 < m["total_ms"] < 8.7
    # 6f: 64 hand-overs of 1.6 MB, hidden .. serial
    s = bench.model_ms("basicmf", 8, "stratified", 100_000_000, 100_000, 64, 64, 16, 64, 23.5)
    assert s["compute_share_ms"] == 2.71 and s["total_ms"][0] == 2.71 and 5.5 < s["total_ms"][1] < 6.5
    # another size: T1 / N
    o = bench.model_ms("basicmf", 2, "minibatch", 5_000_000, 100_000, 64, 2, 1, 0, 23.5 * 0.05)
    assert abs(o["compute_share_ms"] - 23.5 * 0.05 / 2) < 1e-9


def test_schedule_choice_follows_the_preflight_timings():
    import argparse
    a = argparse.Namespace(ratings=100_000_000, items=100_000, factor=0, blocks_per_rank=2, chunks=0, windows=0)
    fast_ring = bench.choose_schedule({"handoff_1.6MB_us": 50.0, "allreduce_13MB_fp16_us": 180.0}, 8, a)
    assert fast_ring["pick"] == "stratified" and fast_ring["handoffs_per_pass"] == 64 and 3.0 < fast_ring["est_stratified_ms"] < 3.4
    assert 7.5 < fast_ring["est_allreduce_step_ms"] < 8.7
    slow_ring = bench.choose_schedule({"handoff_1.6MB_us": 200.0, "allreduce_13MB_fp16_us": 180.0}, 8, a)
    assert slow_ring["pick"] == "minibatch" and slow_ring["est_stratified_ms"] > slow_ring["est_allreduce_step_ms"]
    # a hidden hand-over never makes the estimate smaller than the compute share
    assert bench.choose_schedule({"handoff_1.6MB_us": 1.0, "allreduce_13MB_fp16_us": 180.0}, 8, a)["est_stratified_ms"] == 2.71
