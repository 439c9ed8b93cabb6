#!/usr/bin/env python3
"""(not collected by pytest; lives under tests/ because it runs the oracle) Rate of the device ranker: user
sections per second against a prepared candidate set, HBM bytes per section (the candidate matrix is streamed once per user:
num_cand x k x 4 B) against the 8 TB/s peak, next to the C oracle's ranker on the same stream.  One JSON line.

    python tests/perf_ranker.py [--cand 100000] [--factor 128] [--sections 300] [--top-k 0]
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import svdfeature_amd as sa  # noqa: E402
from oracle import oracle  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cand", type=int, default=100_000)
    ap.add_argument("--users", type=int, default=100_000)
    ap.add_argument("--factor", type=int, default=128)
    ap.add_argument("--sections", type=int, default=300)
    ap.add_argument("--cpu-sections", type=int, default=20)
    ap.add_argument("--top-k", type=int, default=0)
    a = ap.parse_args()
    rng = np.random.default_rng(3)
    conf = [("num_user", a.users), ("num_item", a.cand), ("num_global", 0), ("num_factor", a.factor), ("ui_init_sigma", "0.1"), ("base_score", "3")]
    t = oracle.OracleTrainer("port", 0, 0)
    t.seed(10)
    for k, v in conf:
        t.set_param(k, v)
    t.init_model()
    t.init_trainer()
    tmp = tempfile.mkdtemp()
    path = os.path.join(tmp, "rank.model")
    t.save_model(path)
    t.close()
    items = sa.CSRData.from_rows([(0.0, [], [], [(c, 1.0)]) for c in range(a.cand)])
    secs = []
    for s in range(a.sections):
        pos = rng.choice(a.cand, size=5, replace=False)
        secs.append(sa.CSRData.from_rows([(2.0, [], [(int(rng.integers(0, a.users)), 1.0)], []), (1.0, [], [(int(x), 1.0) for x in pos], []),
                                          (4.0, [], [], [])]))
    out = {"candidates": a.cand, "factor": a.factor, "top_k": a.top_k}
    res = {}
    for name, mk, nsec in (("gpu", lambda: sa.Ranker(0, 0), a.sections), ("cpu_port", lambda: oracle.OracleRanker("port", 0, 0), a.cpu_sections)):
        r = mk()
        r.set_param("top_k", str(a.top_k))
        r.load_model(path)
        r.init_ranker(a.cand)
        t0 = time.time()
        r.process_rows(items)
        r.process_rows(secs[0])    # first section prepares the candidate matrix
        prep = time.time() - t0
        t0 = time.time()
        got = [r.process_rows(s) for s in secs[1:nsec]] or [np.zeros(0, np.int32)]
        dt = (time.time() - t0) / max(1, nsec - 1)
        res[name] = np.concatenate(got)
        out[name] = {"prepare_s": prep, "ms_per_section": dt * 1e3, "sections_per_s": 1.0 / dt}
        if name == "gpu":
            byts = a.cand * (a.factor * 4 + 4 + 4 + 1)
            out[name].update({"host_sorts": r.counter(1), "bytes_per_section": byts, "GBps": byts / dt / 1e9, "frac_of_8TBps": byts / dt / 8e12,
                              "note": "wall clock per process() section: staging of the section's lines, one pinned upload, k_rank_user, k_rank_score over the "
                                      "candidate matrix, k_rank_positions (or key + radix sort for top_k), one readback + sync"})
    # the same sections as ONE svdf_ranker_process_rows call: up to 8 sections in flight on the device
    r = sa.Ranker(0, 0)
    r.set_param("top_k", str(a.top_k))
    r.load_model(path)
    r.init_ranker(a.cand)
    r.process_rows(items)
    r.process_rows(secs[0])
    bulk = sa.CSRData.concat(secs[1:a.sections])
    r.process_rows(bulk)      # first call: every slot's staging / score buffers are allocated
    t0 = time.time()
    got = r.process_rows(bulk)
    dt = (time.time() - t0) / max(1, a.sections - 1)
    out["gpu_bulk"] = {"ms_per_section": dt * 1e3, "sections_per_s": 1.0 / dt, "host_sorts": r.counter(1), "tiles": r.counter(3),
                       "tie_copy_ms_total": r.counter(4) / 1e6, "tie_wait_ms_total": r.counter(5) / 1e6, "event_wait_ms_total": r.counter(6) / 1e6, "flush_ms_total": r.counter(7) / 1e6,
                       "GBps": out["gpu"]["bytes_per_section"] / dt / 1e9, "frac_of_8TBps": out["gpu"]["bytes_per_section"] / dt / 8e12,
                       "identical_to_per_section_calls": bool(np.array_equal(got, res["gpu"]))}
    # the same call with one scoring pass per section (amd:rank_tile = 0: the round-2 pipeline)
    r = sa.Ranker(0, 0)
    r.set_param("amd:rank_tile", "0")
    r.set_param("top_k", str(a.top_k))
    r.load_model(path)
    r.init_ranker(a.cand)
    r.process_rows(items)
    r.process_rows(secs[0])
    r.process_rows(bulk)
    t0 = time.time()
    got1 = r.process_rows(bulk)
    dt1 = (time.time() - t0) / max(1, a.sections - 1)
    out["gpu_bulk_untiled"] = {"ms_per_section": dt1 * 1e3, "sections_per_s": 1.0 / dt1, "identical_to_tiled": bool(np.array_equal(got, got1))}
    n = len(res["cpu_port"])
    if a.cpu_sections > 1:
        out["identical_results"] = bool(np.array_equal(res["gpu"][:n], res["cpu_port"]))
        out["speedup"] = out["cpu_port"]["ms_per_section"] / out["gpu"]["ms_per_section"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
