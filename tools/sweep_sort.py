#!/usr/bin/env python3
"""Effect of the instance order INSIDE a conflict-free batch (free to choose: instances of a batch commute)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse, bench
import svdfeature_amd as sa
ap = argparse.ArgumentParser()
ap.add_argument("--ratings", type=int, default=100_000_000); ap.add_argument("--users", type=int, default=1_000_000)
ap.add_argument("--items", type=int, default=100_000); ap.add_argument("--factor", type=int, default=64)
a = ap.parse_args()
u, i, r = bench.synth_triples(a.ratings, a.users, a.items)
tr = sa.Trainer(0, 0); tr.seed(10)
for k, v in bench.conf_for(a): tr.set_param(k, v)
tr.init_model(); tr.init_trainer()
for mode, name in ((0, "file order"), (1, "by item"), (2, "by user"), (0, "file order")):
    tr.set_knob("sort_batches", mode)
    t0 = time.perf_counter(); ds = tr.dataset_from_triples(u, i, r); tb = time.perf_counter() - t0
    tr.train_dataset(ds); tr.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): tr.train_dataset(ds)
    tr.synchronize(); dt = (time.perf_counter() - t0) / 3
    print("batch order %-10s: build %.1fs  %.2f ms/pass  %.3f G inst/s  %.1f%% of 8 TB/s" % (name, tb, dt * 1e3, a.ratings / dt / 1e9, ds.algorithmic_bytes / dt / 8e12 * 100), flush=True)
    ds.close()
