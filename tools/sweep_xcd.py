#!/usr/bin/env python3
"""A/B of the XCD-aware tile mapping of k_basicmf (xcd_remap knob) on the bench workload."""
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
for sort in (1, 2):
    tr.set_knob("sort_batches", sort)
    ds = tr.dataset_from_triples(u, i, r)
    for remap in (0, 1, 0, 1):
        tr.set_knob("xcd_remap", remap)
        tr.train_dataset(ds); tr.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): tr.train_dataset(ds)
        tr.synchronize(); dt = (time.perf_counter() - t0) / 3
        print("sort %d xcd_remap %d: %.2f ms/pass  %.3f G inst/s  %.1f%% of 8 TB/s" % (sort, remap, dt * 1e3, a.ratings / dt / 1e9, ds.algorithmic_bytes / dt / 8e12 * 100), flush=True)
    ds.close()
