#!/usr/bin/env python3
"""GPU tuning sweep for the basicMF kernel launch knobs (run through gpurun).
Builds the BASELINE configs[1] workload once and times passes for every (groups_per_wave, block_threads)."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import svdfeature_amd as sa

ap = argparse.ArgumentParser()
ap.add_argument("--ratings", type=int, default=100_000_000)
ap.add_argument("--users", type=int, default=1_000_000)
ap.add_argument("--items", type=int, default=100_000)
ap.add_argument("--factor", type=int, default=64)
ap.add_argument("--passes", type=int, default=3)
a = ap.parse_args()
u, i, r = bench.synth_triples(a.ratings, a.users, a.items)
tr = sa.Trainer(0, 0)
tr.seed(10)
for k, v in bench.conf_for(a):
    tr.set_param(k, v)
tr.init_model(); tr.init_trainer()
ds = tr.dataset_from_triples(u, i, r)
print("batches", ds.num_batches, "max", ds.max_batch, flush=True)
tr.train_dataset(ds); tr.synchronize()
for rep in range(2):
    for gpw in (1, 2, 4, 8):
        for bt in (64, 128, 256):
            for sm in ((0, 1, 2) if (gpw, bt) == (4, 128) else (0,)):
                tr.set_knob("groups_per_wave", gpw); tr.set_knob("block_threads", bt); tr.set_knob("store_mode", sm)
                tr.synchronize(); t0 = time.perf_counter()
                for _ in range(a.passes):
                    tr.train_dataset(ds)
                tr.synchronize(); dt = (time.perf_counter() - t0) / a.passes
                print("gpw %d block %3d store_mode %d : %.2f ms/pass  %.3f G inst/s  %.1f%% of 8 TB/s"
                      % (gpw, bt, sm, dt * 1e3, a.ratings / dt / 1e9, ds.algorithmic_bytes / dt / 8e12 * 100), flush=True)
