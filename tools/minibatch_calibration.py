#!/usr/bin/env python3
"""Accuracy contract of the WINDOW-MINIBATCH multi-GPU step, oracle-backed, CPU only.

Replica of BASELINE configs[2] at its density (1000 ratings per item and 100 per user per pass; 100 K users x 10 K items,
10 M ratings, k = 64, demo/basicMF hyper-parameters), 3 passes.  N CPU checkers play the N ranks: every rank runs
svdo_update_csr_batch_stale on its user shard of a window (user side exact, item side read at the window start), the
deltas are summed in rank order and added on every rank.  Prints the held-out RMSE of the sequential reference path and
|dRMSE| per (ranks, windows); the rule bench.py derives from it is the smallest window count with |dRMSE| <= 1e-4 and
a factor ~2 of margin.  Also prints the round-2 scheme (exact levels per rank, stale across ranks) for comparison."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import multi_rank_utils as mru  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--ratings", type=int, default=10_000_000)
ap.add_argument("--users", type=int, default=100_000)
ap.add_argument("--items", type=int, default=10_000)
ap.add_argument("--factor", type=int, default=64)
ap.add_argument("--passes", type=int, default=3)
ap.add_argument("--ranks", default="1,2,8")
ap.add_argument("--windows", default="16,24,32,48,64")
ap.add_argument("--old", default="", help="ranks:windows pairs of the round-2 scheme to print next to it, e.g. 8:32,2:16")
a = ap.parse_args()
n = a.ratings
u, i, r = bench.synth_triples(n + 200_000, a.users, a.items)
tu, ti, tl = u[n:], i[n:], r[n:]
u, i, r = u[:n], i[:n], r[:n]
conf = bench.conf_for(a)


def rmse_of(ranks, world):
    p = mru.merged_predict(ranks, world, tu, ti, tl)
    return float(np.sqrt(np.mean((p.astype(np.float64) - tl) ** 2)))


t0 = time.time()
seq = mru.simulate(conf, u, i, r, 1, 1, a.passes)
ref = rmse_of(seq, 1)
print("sequential reference path: held-out rmse %.6f after %d passes (%.0fs)" % (ref, a.passes, time.time() - t0), flush=True)
for w in [int(x) for x in a.windows.split(",")]:
    for world in [int(x) for x in a.ranks.split(",")]:
        t0 = time.time()
        got = rmse_of(mru.simulate(conf, u, i, r, world, w, a.passes, minibatch=True), world)
        print("window-minibatch  N=%d x %3d windows (%.1f updates per item per window): rmse %.6f  d=%+.2e  (%.0fs)" % (
            world, w, n / a.items / w, got, got - ref, time.time() - t0), flush=True)
for pair in [p for p in a.old.split(",") if p]:
    world, w = [int(x) for x in pair.split(":")]
    t0 = time.time()
    got = rmse_of(mru.simulate(conf, u, i, r, world, w, a.passes), world)
    print("round-2 scheme    N=%d x %3d windows: rmse %.6f  d=%+.2e  (%.0fs)" % (world, w, got, got - ref, time.time() - t0), flush=True)
