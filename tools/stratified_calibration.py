#!/usr/bin/env python3
"""Accuracy of the STRATIFIED window-minibatch schedule, oracle-backed, CPU only.

N ranks: user block r = users with id % N == r, item block b = items [NI b / N, NI (b+1) / N).  A pass is N sub-epochs; in
sub-epoch s rank r trains stratum (r, (r + s) % N) -- the pass's instances whose user is in block r and whose item is in block
(r + s) % N, in file order -- with the window-minibatch step in windows of at most `per_item` updates per item, applying the
item-side sums to ITS item block only (no sum over ranks: the block is exclusively owned during the sub-epoch), then hands the
block to rank r - 1.  Strata of one sub-epoch share neither users nor items, so one checker trainer running them one after the other
IS the N-rank run.  Same replica as tools/minibatch_calibration.py (configs[2] density)."""
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
from svdfeature_amd import CSRData  # noqa: E402
from svdfeature_amd.multi_gpu import stratum_windows  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--ratings", type=int, default=10_000_000)
ap.add_argument("--users", type=int, default=100_000)
ap.add_argument("--items", type=int, default=10_000)
ap.add_argument("--factor", type=int, default=64)
ap.add_argument("--passes", type=int, default=3)
ap.add_argument("--ranks", default="1,2,4,8")
ap.add_argument("--per-item", default="32")
a = ap.parse_args()
n = a.ratings
u, i, r = bench.synth_triples(n + 200_000, a.users, a.items)
tu, ti, tl = u[n:], i[n:], r[n:]
u, i, r = u[:n], i[:n], r[:n]
conf = bench.conf_for(a)
test = CSRData.from_triples(tu, ti, tl)


def rmse_of(t):
    p = t.predict_batch(test)
    return float(np.sqrt(np.mean((p.astype(np.float64) - tl) ** 2)))


t0 = time.time()
seq = mru.make_oracle(conf)
d_all = CSRData.from_triples(u, i, r)
for _ in range(a.passes):
    seq.update_batch(d_all)
ref = rmse_of(seq)
print("sequential reference path: held-out rmse %.6f after %d passes (%.0fs)" % (ref, a.passes, time.time() - t0), flush=True)
for per_item in [float(x) for x in a.per_item.split(",")]:
    for world in [int(x) for x in a.ranks.split(",")]:
        t0 = time.time()
        t = mru.make_oracle(conf)
        plan = [[stratum_windows(u, i, r, rk, world, s, a.items, per_item) for rk in range(world)] for s in range(world)]
        nwin = sum(len(w) for w in plan[0])
        for _ in range(a.passes):
            for s in range(world):
                for rk in range(world):
                    for (wu, wi, wr) in plan[s][rk]:
                        dW, db, dg = t.update_batch_stale(CSRData.from_triples(wu, wi, wr))
                        t.set_view("W_item", t.view("W_item") + dW)
                        t.set_view("i_bias", t.view("i_bias") + db)
        got = rmse_of(t)
        print("stratified  N=%d, <= %.0f updates per item per window (%d window steps per rank and pass): rmse %.6f  d=%+.2e  (%.0fs)" % (
            world, per_item, nwin // world * world if world else 0, got, got - ref, time.time() - t0), flush=True)
