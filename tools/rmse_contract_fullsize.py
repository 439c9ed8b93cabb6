#!/usr/bin/env python3
"""Accuracy contract of the multi-GPU exchange at the FULL BASELINE configs[2] size, on ONE GPU: N trainers play the N
ranks (each with its user shard, windows and tail deferral exactly as bench.py builds them), the all-reduce is an
explicit sum of the packed fp16 deltas.  Prints held-out RMSE of the sequential single-GPU run and of N = 2, 4, 8."""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import torch
import svdfeature_amd as sa
from svdfeature_amd.multi_gpu import HipShard, defer_tails, shard_windows

ap = argparse.ArgumentParser()
ap.add_argument("--ratings", type=int, default=100_000_000); ap.add_argument("--users", type=int, default=1_000_000)
ap.add_argument("--items", type=int, default=100_000); ap.add_argument("--factor", type=int, default=64)
ap.add_argument("--passes", type=int, default=3); ap.add_argument("--ranks", default="2,4,8")
ap.add_argument("--windows", default="", help="override: one window count per entry of --ranks (default: bench.py's rule)")
a = ap.parse_args()
n = a.ratings
u, i, r = bench.synth_triples(n + 1_000_000, a.users, a.items)
tu, ti, tl = u[n:n + 200000], i[n:n + 200000], r[n:n + 200000]
u, i, r = u[:n], i[:n], r[:n]
dev = torch.device("cuda", 0)


def make():
    t = sa.Trainer(0, 0); t.seed(10)
    for k, v in bench.conf_for(a): t.set_param(k, v)
    t.init_model(); t.init_trainer()
    return t


def rmse(trainers, world):
    sse = 0.0
    for rk, t in enumerate(trainers):
        m = (tu % world) == rk
        p = t.predict_batch(sa.CSRData.from_triples(tu[m], ti[m], tl[m]))
        sse += float(np.sum((p.astype(np.float64) - tl[m]) ** 2))
    return float(np.sqrt(sse / len(tl)))


t = make()
ds = t.dataset_from_triples(u, i, r)
for _ in range(a.passes): t.train_dataset(ds)
ref = rmse([t], 1)
print("sequential (1 GPU, exact): rmse %.6f after %d passes" % (ref, a.passes), flush=True)
ds.close(); t.close()
worlds = [int(x) for x in a.ranks.split(",")]
override = [int(x) for x in a.windows.split(",")] if a.windows else None
for wi_, world in enumerate(worlds):
    per_item = a.ratings / a.items
    windows = override[wi_] if override else max(1, int(np.ceil(per_item / (64.0 if world <= 2 else (42.0 if world <= 4 else 32.0)))))
    t0 = time.time()
    ranks = []
    for rk in range(world):
        tr = make()
        ad = HipShard(tr, torch, dev)
        ad.set_wire_half(True)
        sh = defer_tails(shard_windows(u, i, r, rk, world, windows), a.users, a.items, 0.05)
        ranks.append((ad, ad.make_windows(sh)))
    for _ in range(a.passes):
        for w in range(windows):
            ds_ = []
            for ad, wins in ranks:
                if w == 0: ad.delta_begin()
                ad.train(wins[w])
                d = ad.delta_get(); ad.stream.synchronize()
                ds_.append(d.clone())
            total = ds_[0]
            for d in ds_[1:]: total = total + d       # fp16 sum, like the collective's wire format
            torch.cuda.synchronize()
            for ad, _ in ranks: ad.delta_set(total)
    got = rmse([ad.t for ad, _ in ranks], world)
    print("N=%d ranks x %d windows (fp16 deltas, tails deferred): rmse %.6f  d=%.2e  (%.0fs)" % (world, windows, got, got - ref, time.time() - t0), flush=True)
    for ad, wins in ranks:
        for w_ in wins: w_.close()
        ad.t.close()
