#!/usr/bin/env python3
"""Secondary measurements (not the bench.py contract line): throughput and roofline fraction of the other
kernels / shapes named in BASELINE.json configs[3..4] on one MI355X.  Run through gpurun.

  basic128   basicMF k=128                                  (k_basicmf<32>)
  pairwise   BPR-style pairs: nu=1, ni=2 (+1/-1), k=128     (general path, active_type=3, no_user_bias)
  neighbor   4 global features out of 10K + u + i, k=128    (general path; globals serialise)
  svdpp      user blocks, feedback set = the user's items   (k_svdpp, k=128)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import svdfeature_amd as sa
from svdfeature_amd.data import CSRData, PlusBlock

ap = argparse.ArgumentParser()
ap.add_argument("--which", default="basic128,pairwise,neighbor,svdpp")
ap.add_argument("--n", type=int, default=20_000_000)
ap.add_argument("--users", type=int, default=1_000_000)
ap.add_argument("--items", type=int, default=100_000)
ap.add_argument("--factor", type=int, default=128)
ap.add_argument("--passes", type=int, default=3)
ap.add_argument("--use-graph", type=int, default=0)
ap.add_argument("--knob", action="append", default=[], help="extra tuning knob name=value, repeatable")
a = ap.parse_args()


def mk(format_type, active_type, extra):
    t = sa.Trainer(format_type, active_type)
    t.seed(10)
    conf = [("learning_rate", "0.005"), ("wd_item", "0.004"), ("wd_user", "0.004"), ("num_item", a.items), ("num_user", a.users),
            ("num_factor", a.factor)] + extra
    for k, v in conf:
        t.set_param(k, v)
    t.init_model()
    t.init_trainer()
    t.set_knob("use_graph", a.use_graph)
    for kv in a.knob:
        t.set_knob(kv.split("=")[0], int(kv.split("=")[1]))
    return t


def timed(t, fn, n_inst, alg_bytes, label, extra=""):
    fn()
    t.synchronize()
    l0 = t.counter(1)
    t0 = time.perf_counter()
    for _ in range(a.passes):
        fn()
    t.synchronize()
    dt = (time.perf_counter() - t0) / a.passes
    launches = (t.counter(1) - l0) / a.passes
    print(json.dumps({"case": label, "instances": n_inst, "ms_per_pass": dt * 1e3, "inst_per_s": n_inst / dt,
                      "launches_per_pass": launches, "alg_GBps": alg_bytes / dt / 1e9, "frac_of_8TBps": alg_bytes / dt / 8e12,
                      "note": extra}), flush=True)


rng = np.random.default_rng(7)
n = a.n
u = rng.integers(0, a.users, n, dtype=np.uint32)
i = rng.integers(0, a.items, n, dtype=np.uint32)
r = rng.integers(1, 6, n).astype(np.float32)
which = a.which.split(",")

if "basic128" in which:
    t = mk(0, 0, [("base_score", "3"), ("num_global", "0")])
    ds = t.dataset_from_triples(u, i, r)
    timed(t, lambda: t.train_dataset(ds), n, ds.algorithmic_bytes, "basicMF k=%d" % a.factor, "batches %d" % ds.num_batches)
    ds.close(); t.close()

if "pairwise" in which:
    t = mk(0, 3, [("num_global", "0"), ("no_user_bias", "1")])
    j = rng.integers(0, a.items, n, dtype=np.uint32)
    j = np.where(j == i, (j + 1) % a.items, j).astype(np.uint32)
    lo, hi = np.minimum(i, j), np.maximum(i, j)
    row_ptr = np.empty(3 * n + 1, np.int64)
    base = 3 * np.arange(n, dtype=np.int64)
    row_ptr[0:3 * n:3] = base; row_ptr[1:3 * n:3] = base; row_ptr[2:3 * n:3] = base + 1; row_ptr[3 * n] = 3 * n
    idx = np.empty(3 * n, np.uint32); idx[0::3] = u; idx[1::3] = lo; idx[2::3] = hi
    val = np.empty(3 * n, np.float32); val[0::3] = 1.0
    val[1::3] = np.where(lo == i, 1.0, -1.0); val[2::3] = np.where(hi == i, 1.0, -1.0)
    d = CSRData(np.ones(n, np.float32), row_ptr.astype(np.int32), idx, val)
    ds = t.dataset_from_csr(d)
    timed(t, lambda: t.train_dataset(ds), n, ds.algorithmic_bytes, "pairwise nu=1 ni=2 k=%d" % a.factor,
          "kind %d batches %d" % (ds.kind, ds.num_batches))
    ds.close(); t.close()

for ng, G, relax in ((4, 10000, 0), (4, 4_000_000, 0), (4, 10000, 1)):
    # 10 K global ids: every id is shared by ~1600 instances, exact order serialises on them; 4 M ids (item-pair style
    # neighbourhood weights): conflicts are rare and the batches are as large as basicMF's
    if "neighbor" not in which:
        break
    t = mk(0, 0, [("base_score", "3"), ("num_global", str(G)), ("wd_global", "0.001")] + ([("amd:relax_global", "1")] if relax else []))
    nn = min(n, 4_000_000) if (G <= 100_000 and not relax) else n
    g = rng.integers(0, G, (nn, ng), dtype=np.uint32)
    row_ptr = np.empty(3 * nn + 1, np.int64)
    base = (ng + 2) * np.arange(nn, dtype=np.int64)
    row_ptr[0:3 * nn:3] = base; row_ptr[1:3 * nn:3] = base + ng; row_ptr[2:3 * nn:3] = base + ng + 1; row_ptr[3 * nn] = (ng + 2) * nn
    idx = np.empty((nn, ng + 2), np.uint32); idx[:, :ng] = g; idx[:, ng] = u[:nn]; idx[:, ng + 1] = i[:nn]
    val = np.ones((nn, ng + 2), np.float32); val[:, :ng] = rng.uniform(0, 1, (nn, ng))
    d = CSRData(r[:nn], row_ptr.astype(np.int32), idx.ravel(), val.ravel())
    ds = t.dataset_from_csr(d)
    timed(t, lambda: t.train_dataset(ds), nn, ds.algorithmic_bytes, "neighborhood ng=4 of %d global ids k=%d%s" % (G, a.factor, ", RELAXED globals" if relax else ""),
          "kind %d batches %d" % (ds.kind, ds.num_batches))
    ds.close(); t.close()

if "sidefeat" in which:
    # SURVEY 8(d2) side-feature variant: 4 global ids out of 10 K (value U(0,1)) + one extra user-feature id (64 "age
    # bucket" ids after the real users).  Exact order serialises on the shared ids; the opt-in relaxed mode (amd:relax_*)
    # updates them with atomics instead.
    ng, G, NB = 4, 10000, 64
    for relaxed in (0, 1):
        nn = min(n, 2_000_000) if not relaxed else n
        extra = [("base_score", "3"), ("num_global", str(G)), ("wd_global", "0.001"), ("num_user", a.users + NB)]
        if relaxed: extra += [("amd:relax_global", "1"), ("amd:relax_user_from", str(a.users))]
        t = mk(0, 0, extra)
        g = np.sort(rng.integers(0, G, (nn, ng), dtype=np.uint32), axis=1)
        per = ng + 3
        row_ptr = np.empty(3 * nn + 1, np.int64)
        base = per * np.arange(nn, dtype=np.int64)
        row_ptr[0:3 * nn:3] = base; row_ptr[1:3 * nn:3] = base + ng; row_ptr[2:3 * nn:3] = base + ng + 2; row_ptr[3 * nn] = per * nn
        idx = np.empty((nn, per), np.uint32); idx[:, :ng] = g; idx[:, ng] = u[:nn]; idx[:, ng + 1] = a.users + (u[:nn] % NB); idx[:, ng + 2] = i[:nn]
        val = np.ones((nn, per), np.float32); val[:, :ng] = rng.uniform(0, 1, (nn, ng))
        d = CSRData(r[:nn], row_ptr.astype(np.int32), idx.ravel(), val.ravel())
        ds = t.dataset_from_csr(d)
        timed(t, lambda: t.train_dataset(ds), nn, ds.algorithmic_bytes, "side features (4 of 10K globals + 1 of 64 shared user ids) k=%d, %s" % (a.factor, "RELAXED shared ids" if relaxed else "exact"),
              "kind %d batches %d" % (ds.kind, ds.num_batches))
        ds.close(); t.close()

if "svdpp" in which:
    # user-grouped: sort the ratings by user, feedback set = the user's items with value n^-1/2
    nn = min(n, 4_000_000)
    nusers = max(1, nn // 100)
    uu = rng.integers(0, nusers, nn, dtype=np.uint32)
    order = np.argsort(uu, kind="stable")
    uu, ii, rr = uu[order], i[:nn][order], r[:nn][order]
    t = mk(1, 0, [("base_score", "3"), ("num_global", "0"), ("num_ufeedback", a.items), ("wd_ufeedback", "0.004")])
    starts = np.flatnonzero(np.r_[True, uu[1:] != uu[:-1]])
    ends = np.r_[starts[1:], nn]
    blocks = []
    perm = rng.permutation(len(starts))
    for b in perm:
        s, e = starts[b], ends[b]
        fb = np.unique(ii[s:e])
        blocks.append(PlusBlock(fb, np.full(fb.size, 1.0 / np.sqrt(fb.size), np.float32), CSRData.from_triples(uu[s:e], ii[s:e], rr[s:e])))
    k = a.factor
    for simple in (1, 0, 2):   # 2 = fast path with RELAXED item / feedback rows (Hogwild on the rows users share)
        if simple == 2:
            t = mk(1, 0, [("base_score", "3"), ("num_global", "0"), ("num_ufeedback", a.items), ("wd_ufeedback", "0.004"),
                          ("amd:relax_item_from", "0"), ("amd:relax_feedback", "1")])
        elif simple == 0:
            t = mk(1, 0, [("base_score", "3"), ("num_global", "0"), ("num_ufeedback", a.items), ("wd_ufeedback", "0.004")])
        t.set_knob("use_simple_units", 1 if simple else 0)
        t0 = time.perf_counter()
        ds = t.dataset_from_blocks(blocks)
        build_s = time.perf_counter() - t0
        timed(t, lambda: t.train_dataset(ds), nn, ds.algorithmic_bytes, "svdpp user blocks k=%d resident dataset, %s" % (k, ("simple_units=%d" % simple) if simple < 2 else "RELAXED item+feedback rows"),
              "%d users, %d on fast path, %d batches, dataset build %.1fs" % (ds.num_units, ds.num_simple_units, ds.num_batches, build_s))
        ds.close(); t.close()
