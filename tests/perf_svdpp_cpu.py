#!/usr/bin/env python3
"""(not collected by pytest; lives under tests/ because it loads the CPU checkers in oracle/)
Single-thread throughput of the reference's SVD++ path on the user-block workload of tools/bench_variants.py
(100 ratings per user, feedback set = the user's items, k=128), for the comparison quoted in DESIGN.md."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle
from svdfeature_amd.data import CSRData, PlusBlock

K, NI, NU = 128, 100_000, 1_000_000
rng = np.random.default_rng(7)
nn = 400_000
uu = np.sort(rng.integers(0, nn // 100, nn, dtype=np.uint32), kind="stable")
ii = rng.integers(0, NI, nn, dtype=np.uint32)
rr = rng.integers(1, 6, nn).astype(np.float32)
starts = np.flatnonzero(np.r_[True, uu[1:] != uu[:-1]])
ends = np.r_[starts[1:], nn]
blocks = []
for s, e in zip(starts, ends):
    fb = np.unique(ii[s:e])
    blocks.append(PlusBlock(fb, np.full(fb.size, 1.0 / np.sqrt(fb.size), np.float32), CSRData.from_triples(uu[s:e], ii[s:e], rr[s:e])))
kind = "reference" if oracle.have_reference() else "port"
c = oracle.OracleTrainer(kind, 1, 0)
c.seed(10)
for k, v in [("learning_rate", "0.005"), ("wd_item", "0.004"), ("wd_user", "0.004"), ("num_item", NI), ("num_user", NU), ("num_factor", K),
             ("base_score", "3"), ("num_global", "0"), ("num_ufeedback", NI), ("wd_ufeedback", "0.004")]:
    c.set_param(k, v)
c.init_model(); c.init_trainer()
t0 = time.perf_counter()
for b in blocks:
    c.update_block(b)
dt = time.perf_counter() - t0
print(json.dumps({"case": "svdpp %s CPU, 1 thread, k=%d" % (kind, K), "instances": nn, "inst_per_s": nn / dt}))
