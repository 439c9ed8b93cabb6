#!/usr/bin/env python3
"""SVD++ unit latency probe: users with disjoint item sets (one conflict-free batch), varying the number of users."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import svdfeature_amd as sa
from svdfeature_amd.data import CSRData, PlusBlock
K = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ROWS = 100
for nusers in (1, 8, 64, 512, 4096, 20000):
    ni = nusers * ROWS
    t = sa.Trainer(1, 0)
    t.seed(10)
    for k, v in [("learning_rate", "0.005"), ("wd_item", "0.004"), ("wd_user", "0.004"), ("num_item", ni), ("num_user", nusers),
                 ("num_factor", K), ("base_score", "3"), ("num_global", "0"), ("num_ufeedback", ni), ("wd_ufeedback", "0.004")]:
        t.set_param(k, v)
    t.init_model(); t.init_trainer()
    blocks = []
    for b in range(nusers):
        items = np.arange(b * ROWS, (b + 1) * ROWS, dtype=np.uint32)
        blocks.append(PlusBlock(items, np.full(ROWS, 0.1, np.float32), CSRData.from_triples(np.full(ROWS, b, np.uint32), items, np.full(ROWS, 4.0, np.float32))))
    for simple in (1, 0):
        t.set_knob("use_simple_units", simple)
        ds = t.dataset_from_blocks(blocks)
        t.train_dataset(ds); t.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            t.train_dataset(ds)
        t.synchronize()
        dt = (time.perf_counter() - t0) / 5
        print(json.dumps({"k": K, "users": nusers, "simple": simple, "batches": ds.num_batches, "us_per_pass": dt * 1e6,
                          "us_per_row_latency": dt * 1e6 / ROWS, "M_inst_per_s": nusers * ROWS / dt / 1e6}), flush=True)
        ds.close()
    t.close()
