#!/usr/bin/env python3
"""SVD++ single-unit latency split: rows vs feedback list length (one user, fast path)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import svdfeature_amd as sa
from svdfeature_amd.data import CSRData, PlusBlock
KS = [int(x) for x in os.environ.get("SVDPP_K", "16,32,64,128,256").split(",")]
for K in KS:
    for rows, nfb in ((1, 1), (100, 1), (400, 1), (1, 100), (1, 400), (100, 100)):
        ni = max(rows, nfb) + 8
        t = sa.Trainer(1, 0)
        t.seed(10)
        for k, v in [("learning_rate", "0.005"), ("wd_item", "0.004"), ("wd_user", "0.004"), ("num_item", ni), ("num_user", 4),
                     ("num_factor", K), ("base_score", "3"), ("num_global", "0"), ("num_ufeedback", ni), ("wd_ufeedback", "0.004")]:
            t.set_param(k, v)
        t.init_model(); t.init_trainer()
        items = np.arange(rows, dtype=np.uint32)
        b = PlusBlock(np.arange(nfb, dtype=np.uint32), np.full(nfb, 0.1, np.float32), CSRData.from_triples(np.zeros(rows, np.uint32), items, np.full(rows, 4.0, np.float32)))
        ds = t.dataset_from_blocks([b])
        t.train_dataset(ds); t.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            t.train_dataset(ds)
        t.synchronize()
        dt = (time.perf_counter() - t0) / 20
        print(json.dumps({"k": K, "rows": rows, "nfb": nfb, "us": round(dt * 1e6, 1)}), flush=True)
        ds.close(); t.close()
