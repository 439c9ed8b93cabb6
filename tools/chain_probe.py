"""A rank pass in the reference's own FILE order (user-grouped pairs: the order PairwiseRankGenerator emits, apex_svd_data.cpp:946-965) on the demo's shape:
943 users x 1 682 items, ~3 400 pairs per user, k = 128.  Level by level (knob chain_width = 0) against runs of narrow levels inside one launch."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import cases
import svdfeature_amd as sa

nu, ni, per_user, k = 943, 1682, 3400, 128
rng = np.random.default_rng(1)
u = np.repeat(np.arange(nu, dtype=np.uint32), per_user)
p = rng.integers(0, ni, len(u)).astype(np.uint32)
q = ((p + 1 + rng.integers(0, ni - 1, len(u))) % ni).astype(np.uint32)
res, models = {}, []
for cw in (0, 96):
    t = sa.Trainer(0, 3)
    t.seed(10)
    for kk, v in cases.conf_with(cases.PAIR_CONF, num_user=nu, num_item=ni, num_factor=k):
        t.set_param(kk, v)
    t.init_model()
    t.init_trainer()
    t.set_knob("chain_width", cw)
    ds = t.dataset_from_pairs(u, p, q)
    t.train_dataset(ds)
    t.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        t.train_dataset(ds)
    t.synchronize()
    dt = (time.perf_counter() - t0) / 3
    res["chain_width=%d" % cw] = {"ms_per_pass": round(dt * 1e3, 2), "M_pairs_per_s": round(len(u) / dt / 1e6, 2), "levels": ds.num_batches,
                                 "levels_chained_per_pass": t.counter(15) // 4}
    models.append({n: t.view(n) for n in ("W_user", "W_item", "i_bias")})
res["bit_identical"] = all(np.array_equal(models[0][n].view(np.uint32), models[1][n].view(np.uint32)) for n in models[0])
res["pairs"] = len(u)
print(json.dumps(res))
