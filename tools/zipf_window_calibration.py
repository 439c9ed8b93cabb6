"""Calibration of `window_per_target_max` (svdf_wunit.cpp): BASELINE configs[1] with Zipf(0.7) items (benchlib/orders.py), ratings through the exact
pass and through the window step with the hot item bounded at 32 ... 512 updates per window; held-out RMSE after 3 passes against the exact run's.
python tools/zipf_window_calibration.py [ratings] [exponent]"""
import sys
import time
import types

import numpy as np

sys.path.insert(0, ".")
import bench
from benchlib import orders
import svdfeature_amd as sa

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
if len(sys.argv) > 2:
    orders.ZIPF_EXPONENT = float(sys.argv[2])
a = types.SimpleNamespace(users=1_000_000, items=100_000, factor=64, globals=0)
ctx = types.SimpleNamespace(Planted=bench.Planted)
u, i, r = orders.synth_zipf_triples(ctx, n + 200_000, a.users, a.items, 4321)
test = sa.CSRData.from_triples(u[n:], i[n:], r[n:])
cnt = np.bincount(i[:n], minlength=a.items)
print("Zipf(%.2f): %d ratings, top item %d (%.2f %%), sum c^2 / sum c = %.0f" % (orders.ZIPF_EXPONENT, n, cnt.max(), 100.0 * cnt.max() / n, float((cnt.astype(np.float64) ** 2).sum() / n)), flush=True)
PASSES = 3


def run(extra, knobs=()):
    t = bench.make_trainer(sa, "basicmf", a, 64, 0, extra=extra)
    for k, v in knobs:
        t.set_knob(k, v)
    ds = t.dataset_from_triples(u[:n], i[:n], r[:n])
    ms = []
    for _ in range(PASSES):
        t.synchronize()
        t0 = time.perf_counter()
        t.train_dataset(ds)
        t.synchronize()
        ms.append((time.perf_counter() - t0) * 1e3)
    if extra:
        import os, tempfile
        path = os.path.join(tempfile.mkdtemp(), "m")
        t.save_model(path)
        tw = sa.Trainer(0, 0)
        tw.load_model(path)
        tw.init_trainer()
        rm = bench.rmse(tw.predict_batch(test), r[n:])
        tw.close()
    else:
        rm = bench.rmse(t.predict_batch(test), r[n:])
    nb = ds.num_batches
    ds.close()
    t.close()
    return rm, min(ms), nb


ex, ms, lv = run([])
print("exact: rmse %.6f, %.1f ms per pass, %d levels" % (ex, ms, lv), flush=True)
for fmt in ("fp32", "bf16"):
    for cap in (32, 64, 96, 128, 256, 512):
        rm, ms, nw = run([("amd:step", "minibatch"), ("amd:contrib", fmt)], [("window_per_target_max", cap)])
        print("window step, contributions %s, max %3d updates per row and window: %5d windows, %.1f ms per pass = %.0f M inst/s, rmse %.6f (%+.2e)" % (
            fmt, cap, nw, ms, n / ms / 1e3, rm, rm - ex), flush=True)
