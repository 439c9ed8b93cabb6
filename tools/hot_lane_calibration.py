"""Calibration of the ordered sub-steps for hot items (svdf_k_window.hip: k_window_apply; knobs window_hot_sub / window_hot_max): BASELINE configs[1] with
Zipf(0.7) items (benchlib/orders.py), 3 passes through the exact pass and through the one-GPU window step -- the round-5 rule (window_hot_sub = 0: no row more
than 128 updates per window) and the hot lane with window_hot_max = 512 ... 4096 --, held-out RMSE against the exact run's, ms per pass.
python tools/hot_lane_calibration.py [ratings] [seeds] [hot_max list] [exponent]"""
import sys
import time
import types

import numpy as np

sys.path.insert(0, ".")
import bench
from benchlib import orders
import svdfeature_amd as sa

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
seeds = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0").split(",")]
caps = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "512,1024,2048,4096").split(",")]
if len(sys.argv) > 4:
    orders.ZIPF_EXPONENT = float(sys.argv[4])
a = types.SimpleNamespace(users=1_000_000, items=100_000, factor=64, globals=0)
ctx = types.SimpleNamespace(Planted=bench.Planted)
PASSES = 3
for seed in seeds:
    u, i, r = orders.synth_zipf_triples(ctx, n + 200_000, a.users, a.items, 4321 + seed)
    test = sa.CSRData.from_triples(u[n:], i[n:], r[n:])
    cnt = np.bincount(i[:n], minlength=a.items)
    print("seed %d, Zipf(%.2f): %d ratings, top item %d (%.2f %%)" % (seed, orders.ZIPF_EXPONENT, n, cnt.max(), 100.0 * cnt.max() / n), flush=True)

    def run(extra, knobs=()):
        t = bench.make_trainer(sa, "basicmf", a, 64, 0, extra=extra)
        for k, v in knobs:
            t.set_knob(k, v)
        t0 = time.perf_counter()
        ds = t.dataset_from_triples(u[:n], i[:n], r[:n])
        t.synchronize()
        build = time.perf_counter() - t0
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
        return rm, min(ms), nb, build

    import os
    if os.environ.get("SVDF_HOT_ONLY"):   # (profiling: only the hot-lane configurations)
        ex = 0.0
        for cap in caps:
            for sub in [int(x) for x in os.environ.get("SVDF_HOT_SUBS", "128").split(",")]:
                rm, ms, nw, bs = run([("amd:step", "minibatch")], [("window_hot_max", cap), ("window_hot_sub", sub), ("window_per_target", 100000)])
                print("  hot rows <= %4d per window, sub-steps of %d: %5d windows, %.1f ms per pass" % (cap, sub, nw, ms), flush=True)
        continue
    ex, ms, lv, bs = run([])
    print("  exact: rmse %.6f, %.1f ms per pass, %d levels, build %.1f s" % (ex, ms, lv, bs), flush=True)
    if seed == seeds[0]:
        rm, ms, nw, bs = run([("amd:step", "minibatch")], [("window_hot_sub", 0)])
        print("  window step, round-5 rule (no row > 128 per window): %5d windows, %.1f ms per pass = %.0f M inst/s, rmse %+.2e, build %.1f s" % (nw, ms, n / ms / 1e3, rm - ex, bs), flush=True)
    for cap in caps:
        rm, ms, nw, bs = run([("amd:step", "minibatch")], [("window_hot_max", cap)])
        print("  window step, sub-steps of 128, hot rows <= %4d per window: %5d windows, %.1f ms per pass = %.0f M inst/s (frac %.3f), rmse %+.2e, build %.1f s" % (
            cap, nw, ms, n / ms / 1e3, n / ms / 1e3 * 1072e6 / 8e12, rm - ex, bs), flush=True)
