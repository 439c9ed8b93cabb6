"""The window step on Zipf(0.7) items (secondary.orders.zipf_c2.window at a fraction of its size): ms per pass and windows, for kernel-level
profiles of the small windows the max-updates rule cuts.  python tools/zipf_window_probe.py [ratings] [passes]"""
import os
import sys
import time
import types

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from benchlib import orders  # noqa: E402
import svdfeature_amd as sa  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 2
a = types.SimpleNamespace(users=1_000_000, items=100_000, factor=64)
u, i, r = orders.synth_zipf_triples(types.SimpleNamespace(Planted=bench.Planted), n, a.users, a.items, 4321)
t = sa.Trainer(0, 0)
t.seed(10)
for k, v in [("learning_rate", "0.005"), ("wd_item", "0.004"), ("wd_user", "0.004"), ("num_item", str(a.items)), ("num_user", str(a.users)), ("num_factor", "64"),
             ("base_score", "3"), ("num_global", "0"), ("amd:step", "minibatch"), ("amd:contrib", "bf16")]:
    t.set_param(k, v)
t.init_model()
t.init_trainer()
for kv in os.environ.get("SVDF_KNOBS", "").split(","):
    if "=" in kv:
        t.set_knob(kv.split("=")[0], int(kv.split("=")[1]))
t0 = time.perf_counter()
ds = t.dataset_from_triples(u, i, r)
build = time.perf_counter() - t0
t.train_dataset(ds)
t.synchronize()
t0 = time.perf_counter()
for _ in range(passes):
    t.train_dataset(ds)
t.synchronize()
dt = (time.perf_counter() - t0) / passes
print("zipf window step: %d ratings, %d windows, build %.2f s, %.2f ms per pass = %.1f M inst/s, %.1f us per window" % (
    n, ds.num_batches, build, dt * 1e3, n / dt / 1e6, dt * 1e6 / max(ds.num_batches, 1)))
