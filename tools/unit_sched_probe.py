"""Schedule of a resident SVD++ data set on the device against the host scan (svdf_k_sched.hip: device_schedule_units vs
Engine::schedule_units): same digest, time of the schedule itself (counter 24).  python tools/unit_sched_probe.py [users] [per_user]"""
import os
import sys
import time
import types

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
import svdfeature_amd as sa  # noqa: E402

users = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
per = int(sys.argv[2]) if len(sys.argv) > 2 else 100
a = types.SimpleNamespace(users=1_000_000, items=100_000, factor=128)
t0 = time.perf_counter()
train, _ = bench.synth_user_blocks(users, per, a.users, a.items, 4242)
print("synthetic blocks: %d users x %d in %.1f s" % (users, per, time.perf_counter() - t0), flush=True)
res = {}
for dev in (1, 0):
    t = sa.Trainer(1, 0)
    t.seed(10)
    for k, v in bench.workload_conf("svdpp", a, 128):
        t.set_param(k, v)
    t.init_model()
    t.init_trainer()
    t.set_knob("device_schedule", dev)
    t0 = time.perf_counter()
    ds = t.dataset_from_blocks(train)
    dt = time.perf_counter() - t0
    res[dev] = ds.info(7)
    print("device_schedule=%d: dataset_from_blocks %.2f s, of it the schedule %.1f ms (on device: %d), %d levels, widest %d, fast-path units %d / %d"
          % (dev, dt, t.counter(24) / 1e3, t.counter(25), ds.num_batches, ds.max_batch, ds.num_simple_units, ds.num_units), flush=True)
    ds.close()
    t.close() if hasattr(t, "close") else None
print("same schedule (digest of level_ptr, level_mid, order):", res[0] == res[1])
