"""Schedule build time of resident data sets (device scheduler vs host scheduler): neighbourhood rows (4 global ids each).
usage: python tools/sched_build_probe.py [n]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import bench
import svdfeature_amd as sa

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
import argparse
ARGS = argparse.Namespace(users=480_189, items=17_770, globals=10_000)
d = bench.synth_neighbourhood(n, ARGS.users, ARGS.items, ARGS.globals, 4)
for dev in (1, 0, 1):
    t = sa.Trainer(0, 0)
    for k, v in bench.workload_conf("neighbourhood", ARGS, 128):
        t.set_param(k, v)
    t.init_model(); t.init_trainer()
    t.set_knob("device_schedule", dev)
    t0 = time.time()
    ds = t.dataset_from_csr(d)
    el = time.time() - t0
    print("neighbourhood n=%d device_schedule=%d: %.3f s, %d batches (largest %d)" % (n, dev, el, ds.num_batches, ds.max_batch), flush=True)
