"""Rate of the exact pass on the generator's pair order (benchlib/orders.py: user blocks in random order, 200 consecutive pairs each, 100 K items, k = 128):
level by level (knob pair_units = 0: chained narrow levels), user-run units at several caps, the reference CPU path.  python tools/punit_probe.py [pairs]"""
import sys
import time
import types

import numpy as np

sys.path.insert(0, ".")
import bench
from benchlib import orders
import svdfeature_amd as sa

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
a = types.SimpleNamespace(users=1_000_000, items=100_000, factor=128, globals=0, pairs=n)
u, p, q = orders.synth_generator_pairs(types.SimpleNamespace(Planted=bench.Planted), n, a.users, a.items, 200, 99)


def run(knobs):
    t = bench.make_trainer(sa, "pairwise", a, 128, 0)
    for k, v in knobs:
        t.set_knob(k, v)
    t0 = time.perf_counter()
    ds = t.dataset_from_pairs(u, p, q)
    t.synchronize()
    build = time.perf_counter() - t0
    t.train_dataset(ds); t.synchronize()
    t0 = time.perf_counter()
    t.train_dataset(ds); t.synchronize()
    dt = time.perf_counter() - t0
    out = (ds.kind, ds.num_batches, n / dt / 1e6, build, t.view("W_item").copy())
    ds.close(); t.close()
    return out


base = run([("pair_units", 0)])
print("level by level: kind %d, %d levels, %.2f M pairs/s, build %.2f s" % base[:4], flush=True)
for cap in (8, 16, 24, 32, 48, 64):
    r = run([("pair_unit_cap", cap)])
    print("user-run units, cap %3d: kind %d, %d levels, %.2f M pairs/s, build %.2f s, same bits: %s" % ((cap,) + r[:4] + (np.array_equal(r[4].view(np.uint32), base[4].view(np.uint32)),)), flush=True)
try:
    from oracle import oracle
    oracle.build()
    kind = "reference" if oracle.have_reference() else "port"
    o = oracle.OracleTrainer(kind, 0, 3)
    o.seed(10)
    for k, v in bench.workload_conf("pairwise", a, 128):
        o.set_param(k, v)
    o.init_model(); o.init_trainer()
    m = min(n, 2_000_000)
    csr = sa.pairs_as_csr(u[:m], p[:m], q[:m])
    t0 = time.perf_counter(); o.update_batch(csr); dt = time.perf_counter() - t0
    print("CPU (%s, 1 thread): %.2f M pairs/s" % (kind, m / dt / 1e6))
except Exception as e:
    print("cpu baseline skipped: %r" % (e,))
