cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r05i
timeout 900 python -m pytest tests/test_gpu_pivot.py -x -q 2>&1 | grep -v amdgpu.ids | grep -E "passed|failed|Error" | head
cat > /tmp/z.py <<'PY'
import sys, time, types
import numpy as np
sys.path.insert(0, "."); import bench
from benchlib import orders
import svdfeature_amd as sa
n = 100_000_000
a = types.SimpleNamespace(users=1_000_000, items=100_000, factor=64, globals=0)
u, i, r = orders.synth_zipf_triples(types.SimpleNamespace(Planted=bench.Planted), n, a.users, a.items, 4321)
for run, long_, pmin in ((128, 128, 4096), (128, 1024, 2048), (128, 1024, 1024), (256, 256, 2048), (256, 1024, 1024), (128, 128, 1024), (192, 192, 1024)):
    t = bench.make_trainer(sa, "basicmf", a, 64, 0)
    t.set_knob("pivot_run", run); t.set_knob("pivot_run_long", long_); t.set_knob("pivot_min", pmin)
    t0 = time.time(); ds = t.dataset_from_triples(u, i, r); b = time.time() - t0
    ms = []
    for _ in range(2):
        t.synchronize(); t0 = time.perf_counter(); t.train_dataset(ds); t.synchronize(); ms.append((time.perf_counter() - t0) * 1e3)
    print("pivot_run %d / %d, pivot_min %d: build %.1f s, %.1f ms per pass = %.1f M inst/s, %d levels" % (run, long_, pmin, b, min(ms), n / min(ms) / 1e3, ds.num_batches), flush=True)
    ds.close(); t.close()
# ratings SORTED BY USER (the order of demo/basicMF/ua.base), items uniform / zipf
for name, items in (("user-sorted, uniform items", None), ("user-sorted, Zipf items", i)):
    m = 20_000_000
    uu = np.repeat(np.arange(m // 100, dtype=np.uint32), 100)
    ii = np.random.default_rng(5).integers(0, a.items, m, dtype=np.uint32) if items is None else items[:m].copy()
    rr = r[:m]
    for pmin in (4096,):
        t = bench.make_trainer(sa, "basicmf", a, 64, 0)
        t.set_knob("pivot_min", pmin)
        t0 = time.time(); ds = t.dataset_from_triples(uu, ii, rr); b = time.time() - t0
        ms = []
        for _ in range(2):
            t.synchronize(); t0 = time.perf_counter(); t.train_dataset(ds); t.synchronize(); ms.append((time.perf_counter() - t0) * 1e3)
        print("%s, 20 M ratings, pivot_min %d: kind %d, build %.1f s, %.1f ms per pass = %.1f M inst/s, %d levels" % (name, pmin, ds.kind, b, min(ms), m / min(ms) / 1e3, ds.num_batches), flush=True)
        ds.close(); t.close()
PY
SVDF_QUIET=1 timeout 1200 python /tmp/z.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05i/zipf_pivot_runs.txt
