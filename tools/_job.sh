cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r05g
timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_parity.py -x -q 2>&1 | grep -v amdgpu.ids | grep -E "passed|failed" > gpurun_out/r05g/pytest.log; cat gpurun_out/r05g/pytest.log
cat > /tmp/z.py <<'PY'
import sys, time, types
import numpy as np
sys.path.insert(0, "."); import bench
from benchlib import orders
import svdfeature_amd as sa
n = 100_000_000
a = types.SimpleNamespace(users=1_000_000, items=100_000, factor=64, globals=0)
u, i, r = orders.synth_zipf_triples(types.SimpleNamespace(Planted=bench.Planted), n, a.users, a.items, 4321)
for cw, g in ((128, 1),):
    t = bench.make_trainer(sa, "basicmf", a, 64, 0)
    t.set_knob("chain_width", cw)
    ds = t.dataset_from_triples(u, i, r)
    ms = []
    for _ in range(2):
        t.synchronize(); t0 = time.perf_counter(); t.train_dataset(ds); t.synchronize(); ms.append((time.perf_counter() - t0) * 1e3)
    print("chain_g %d chain_width %d: %.1f ms per pass = %.1f M inst/s, %d levels, %d chained" % (g, cw, min(ms), n / min(ms) / 1e3, ds.num_batches, t.counter(15) // 2), flush=True)
    ds.close(); t.close()
PY
SVDF_QUIET=1 timeout 900 python /tmp/z.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05g/zipf_chain.txt
