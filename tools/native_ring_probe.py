"""Host cost of the stratified hand-overs issued from C++ (svdf_rccl.cpp) at the step size of ONE RANK OF 8 (BASELINE configs[2]): 64 stratum steps per
pass of 195 K ratings on an item block of 6 250 rows, every block handed over after its step.  One GPU: the ring has one rank (the block goes to this rank
itself through ncclSend / ncclRecv and is put back two steps later) -- the calls, streams and events of the real ring, not its link time.
Prints per pass: host time until everything is enqueued and wall time, with and without the hand-overs."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import cases
import svdfeature_amd as sa
from svdfeature_amd.multi_gpu import HipShard, StratifiedTrainer, stratified_plan

n, nu, ni, k, chunks, P = 12_500_000, 125_000, 12_500, 64, 32, 2
rng = np.random.default_rng(5)
u = rng.integers(0, nu, n).astype(np.uint32)
i = rng.integers(0, ni, n).astype(np.uint32)
r = rng.integers(1, 6, n).astype(np.float32)
conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=k)
res = {}
for native in (False, True, "ipc"):
    t = sa.Trainer(0, 0)
    t.seed(10)
    for kk, v in conf:
        t.set_param(kk, v)
    t.init_model()
    t.init_trainer()
    ad = HipShard(t, torch, torch.device("cuda", 0), minibatch=True)
    ad.set_wire_half(False)
    plan = [[ad.make_windows(sub) for sub in chunk] for chunk in stratified_plan(u, i, r, 0, 1, chunks, ni, 32.0, blocks_per_rank=P)]
    if native == "ipc":
        ad.ipc_open(None, 0, 1, blocks=P)
        ad.ipc_self_ring = True
    elif native:
        ad.rccl_open(None, 0, 1)
        ad.rccl_self_ring = True
    st = StratifiedTrainer(ad, plan, 1, 0, None, blocks_per_rank=P)
    st.train_pass()
    t.synchronize()
    steps = 5
    t0 = time.perf_counter()
    for _ in range(steps):
        st.train_pass()
    enq = time.perf_counter() - t0
    t.synchronize()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    key = "with IPC hand-overs" if native == "ipc" else "with native hand-overs" if native else "no hand-overs"
    res[key] = {"enqueue_ms_per_pass": round(enq * 1e3 / steps, 3), "ms_per_pass": round(wall * 1e3 / steps, 3),
                "window_steps_per_pass": sum(len(s) for c in plan for s in c), "handoffs_per_pass": (64 if native == "ipc" else t.rccl_counter(0) // (steps + 1)) if native else 0}
    if native == "ipc":
        t.ipc_close()
    elif native:
        ad.rccl_close()
print(json.dumps(res))
