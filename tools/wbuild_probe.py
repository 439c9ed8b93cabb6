"""Build time of the one-GPU window sequence (amd:step = minibatch) of BASELINE configs[2] / [4] shaped data: host builder (knob device_window = 0)
against the device builder (svdf_k_wbuild.hip); one pass trained on each, models compared bit for bit."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import cases
import svdfeature_amd as sa


def run(pairs, n, nu, ni, k, dev):
    rng = np.random.default_rng(3)
    u = rng.integers(0, nu, n).astype(np.uint32)
    i = rng.integers(0, ni, n).astype(np.uint32)
    if pairs:
        c = (u, i, ((i + 1 + rng.integers(0, ni - 1, n)) % ni).astype(np.uint32))
    else:
        c = (u, i, rng.integers(1, 6, n).astype(np.float32))
    t = sa.Trainer(0, 3 if pairs else 0)
    t.seed(10)
    for kk, v in cases.conf_with(cases.PAIR_CONF if pairs else cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=k) + [("amd:step", "minibatch")]:
        t.set_param(kk, v)
    t.init_model()
    t.init_trainer()
    t.set_knob("device_window", dev)
    t0 = time.perf_counter()
    ds = t.dataset_from_pairs(*c) if pairs else t.dataset_from_triples(*c)
    t.synchronize()
    b = time.perf_counter() - t0
    t0 = time.perf_counter()
    t.train_dataset(ds)
    t.synchronize()
    p = time.perf_counter() - t0
    return b, p, ds.num_batches, {x: t.view(x) for x in ("W_user", "W_item", "i_bias")}


for name, pairs, n, k in (("configs[2] ratings 100 M, k=64", False, int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000, 64),
                          ("configs[4] pairs 200 M, k=128", True, int(sys.argv[2]) if len(sys.argv) > 2 else 200_000_000, 128)):
    h = run(pairs, n, 1_000_000, 100_000, k, 0)
    d = run(pairs, n, 1_000_000, 100_000, k, 1)
    same = all(np.array_equal(h[3][x].view(np.uint32), d[3][x].view(np.uint32)) for x in h[3])
    print(json.dumps({"data": name, "windows": d[2], "host_build_s": round(h[0], 3), "device_build_s": round(d[0], 3), "first_pass_s": round(d[1], 4),
                      "bit_identical_models": bool(same)}), flush=True)
