"""(not collected by pytest) Randomised differential run of the one-GPU window step with ordered sub-steps (svdf_k_window.hip: k_window_apply,
k_window_users*; svdf_wunit.cpp: wseq_windows_hot) against oracle/svdf_oracle.c: svdo_update_window_substeps: random sizes, widths 1 .. 256, links, decays,
user bias on / off, sub-step sizes 1 .. 128, caps below and above the sub-step, uniform and Zipf items, 1 - 3 passes; every parameter compared bit for bit.
usage: python tests/fuzz_hot.py [--iters N] [--seed S]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases  # noqa: E402
import svdfeature_amd as sa  # noqa: E402
from oracle import oracle  # noqa: E402

NAMES = ("W_item", "i_bias", "W_user", "u_bias")


def one(rng, case):
    nu, ni = int(rng.integers(20, 4000)), int(rng.integers(3, 600))
    n = int(rng.integers(500, 60000))
    k = int(rng.choice([1, 2, 3, 4, 7, 8, 12, 16, 24, 31, 32, 48, 64, 64, 64, 96, 100, 128, 128, 160, 256]))
    sub = int(rng.choice([1, 2, 3, 8, 16, 33, 64, 100, 128]))
    cap = int(sub * rng.choice([0.5, 1, 2, 7, 20, 100]) + 1)
    passes = int(rng.integers(1, 4))
    active = int(rng.choice([0, 0, 0, 1, 2, 5]))
    u, i, r = cases.planted_triples(n, nu, ni, seed=int(rng.integers(0, 1 << 30)), zipf=bool(rng.integers(0, 4)))
    if active != 0:
        r = (r > 3).astype(np.float32)
    extra = []
    if active in (1, 2):
        extra.append(("base_score", "0.5"))
    if rng.integers(0, 5) == 0:
        extra.append(("no_user_bias", "1"))
    reg = int(rng.choice([0, 0, 0, 1, 3]))
    if reg:
        extra.append(("reg_method", str(reg)))
    if rng.integers(0, 4) == 0:
        extra += [("wd_item", "0.02"), ("wd_item_bias", "0.001")]
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=k, learning_rate="0.002") + extra
    t = sa.Trainer(0, active)
    o = oracle.OracleTrainer("port", 0, active)
    for x in (t, o):   # (one after the other: both draw their model from the process's libc rand() stream)
        x.seed(10)
        for kk, v in conf:
            x.set_param(kk, str(v))
        if x is t:
            t.set_param("amd:step", "minibatch")
        x.init_model()
        x.init_trainer()
    for kk, v in (("window_hot_sub", sub), ("window_hot_max", cap), ("window_per_target", int(rng.choice([24, 100000])))):
        t.set_knob(kk, v)
    ds = t.dataset_from_triples(u, i, r)
    W = ds.num_batches
    ws = [sa.CSRData.from_triples(u[n * w // W:n * (w + 1) // W], i[n * w // W:n * (w + 1) // W], r[n * w // W:n * (w + 1) // W]) for w in range(W)]
    for _ in range(passes):
        t.train_dataset(ds)
        for d in ws:
            o.update_window_substeps(d, sub)
    t.synchronize()
    ok = True
    for name in NAMES:
        a, b = t.view(name), o.view(name)
        if a is None and b is None:
            continue
        if not np.array_equal(a.view(np.uint32), b.view(np.uint32)):
            print("MISMATCH case %d %s: nu %d ni %d n %d k %d sub %d cap %d W %d active %d extra %s" % (case, name, nu, ni, n, k, sub, cap, W, active, extra), flush=True)
            ok = False
    ds.close()
    t.close()
    return ok


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    oracle.build()
    rng = np.random.default_rng(a.seed)
    good = sum(1 for c in range(a.iters) if one(rng, c))
    print(json.dumps({"iters": a.iters, "exact": good, "failed": a.iters - good}))


if __name__ == "__main__":
    main()
