"""Randomised differential runs of the device-side builders against the host forms of the same engine:
  init:   SVDModel::rand_init on the device (svdf_k_init.hip) vs the host loop -- random shapes, sigmas, seeds, formats, margins; model bits and
          the next libc rand() draws must be equal;
  window: window data sets of ratings / rank pairs regrouped on the device (svdf_k_wbuild.hip) vs the host builder -- random sizes, skews and
          window lengths through the one-GPU window sequence; the trained models must be equal bit for bit.
usage: python tests/fuzz_builders.py [--iters N] [--seed S]"""
import argparse
import ctypes
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import cases
import svdfeature_amd as sa

libc = ctypes.CDLL(None)
libc.rand.restype = ctypes.c_int


def init_case(rng):
    fmt = int(rng.integers(0, 2))
    kw = dict(num_user=int(rng.integers(0, 3000)), num_item=int(rng.integers(1, 2000)), num_global=int(rng.integers(0, 4)),
              num_factor=int(rng.choice([1, 2, 3, 5, 8, 16, 31, 64, 100, 128, 200, 256])))
    if fmt == 1:
        kw["num_ufeedback"] = int(rng.integers(1, 1500))
        if rng.random() < 0.5:
            kw["ufeedback_init_sigma"] = "%g" % rng.choice([0.0, 0.001, 0.05])
    if rng.random() < 0.4:
        kw["u_init_sigma"] = "%g" % rng.choice([0.0, 0.003, 0.1, 1.0])
    if rng.random() < 0.4:
        kw["i_init_sigma"] = "%g" % rng.choice([0.0, 0.02, 0.5])
    if rng.random() < 0.2:
        kw["user_nonnegative"] = 1
    if rng.random() < 0.2:
        kw["item_nonnegative"] = 1
    if rng.random() < 0.2 and kw["num_user"] > 1:
        kw["num_randinit_ufactor"] = int(rng.integers(1, kw["num_user"] + 1))
    if rng.random() < 0.2 and kw["num_item"] > 1:
        kw["num_randinit_ifactor"] = int(rng.integers(1, kw["num_item"] + 1))
    seed = int(rng.integers(0, 2 ** 31 - 1))
    margin = int(rng.choice([46, 46, 34, 27, 12]))   # 12: wider than the float spacing -> everything reported -> the host loop takes over
    skip = int(rng.integers(0, 100))   # the generator is somewhere in its stream, not right behind srand
    out = []
    for dev in (0, 1):
        t = sa.Trainer(fmt, 0)
        t.set_knob("device_init", dev)
        t.set_knob("device_init_margin_log2", margin)
        t.seed(seed)
        for _ in range(skip):
            libc.rand()
        for k, v in kw.items():
            t.set_param(k, str(v))
        t.init_model()
        nxt = [libc.rand() for _ in range(3)]
        views = {n: t.view(n) for n in ("W_user", "W_item", "W_ufeedback")}
        out.append((views, nxt, t.counter(14)))
    ok = out[0][1] == out[1][1]
    for n in out[0][0]:
        a, b = out[0][0][n], out[1][0][n]
        ok = ok and ((a is None and b is None) or (a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))))
    return ok, dict(kind="init", fmt=fmt, seed=seed, margin=margin, skip=skip, **kw)


def window_case(rng):
    pairs = bool(rng.integers(0, 2))
    nu, ni = int(rng.integers(1, 4000)), int(rng.integers(2, 1500))
    n = int(rng.integers(1, 40000))
    k = int(rng.choice([16, 64, 128])) if not pairs else int(rng.choice([64, 128]))
    window = int(rng.integers(1, n + 1)) if rng.random() < 0.7 else n
    if n // window > 300:
        window = n // 300 + 1
    u = np.where(rng.random(n) < rng.random(), rng.integers(0, max(nu // 20, 1), n), rng.integers(0, nu, n)).astype(np.uint32)
    i = (rng.zipf(1.2 + rng.random(), n) % ni).astype(np.uint32) if rng.random() < 0.5 else rng.integers(0, ni, n).astype(np.uint32)
    if pairs:
        cols = (u, i, ((i + 1 + rng.integers(0, ni - 1, n)) % ni).astype(np.uint32))
    else:
        cols = (u, i, rng.integers(1, 6, n).astype(np.float32))
    conf = cases.conf_with(cases.PAIR_CONF if pairs else cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=k)
    res = []
    for dev in (0, 1):
        t = sa.Trainer(0, 3 if pairs else 0)
        t.seed(10)
        for kk, v in conf + [("amd:step", "minibatch"), ("amd:window", str(window))]:
            t.set_param(kk, v)
        t.init_model()
        t.init_trainer()
        t.set_knob("device_window", dev)
        ds = t.dataset_from_pairs(*cols) if pairs else t.dataset_from_triples(*cols)
        t.train_dataset(ds)
        res.append({x: t.view(x) for x in ("W_user", "W_item", "i_bias")})
    ok = all(np.array_equal(res[0][x].view(np.uint32), res[1][x].view(np.uint32)) for x in res[0])
    return ok, dict(kind="window", pairs=pairs, nu=nu, ni=ni, n=n, k=k, window=window)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    bad, counts = [], {"init": 0, "window": 0}
    for it in range(a.iters):
        ok, desc = (init_case if it % 2 == 0 else window_case)(rng)
        counts[desc["kind"]] += 1
        if not ok:
            bad.append(desc)
            print("MISMATCH", json.dumps(desc), flush=True)
    print(json.dumps({"iters": a.iters, "seed": a.seed, "cases": counts, "failed": len(bad)}))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
