"""(not collected by pytest) Randomised differential run of the user-run units of rank pairs (svdf_punit.cpp, k_pair_units) against the level-by-level pass
of the same engine (knob pair_units = 0: k_fewrow_slots / k_fused, itself bit-exact against the oracle): random users / items / pairs per user, widths
1 .. 256, links, decays, user bias on / off, unit caps 1 .. 64, partially grouped streams; every parameter and every prediction compared bit for bit.
usage: python tests/fuzz_punit.py [--iters N] [--seed S]"""
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

NAMES = ("W_user", "W_item", "i_bias", "u_bias")


def one(rng, case):
    nu, ni = int(rng.integers(5, 800)), int(rng.integers(4, 1500))
    per = int(rng.integers(2, 120))
    users = rng.permutation(nu).astype(np.uint32)
    u = np.repeat(users, per)
    if rng.integers(0, 3) == 0:   # some blocks broken up: runs of different lengths, users coming back
        cut = rng.integers(0, len(u), max(1, len(u) // 50))
        u[cut] = rng.integers(0, nu, len(cut)).astype(np.uint32)
    n = len(u)
    p = rng.integers(0, ni, n).astype(np.uint32)
    q = ((p + 1 + rng.integers(0, ni - 1, n)) % ni).astype(np.uint32)
    k = int(rng.choice([1, 3, 8, 16, 24, 32, 64, 64, 100, 128, 128, 129, 200, 256]))
    active = int(rng.choice([3, 3, 3, 0, 2, 5]))
    extra = [("active_type", str(active))]
    if rng.integers(0, 3) == 0:
        extra.append(("no_user_bias", "0"))
    reg = int(rng.choice([0, 0, 0, 1, 2, 3]))
    if reg:
        extra.append(("reg_method", str(reg)))
    if reg == 2:
        extra += [("wd_item", "4.0"), ("wd_user", "4.0")]
    cap = int(rng.choice([1, 2, 5, 16, 16, 24, 64]))
    passes = int(rng.integers(1, 3))
    conf = cases.conf_with(cases.PAIR_CONF, num_user=nu, num_item=ni, num_factor=k) + extra
    res = []
    for units in (0, 1):
        t = sa.Trainer(0, active)
        t.seed(10)
        for kk, v in conf:
            t.set_param(kk, str(v))
        t.init_model()
        t.init_trainer()
        t.set_knob("pair_units", units)
        t.set_knob("pair_unit_cap", cap)
        ds = t.dataset_from_pairs(u, p, q)
        for _ in range(passes):
            t.train_dataset(ds)
        pr = t.predict_dataset(ds)
        res.append((ds.kind, {nm: (None if t.view(nm) is None else t.view(nm).copy()) for nm in NAMES}, pr.copy()))
        ds.close()
        t.close()
    ok = True
    for nm in NAMES:
        a, b = res[0][1][nm], res[1][1][nm]
        if a is not None and not np.array_equal(a.view(np.uint32), b.view(np.uint32)):
            ok = False
    if not np.array_equal(res[0][2].view(np.uint32), res[1][2].view(np.uint32)):
        ok = False
    if not ok:
        print("MISMATCH case %d: nu %d ni %d per %d k %d active %d extra %s cap %d kinds %d/%d" % (case, nu, ni, per, k, active, extra, cap, res[0][0], res[1][0]), flush=True)
    return ok, res[1][0] == 11


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    good = units = 0
    for c in range(a.iters):
        ok, was_units = one(rng, c)
        good += ok
        units += was_units
    print(json.dumps({"iters": a.iters, "exact": good, "failed": a.iters - good, "walked_as_units": units}))


if __name__ == "__main__":
    main()
