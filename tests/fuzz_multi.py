#!/usr/bin/env python3
"""(not collected by pytest) Randomised differential run of the amd:gpus handle (svdf_multi.cpp) on virtual ranks: random rank counts, window
counts, widths, links, regularisers, both window steps, staged update() calls or resident data sets, (user, item, rating) rows or rank
pairs -- against the oracle-backed simulation of the same algorithm (tests/multi_rank_utils.simulate), bit for bit with fp32 deltas.
usage: python tests/fuzz_multi.py --iters 300 --seed 1"""
import argparse, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cases
import svdfeature_amd as sa
from multi_rank_utils import simulate
from svdfeature_amd.multi_gpu import Pairs


def one(rng):
    world = int(rng.integers(2, 9))
    windows = int(rng.integers(1, 7))
    k = int(rng.choice([4, 10, 16, 33, 64, 100, 128]))
    nu, ni = int(rng.integers(world * 4, 900)), int(rng.integers(8, 260))
    n = windows * int(rng.integers(20, 2500))
    passes = int(rng.integers(1, 3))
    pairs = bool(rng.integers(0, 3) == 0)
    step = str(rng.choice(["minibatch", "levels"]))
    resident = bool(rng.integers(0, 2))
    extra = {}
    if pairs:
        active = 3
        base = cases.PAIR_CONF
        extra.update(learning_rate=0.05, ui_init_sigma=0.1)
        if resident:
            step = "minibatch"   # the handle's pair entry point is the window-minibatch step; staged pair rows take the level scheme
    else:
        active = int(rng.choice([0, 0, 2]))
        base = cases.BASICMF_CONF
        if active == 2:
            extra.update(base_score=0.5)
        if rng.integers(0, 3) == 0:
            extra.update(reg_method=1)
        if rng.integers(0, 4) == 0:
            extra.update(no_user_bias=1)
    conf = cases.conf_with(base, num_user=nu, num_item=ni, num_factor=k, **extra)
    seed = int(rng.integers(0, 1 << 30))
    if pairs:
        u, p, q = cases.planted_pairs(n, nu, ni, seed=seed)
        data = Pairs(u, p, q)
        csr = sa.pairs_as_csr(u, p, q)
    else:
        u, i, r = cases.planted_triples(n, nu, ni, seed=seed)
        if active == 2:
            r = (r > 3).astype(np.float32)
        csr = sa.CSRData.from_triples(u, i, r)
    t = sa.Trainer(0, active)
    t.seed(10)
    for kk, v in list(conf) + [("amd:gpus", world), ("amd:delta_half", 0), ("amd:window", n // windows), ("amd:step", step)]:
        t.set_param(kk, str(v))
    t.init_model()
    t.init_trainer()
    ds = None
    if resident:
        ds = t.dataset_from_pairs(u, p, q) if pairs else t.dataset_from_triples(u, i, r)
    for _ in range(passes):
        if resident:
            t.train_dataset(ds)
        else:
            t.update_batch(csr)
        t.finish_round()
    # which step the handle really took: staged pair rows are not plain triples -> levels
    minibatch = step == "minibatch" and (resident or not pairs)
    sim = simulate(conf, data if pairs else u, None if pairs else i, None if pairs else r, world, windows, passes, active=active, minibatch=minibatch)
    ok = t.counter(8) == passes * windows
    names = ["W_item", "i_bias"]
    for name in names:
        ok = ok and np.array_equal(t.view(name).view(np.uint32), sim[0].t.view(name).view(np.uint32))
    wu = t.view("W_user")
    bu = t.view("u_bias")
    for rk in range(world):
        own = (np.arange(nu) % world) == rk
        ok = ok and np.array_equal(wu[own].view(np.uint32), sim[rk].t.view("W_user")[own].view(np.uint32))
        if bu is not None and bu.size:
            ok = ok and np.array_equal(bu[own].view(np.uint32), sim[rk].t.view("u_bias")[own].view(np.uint32))
    if ds is not None:
        ds.close()
    t.close()
    return bool(ok), dict(world=world, windows=windows, k=k, nu=nu, ni=ni, n=n, passes=passes, pairs=pairs, step=step, resident=resident, active=active, extra=extra, seed=seed)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    exact = failed = 0
    for it in range(a.iters):
        ok, info = one(rng)
        if ok:
            exact += 1
        else:
            failed += 1
            print("MISMATCH", json.dumps(info), flush=True)
    print(json.dumps({"iters": a.iters, "exact": exact, "failed": failed}))
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
