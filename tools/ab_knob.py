"""A/B of one tuning knob inside ONE process on the contract workload (100 M ratings, k=64): alternates the knob's values between
groups of passes over the same resident data set, so box-to-box and run-to-run variation cancels.  Timing experiment only.
usage: python tools/ab_knob.py load_mode=0 load_mode=1 load_mode=1,store_mode=1 [--pairwise]   (each argument = one setting)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import svdfeature_amd as sa  # noqa: E402

if os.environ.get("AB_LIB"):   # another build of the engine (e.g. the previous commit's), for A/B across builds on one box
    sa.LIB_PATH = os.environ["AB_LIB"]


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    pairwise = "--pairwise" in sys.argv
    svdpp = "--svdpp" in sys.argv
    neigh = "--neighbourhood" in sys.argv
    factor = [int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--factor=")]
    values = args
    nu, ni = 1_000_000, 100_000
    if neigh:
        d_all = bench.synth_neighbourhood(4_000_000, nu, ni, 10000, 4)
        t = sa.Trainer(0, 0)
        conf = [("learning_rate", "0.005"), ("wd_item", "0.004"), ("wd_user", "0.004"), ("num_item", str(ni)), ("num_user", str(nu)),
                ("num_factor", "128"), ("base_score", "3"), ("num_global", "10000"), ("wd_global", "0.001")]
    elif svdpp:
        train, _ = bench.synth_user_blocks(40000, 100, nu, ni)
        t = sa.Trainer(1, 0)
        conf = [("learning_rate", "0.005"), ("wd_item", "0.004"), ("wd_user", "0.004"), ("num_item", str(ni)), ("num_user", str(nu)),
                ("num_factor", "128"), ("base_score", "3"), ("num_global", "0"), ("num_ufeedback", str(ni)), ("wd_ufeedback", "0.004")]
    elif pairwise:
        n, k = 50_000_000, 128
        u, p, q = bench.synth_pairs(n, nu, ni)
        t = sa.Trainer(0, 3)
        conf = [("learning_rate", "0.005"), ("wd_item", "0.004"), ("wd_user", "0.004"), ("num_item", str(ni)), ("num_user", str(nu)),
                ("num_factor", str(k)), ("num_global", "0"), ("no_user_bias", "1")]
    else:
        n, k = 100_000_000, (factor[0] if factor else 64)
        if k > 64:
            n = n * 64 // k
        u, i, r = bench.synth_triples(n, nu, ni)
        t = sa.Trainer(0, 0)
        conf = [("base_score", "3"), ("learning_rate", "0.005"), ("wd_item", "0.004"), ("wd_user", "0.004"), ("num_item", str(ni)),
                ("num_user", str(nu)), ("num_global", "0"), ("num_factor", str(k)), ("active_type", "0")]
    t.seed(10)
    for kk, v in conf:
        t.set_param(kk, v)
    t.init_model()
    t.init_trainer()
    for a in sys.argv[1:]:   # --pre=use_fused=0: a knob set once BEFORE the data sets are built (routes them to other kernels)
        if a.startswith("--pre="):
            t.set_knob(a[6:].split("=")[0], int(a[6:].split("=")[1]))
    # knobs that shape the data set (batch order) need a data set of their own per value: "sort_batches=2" inside a setting
    dsets = {}
    for v in values:
        sb = [kv for kv in v.split(",") if kv.startswith("sort_batches=")]
        key = sb[0] if sb else ""
        if key not in dsets:
            if sb:
                t.set_knob("sort_batches", int(sb[0].split("=")[1]))
            dsets[key] = t.dataset_from_csr(d_all) if neigh else t.dataset_from_blocks(train) if svdpp else (t.dataset_from_pairs(u, p, q) if pairwise else t.dataset_from_triples(u, i, r))
    for ds in dsets.values():
        t.train_dataset(ds)
    t.synchronize()
    res = {v: [] for v in values}
    for rep in range(6):
        for v in values:
            sb = [kv for kv in v.split(",") if kv.startswith("sort_batches=")]
            ds = dsets[sb[0] if sb else ""]
            for kv in v.split(","):
                if not kv.startswith("sort_batches="):
                    t.set_knob(kv.split("=")[0], int(kv.split("=")[1]))
            t.train_dataset(ds)
            t.synchronize()
            t0 = time.time()
            for _ in range(3):
                t.train_dataset(ds)
            t.synchronize()
            res[v].append((time.time() - t0) / 3 * 1e3)
    for v in values:
        a = np.array(res[v])
        print("%-40s ms/pass median %.3f  min %.3f  max %.3f   %s" % (v, np.median(a), a.min(), a.max(), np.round(a, 2).tolist()))


if __name__ == "__main__":
    main()
