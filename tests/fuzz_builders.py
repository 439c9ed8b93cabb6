"""Randomised differential runs of the device-side builders against the host forms of the same engine:
  init:   SVDModel::rand_init on the device (svdf_k_init.hip) vs the host loop -- random shapes, sigmas, seeds, formats, margins; model bits and
          the next libc rand() draws must be equal;
  window: window data sets of ratings / rank pairs regrouped on the device (svdf_k_wbuild.hip) vs the host builder -- random sizes, skews and
          window lengths through the one-GPU window sequence; the trained models must be equal bit for bit;
  units:  level schedules of user-group (SVD++) data sets built on the device (svdf_k_sched.hip: device_schedule_units) vs the host scan --
          random block streams (split users, repeated items / feedback ids, rows with global entries or two item entries, hot items that
          chain the units); the schedule digests (level_ptr, level_mid, order), the unit counts and the trained models must be equal.
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


def units_case(rng):
    nu, ni = int(rng.integers(2, 1500)), int(rng.integers(2, 800))
    nb = int(rng.integers(1, min(nu, 600) + 1))
    k = int(rng.choice([8, 16, 64, 128]))
    seed = int(rng.integers(0, 2 ** 31 - 1))
    blocks = cases.user_blocks(nb, nu, ni, ni, seed=seed, max_rows=int(rng.integers(1, 40)), max_fb=int(rng.integers(1, min(ni, 40) + 1)),
                               split_every=int(rng.choice([0, 0, 3, 7])))
    hot = int(rng.integers(0, ni)) if rng.random() < 0.3 else -1
    for b in blocks:
        r = rng.random()
        n = b.data.num_row
        if hot >= 0 and n >= 1 and rng.random() < 0.5:
            b.data.feat_index[1] = hot                                   # many units meet on one item row: a chain
        if r < 0.10 and n >= 2:
            b.data.feat_index[2 * (n - 1) + 1] = b.data.feat_index[1]    # the same item twice inside the unit (row_fresh)
        elif r < 0.18 and b.num_ufeedback >= 2:
            b.index_ufeedback[b.num_ufeedback - 1] = b.index_ufeedback[0]   # a feedback id listed twice: not simple
        elif r < 0.24 and n >= 1:
            b.data.feat_value[int(rng.integers(0, 2 * n))] = 0.25        # non-unit value
        elif r < 0.28 and n >= 2:
            b.data.feat_index[2] = (int(b.data.feat_index[0]) + 1) % nu   # a second user id inside the unit: not simple
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=k, num_ufeedback=ni, wd_ufeedback=0.004, scale_lr_ufeedback=0.7,
                           ufeedback_init_sigma=0.01, learning_rate=0.01)
    res = []
    for dev in (0, 1):
        t = sa.Trainer(1, 0)
        t.seed(7)
        for kk, v in conf:
            t.set_param(kk, v)
        t.init_model()
        t.init_trainer()
        t.set_knob("device_schedule", dev)
        t.set_knob("device_schedule_min", 1)
        ds = t.dataset_from_blocks(blocks)
        if ds.kind != 3:
            return True, dict(kind="units", skipped="no feedback in the stream", seed=seed)
        facts = [ds.info(w) for w in (0, 1, 2, 5, 6, 7)]
        t.train_dataset(ds)
        res.append((facts, {x: t.view(x) for x in ("W_user", "W_item", "W_ufeedback", "u_bias", "i_bias", "ufeedback_bias")}, t.counter(25)))
    ok = res[0][0] == res[1][0] and res[0][2] == 0 and res[1][2] == 1
    ok = ok and all(np.array_equal(res[0][1][x].view(np.uint32), res[1][1][x].view(np.uint32)) for x in res[0][1])
    return ok, dict(kind="units", nu=nu, ni=ni, nb=nb, k=k, seed=seed, hot=hot, facts_host=res[0][0], facts_device=res[1][0])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    bad, counts = [], {"init": 0, "window": 0, "units": 0}
    for it in range(a.iters):
        ok, desc = (init_case, window_case, units_case)[it % 3](rng)
        counts[desc["kind"]] += 1
        if not ok:
            bad.append(desc)
            print("MISMATCH", json.dumps(desc), flush=True)
    print(json.dumps({"iters": a.iters, "seed": a.seed, "cases": counts, "failed": len(bad)}))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
