#!/usr/bin/env python3
"""(not collected by pytest) Randomised differential run of the window-minibatch step for user units (svdf_k_wunit.hip, svdf_wunit.cpp): random
user-group block streams (DEFAULT and split blocks, users without feedback, repeated users, global entries, two item entries) or rows with global
features, 1 ... 4 simulated ranks, 1 ... 5 windows, widths incl. 64 / 128 (slot kernel), links, regularisers, fp32 / bf16 contribution rows, the
lane-group kernel forced or not -- through HipShard(minibatch=True) with an explicit sum in rank order, against the oracle-backed simulation
(tests/multi_rank_utils.simulate), bit for bit.  usage: python tests/fuzz_wunit.py --iters 300 --seed 1"""
import argparse, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cases
import multi_rank_utils
import svdfeature_amd as sa
from multi_rank_utils import simulate
from svdfeature_amd import BlockArrays, CSRData
from svdfeature_amd.data import PlusBlock
from svdfeature_amd.multi_gpu import HipShard, shard_block_windows, shard_csr_windows


def rows_of(rng, n, users, ni, ng, max_g, two_items, uvals):
    rows = []
    for _ in range(n):
        g = sorted(int(x) for x in rng.choice(ng, size=int(rng.integers(0, max_g + 1)), replace=False)) if ng else []
        it = [(int(rng.integers(0, ni)), float(rng.choice([1.0, 1.0, 0.5, -1.0])) if two_items else 1.0)]
        if two_items and rng.random() < 0.3:
            x = int(rng.integers(0, ni))
            if x != it[0][0]:
                it = sorted(it + [(x, -0.5)])
        rows.append((float(rng.integers(1, 6)), [(x, float(rng.uniform(0.1, 1.0))) for x in g],
                     [(int(rng.choice(users)), float(rng.choice([1.0, 0.5])) if uvals else 1.0)], it))
    return rows


def one(rng, torch, wave=False, onegpu=False):
    """wave: the shapes k_wunit_wave takes (feedback blocks, fixed row layout without global entries, k = 64 NR) with long units -- up to 150
    rows and 140 feedback ids: several 64-record blocks, partial groups, partial batches"""
    world, windows, passes = int(rng.integers(1, 5)), int(rng.integers(1, 6)), int(rng.integers(1, 3))
    if onegpu:
        world = 1
    k = int(rng.choice([64, 128, 128, 192, 256])) if wave else int(rng.choice([4, 10, 16, 33, 64, 64, 100, 128, 128, 200]))
    nu, ni = int(rng.integers(world * 4, 500)), int(rng.integers(8, 200))
    if onegpu and rng.random() < 0.6:
        ni = int(rng.integers(500, 5000))   # many more items than rows per window: most contributions are the only one of their row (applied in place)
    ng = 0 if wave else int(rng.choice([0, 0, 6, 30]))
    blocks_mode = True if wave else bool(rng.integers(0, 2))
    fixed = True if wave else bool(rng.integers(0, 2))          # fixed row layout (slot kernel at k = 64 / 128) or ragged rows
    active = int(rng.choice([0, 0, 0, 2]))
    extra = {}
    r = int(rng.integers(0, 8))
    if r == 0: extra.update(reg_method=1)
    elif r == 1: extra.update(reg_method=2, wd_user=0.5, wd_item=0.5)
    elif r == 2: extra.update(no_user_bias=1)
    elif r == 3: extra.update(user_nonnegative=1)
    elif r == 4 and ng: extra.update(reg_global=1, num_regfree_global=2)
    elif r == 5: extra.update(reg_method=3)
    if active == 2: extra.update(base_score=0.5)
    bf16 = bool(rng.integers(0, 3) == 0)
    knobs = [("wunit_fast", int(rng.integers(0, 2)))] if (rng.integers(0, 3) == 0 and not wave) else []   # default 2: one wave per unit where it applies
    if blocks_mode:
        nb = int(rng.integers(windows, 60 if wave else 260))
        long_units = wave and rng.random() < 0.6
        if wave:
            ni = max(ni, 160)
        blocks = []
        users = rng.integers(0, nu, nb)
        for b in range(nb):
            uid = int(users[b])
            nrow = int(rng.integers(1, 150 if long_units else 14))
            nfb = 0 if rng.random() < 0.15 else int(rng.integers(1, min(ni, 140 if long_units else 12) + 1))
            fb_idx = np.sort(rng.choice(ni, size=nfb, replace=False)).astype(np.uint32)
            fb_val = np.full(nfb, 1.0 / np.sqrt(max(nfb, 1)), np.float32) if rng.random() < 0.7 else rng.uniform(0.1, 1.0, nfb).astype(np.float32)
            rows = rows_of(rng, nrow, [uid], ni, 0 if fixed else ng, 3, not fixed, not fixed)
            if fixed and ng:   # fixed layout with global entries on a user-group trainer (general kernel)
                rows = [(l, [(int(x), 0.5) for x in sorted(rng.choice(ng, size=2, replace=False))], u, i) for (l, _, u, i) in rows]
            d = CSRData.from_rows(rows)
            if active == 2:
                d.row_label[:] = (d.row_label > 3).astype(np.float32)
            if nrow >= 3 and rng.random() < 0.25:
                e = np.zeros(0, np.uint32), np.zeros(0, np.float32)
                blocks += [PlusBlock(fb_idx, fb_val, d.slice_rows(0, 1), 1), PlusBlock(e[0], e[1], d.slice_rows(1, nrow - 1), 3), PlusBlock(fb_idx, fb_val, d.slice_rows(nrow - 1, nrow), 2)]
            else:
                blocks.append(PlusBlock(fb_idx, fb_val, d, 0))
        data = BlockArrays.from_blocks(blocks)
        fmt = 1
        conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=k, num_global=ng, num_ufeedback=ni, wd_ufeedback=0.004, ufeedback_init_sigma=0.01,
                               wd_global=0.002, scale_lr_ufeedback=float(rng.choice([1.0, 0.5])), **extra)
        names = ("W_item", "i_bias", "W_ufeedback", "ufeedback_bias", "W_user", "u_bias") + (("g_bias",) if ng else ())
    else:
        n = windows * int(rng.integers(10, 1500))
        mg = 4 if ng else 0
        if fixed and ng:
            rows = []
            for _ in range(n):
                g = sorted(int(x) for x in rng.choice(ng, size=4, replace=False))
                rows.append((float(rng.integers(1, 6)), [(x, float(rng.uniform(0.1, 1.0))) for x in g], [(int(rng.integers(0, nu)), 1.0)], [(int(rng.integers(0, ni)), 1.0)]))
        else:
            rows = rows_of(rng, n, np.arange(nu), ni, ng, mg, not fixed, not fixed)
        data = CSRData.from_rows(rows)
        if active == 2:
            data.row_label[:] = (data.row_label > 3).astype(np.float32)
        fmt = 0
        conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=k, num_global=ng, wd_global=0.002, **extra)
        names = ("W_item", "i_bias", "W_user", "u_bias") + (("g_bias",) if ng else ())
    if onegpu:   # `amd:step = minibatch` on one handle: a window sequence (kind 8) against the one-rank simulation with the same cuts
        nrows = data.num_row
        t = sa.Trainer(fmt, active)
        t.seed(10)
        for kk, v in conf + [("amd:step", "minibatch"), ("amd:window", max(1, -(-nrows // windows)))] + ([("amd:contrib", "bf16")] if bf16 else []):
            t.set_param(kk, str(v))
        t.init_model()
        t.init_trainer()
        inplace = int(rng.integers(0, 4) != 0)
        for kk, v in knobs + [("wunit_inplace", inplace), ("wunit_defer_fb", int(rng.integers(0, 3) != 0))]:
            t.set_knob(kk, v)
        ds = t.dataset_from_blocks(data) if blocks_mode else t.dataset_from_csr(data)
        for _ in range(passes):
            t.train_dataset(ds)
        t.synchronize()
        multi_rank_utils.CONTRIB_BF16 = bf16
        try:
            sim = simulate(conf, data, None, None, 1, ds.num_batches, passes, fmt=fmt, active=active, minibatch=True)
        finally:
            multi_rank_utils.CONTRIB_BF16 = False
        ok = True
        for name in names:
            a, b = t.view(name), sim[0].t.view(name)
            both_nan = np.isnan(a) & np.isnan(b)
            if not np.array_equal(np.where(both_nan, 0, a.view(np.uint32)), np.where(both_nan, 0, b.view(np.uint32))):
                ok = False
                if os.environ.get("FUZZ_WUNIT_DEBUG"):
                    bad = np.argwhere(a.view(np.uint32) != b.view(np.uint32))
                    print("  differs:", name, "elements", len(bad), "of", a.size, "first", bad[:6].tolist(), flush=True)
        desc = dict(onegpu=True, windows=ds.num_batches, passes=passes, k=k, nu=nu, ni=ni, ng=ng, blocks=blocks_mode, fixed=fixed, active=active, extra=extra, bf16=bf16, knobs=knobs, inplace=inplace)
        ds.close()
        t.close()
        return ok, desc
    dev = torch.device("cuda", 0)
    defer_ranks = int(rng.integers(0, 3) != 0)   # feedback contributions formed by the sum kernel (default) or written as rows by the walk
    ranks = []
    for rk in range(world):
        t = sa.Trainer(fmt, active)
        t.seed(10)
        for kk, v in conf + ([("amd:contrib", "bf16")] if bf16 else []):
            t.set_param(kk, str(v))
        t.init_model()
        t.init_trainer()
        for kk, v in knobs + [("wunit_defer_fb", defer_ranks)]:
            t.set_knob(kk, v)
        ad = HipShard(t, torch, dev, minibatch=True)
        ad.set_wire_half(False)
        sh = shard_block_windows(data, rk, world, windows) if blocks_mode else shard_csr_windows(data, rk, world, windows)
        ranks.append((ad, ad.make_windows(sh)))
    for _ in range(passes):
        for w in range(windows):
            ds_ = []
            for ad, wins in ranks:
                ad.train(wins[w])
                d = ad.delta_get()
                ad.stream.synchronize()
                ds_.append(d.clone())
            total = ds_[0]
            for d in ds_[1:]:
                total = total + d
            torch.cuda.synchronize()
            for ad, _ in ranks:
                ad.delta_set(total)
    for ad, _ in ranks:
        ad.t.synchronize()
    multi_rank_utils.CONTRIB_BF16 = bf16
    try:
        sim = simulate(conf, data, None, None, world, windows, passes, fmt=fmt, active=active, minibatch=True)
    finally:
        multi_rank_utils.CONTRIB_BF16 = False
    ok = True
    for (ad, _), s_ in zip(ranks, sim):
        for name in names:
            a, b = ad.t.view(name), s_.t.view(name)
            both_nan = np.isnan(a) & np.isnan(b)   # a diverged run (few global ids, large windows): NaN payload bits are the FPU's, not the algorithm's
            if not np.array_equal(np.where(both_nan, 0, a.view(np.uint32)), np.where(both_nan, 0, b.view(np.uint32))):
                ok = False
                if os.environ.get("FUZZ_WUNIT_DEBUG"):
                    bad = np.argwhere(a.view(np.uint32) != b.view(np.uint32))
                    print("  differs:", name, "elements", len(bad), "of", a.size, "first", bad[:6].tolist(), "gpu", a[tuple(bad[0])], "oracle", b[tuple(bad[0])],
                          "max abs", float(np.abs(a - b).max()), flush=True)
    desc = dict(world=world, windows=windows, passes=passes, k=k, nu=nu, ni=ni, ng=ng, blocks=blocks_mode, fixed=fixed, active=active, extra=extra, bf16=bf16, knobs=knobs)
    for ad, wins in ranks:
        for w in wins:
            w.close()
        ad.t.close()
    return ok, desc


def main():
    import torch
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--wave", action="store_true", help="only the shapes the one-wave-per-unit kernel takes, long units")
    ap.add_argument("--one-gpu", action="store_true", help="amd:step = minibatch window sequences on one handle (single contributions applied in place)")
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    tot = dict(iters=0, exact=0, failed=0)
    for it in range(a.iters):
        ok, desc = one(rng, torch, a.wave, a.one_gpu)
        tot["iters"] += 1
        if ok:
            tot["exact"] += 1
        else:
            tot["failed"] += 1
            print("MISMATCH", json.dumps(desc, default=str), flush=True)
            if os.environ.get("FUZZ_WUNIT_DEBUG"):
                break
    print(json.dumps(tot))


if __name__ == "__main__":
    main()
