#!/usr/bin/env python3
"""Accuracy contract of the N > 1 steps at the FULL BASELINE configs[2] size over data seeds: for every seed the
exact sequential run, the all-reduce window-minibatch step (its result does not depend on the number of ranks: one rank plays them all,
DESIGN.md 6a) and the stratified schedule at 2 / 4 / 8 ranks (N trainers of ONE process play the ranks, hand-overs are device copies:
bit-identical to N processes, tests/test_gpu_window.py / test_window_minibatch.py), held-out RMSE after 3 and after 10 passes.
usage: contract_seeds.py SEEDS [ranks] [--ratings N --users U --items I --chunks C --blocks-per-rank P --windows W]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("seeds")
    ap.add_argument("ranks", nargs="?", default="2,4,8")
    ap.add_argument("--ratings", type=int, default=100_000_000)
    ap.add_argument("--users", type=int, default=1_000_000)
    ap.add_argument("--items", type=int, default=100_000)
    ap.add_argument("--factor", type=int, default=64)
    ap.add_argument("--chunks", type=int, default=4)
    ap.add_argument("--blocks-per-rank", type=int, default=2)
    ap.add_argument("--per-item", type=float, default=32.0)
    ap.add_argument("--checks", default="3,10", help="passes after which the held-out RMSE is taken")
    ap.add_argument("--skip-allreduce", action="store_true")
    ap.add_argument("--zipf", type=float, default=0.0, help="> 0: items Zipf-distributed with this exponent (benchlib/orders.py: the skew of demo/basicMF/ua.base is ~0.7) instead of uniform")
    ap.add_argument("--per-item-max", type=float, default=128.0, help="all-reduce step: the most updates ANY item may meet per window (svdf_wunit.cpp: window_per_target_max)")
    ap.add_argument("--contrib", choices=["fp32", "bf16"], default="fp32", help="amd:contrib of the window-minibatch trainers")
    a = ap.parse_args()
    import torch
    import svdfeature_amd as sa
    from svdfeature_amd.multi_gpu import HipShard, ShardedTrainer, shard_windows, stratified_plan_all_ranks
    dev = torch.device("cuda", 0)
    checks = [int(x) for x in a.checks.split(",")]
    n = a.ratings
    args = argparse.Namespace(users=a.users, items=a.items, factor=a.factor)

    def trainer(window=True):
        t = sa.Trainer(0, 0)
        t.seed(10)
        for k, v in bench.conf_for(args) + ([("amd:contrib", a.contrib)] if (window and a.contrib != "fp32") else []):
            t.set_param(k, v)
        t.init_model()
        t.init_trainer()
        return t

    def out(**kw):
        print(json.dumps(kw), flush=True)
    for seed in [int(x) for x in a.seeds.split(",")]:
        t0 = time.time()
        if a.zipf > 0:
            import types
            from benchlib import orders
            orders.ZIPF_EXPONENT = a.zipf
            u, i, r = orders.synth_zipf_triples(types.SimpleNamespace(Planted=bench.Planted), n + 1_000_000, a.users, a.items, 4321 + seed)
        else:
            u, i, r = bench.synth_triples(n + 1_000_000, a.users, a.items, 12345 + seed)
        tu, ti, tl = u[n:n + 200000], i[n:n + 200000], r[n:n + 200000]
        u, i, r = u[:n], i[:n], r[:n]
        test = sa.CSRData.from_triples(tu, ti, tl)
        print("[contract] seed %d: data in %.0fs" % (seed, time.time() - t0), file=sys.stderr, flush=True)
        # ---- exact sequential SGD (the reference's result)
        sq = trainer(False)
        dsq = sq.dataset_from_triples(u, i, r)
        seq = {}
        for p in range(1, max(checks) + 1):
            sq.train_dataset(dsq)
            if p in checks:
                seq[p] = bench.rmse(sq.predict_batch(test), tl)
        dsq.close(); sq.close()
        out(seed=seed, scheme="sequential", rmse={str(k): v for k, v in seq.items()})
        # ---- all-reduce window-minibatch step: one rank plays all of them
        if not a.skip_allreduce:
            cnt_i = np.bincount(i, minlength=a.items).astype(np.float64)
            nwin = max(1, int(np.ceil(max(float((cnt_i * cnt_i).sum() / max(cnt_i.sum(), 1.0)) / a.per_item, float(cnt_i.max()) / a.per_item_max))))
            t = trainer()
            ad = HipShard(t, torch, dev, minibatch=True)
            ad.set_wire_half(True)
            st = ShardedTrainer(ad, ad.make_windows(shard_windows(u, i, r, 0, 1, nwin)), 1, None, half_delta=True)
            d = {}
            for p in range(1, max(checks) + 1):
                st.train_pass()
                if p in checks:
                    d[p] = bench.rmse(t.predict_batch(test), tl) - seq[p]
            out(seed=seed, scheme="allreduce_minibatch", windows=nwin, wire="fp16", contrib=a.contrib, d={str(k): v for k, v in d.items()})
            for w in st.windows:
                w.close()
            t.close()
        # ---- stratified schedule at N ranks
        for world in [int(x) for x in a.ranks.split(",")]:
            P = a.blocks_per_rank
            B = world * P
            t0 = time.time()
            plans = stratified_plan_all_ranks(u, i, r, world, a.chunks, a.items, a.per_item, P)
            ranks = []
            for rk in range(world):
                ad = HipShard(trainer(), torch, dev, minibatch=True)
                ad.set_wire_half(False)
                ranks.append((ad, [[ad.make_windows(sub) for sub in chunk] for chunk in plans[rk]]))
            del plans
            print("[contract] seed %d, %d ranks: windows built in %.0fs" % (seed, world, time.time() - t0), file=sys.stderr, flush=True)
            d = {}
            for p in range(1, max(checks) + 1):
                for c in range(a.chunks):
                    for s in range(B):
                        for rk, (ad, plan) in enumerate(ranks):
                            b = (rk * P + s) % B
                            for w in plan[c][s]:
                                ad.train(w)
                                ad.apply_local(w, b, B)
                        if world > 1:   # the block trained by rank rk + 1 in this step reaches rank rk before its step s + P
                            outs = []
                            for rk, (ad, _) in enumerate(ranks):
                                blk = ad.block_get((rk * P + s) % B, B)
                                ad.stream.synchronize()
                                outs.append(blk.clone())
                            torch.cuda.synchronize()
                            for rk, (ad, _) in enumerate(ranks):
                                ad.block_set(((rk + 1) * P + s) % B, B, outs[(rk + 1) % world])
                if p in checks:
                    for b in range(B):   # gather_blocks: the item side complete everywhere before scoring
                        own = b // P
                        blk = ranks[own][0].block_get(b, B)
                        ranks[own][0].stream.synchronize()
                        blk = blk.clone()
                        for rk, (ad, _) in enumerate(ranks):
                            if rk != own:
                                ad.block_set(b, B, blk)
                    sse, cnt = 0.0, 0
                    for rk, (ad, _) in enumerate(ranks):
                        m = (tu % world) == rk
                        pr = ad.t.predict_batch(sa.CSRData.from_triples(tu[m], ti[m], tl[m]))
                        sse += float(np.sum((pr.astype(np.float64) - tl[m].astype(np.float64)) ** 2)); cnt += int(m.sum())
                    d[p] = float(np.sqrt(sse / cnt)) - seq[p]
            out(seed=seed, scheme="stratified", ranks=world, chunks=a.chunks, per_item=a.per_item, blocks_per_rank=P, contrib=a.contrib, d={str(k): v for k, v in d.items()})
            for ad, plan in ranks:
                for chunk in plan:
                    for sub in chunk:
                        for w in sub:
                            w.close()
                ad.t.close()


if __name__ == "__main__":
    main()
