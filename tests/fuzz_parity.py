#!/usr/bin/env python3
"""Randomised differential run of the HIP engine against the C oracle (tests/ because it drives the oracle; not collected
by pytest).  Every iteration draws a configuration -- format, link, regulariser, decay modes, per-range decay, factor
width, flags, shared parameter spaces, side tables, staging window / chunking, resident data set or staged calls -- and
seeded data, trains both engines the same way and compares every parameter and the predictions bit for bit (sigmoid links
included).  Prints one JSON line; exits non-zero on the first mismatch with the configuration.

    python tests/fuzz_parity.py --iters 300 --seed 1
"""
import argparse
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import cases  # noqa: E402
import svdfeature_amd as sa  # noqa: E402
from oracle import oracle  # noqa: E402

EXPF = {1, 2, 3, 7}
def same_bits(a, b):
    """bit-identical, except that a NaN matches a NaN: once a run has diverged (large learning rate on side-table data) x86 and
    gfx950 produce NaNs with different sign / payload bits, which is outside the contract"""
    return bool(np.all((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))))


VIEWS = ("W_user", "W_item", "u_bias", "i_bias", "g_bias", "W_ufeedback", "ufeedback_bias")


def draw(rng, tmp, wide=False, big=False):
    fmt = int(rng.integers(0, 2))
    active = int(rng.choice([0, 0, 0, 1, 2, 3, 5, 6, 7]))
    binary = active != 0
    shared = int(rng.integers(0, 6)) == 0
    nu = int(rng.integers(8, 3000 if big else 60))
    ni = nu if shared else int(rng.integers(6, 800 if big else 50))
    ng = int(rng.integers(0, 10))
    k = int(rng.choice([257, 400, 512, 700, 1000, 1024] if wide else [1, 3, 4, 7, 8, 12, 16, 31, 32, 33, 64, 65, 100, 128, 130, 200, 256, 300]))
    reg_method = int(rng.choice([0, 0, 1, 2, 3, 4, 5]))
    reg_global = int(rng.choice([0, 0, 1, 4, 5]))
    conf = dict(num_user=nu, num_item=ni, num_global=ng, num_factor=k, learning_rate=float(rng.choice([0.005, 0.01, 0.05])),
                wd_user=float(rng.choice([0.0, 0.00005, 0.004, 0.02])), wd_item=float(rng.choice([0.0, 0.00002, 0.004, 0.03])),
                wd_global=float(rng.choice([0.0, 0.002, 0.05])), wd_user_bias=float(rng.choice([0.0, 0.001])),
                wd_item_bias=float(rng.choice([0.0, 0.003])), reg_method=reg_method, reg_global=reg_global,
                num_regfree_global=int(rng.integers(0, 3)), no_user_bias=int(rng.integers(0, 4) == 0),
                user_nonnegative=int(rng.integers(0, 5) == 0), decay_learning_rate=int(rng.integers(0, 3) == 0), decay_rate=0.9)
    if binary:
        conf["base_score"] = 0.4
    if shared:
        conf.update(common_latent_space=1, common_feedback_space=1)
    extra = []
    if rng.integers(0, 4) == 0:
        extra += [("up:wd", "0.01"), ("up:bound", str(max(1, nu // 2))), ("up:wd", "0.001"), ("up:bound", str(nu)),
                  ("ip:wd", "0.02"), ("ip:bound", str(ni))]
        if ng:
            extra += [("gp:wd", "0.05"), ("gp:bound", str(ng))]
    side = fmt == 0 and not shared and rng.integers(0, 4) == 0
    if side:
        fu, fi = os.path.join(tmp, "fu.txt"), os.path.join(tmp, "fi.txt")
        cases.write_side_table(fu, max(1, nu - 3), nu, int(rng.integers(0, 1 << 30)))
        cases.write_side_table(fi, ni, ni, int(rng.integers(0, 1 << 30)))
        conf.update(feature_user=fu, feature_item=fi)
    seed = int(rng.integers(0, 1 << 30))
    if fmt == 0:
        shape = int(rng.integers(0, 3))
        if shape == 0:   # basicMF triples
            u, i, r = cases.planted_triples(int(rng.integers(50, 40000 if big else 1500)), nu, ni, seed)
            if binary:
                r = (r > 3).astype(np.float32)
            train = sa.CSRData.from_triples(u, i, r)
        else:
            train = cases.sparse_feature_rows(int(rng.integers(50, 6000 if big else 900)), nu, ni, ng, seed, max_u=2 if shape == 1 else 3,
                                              max_i=2 if shape == 1 else 3, binary_label=binary, allow_dup=shape == 2)
        data = dict(train=train)
    else:
        nfb = nu if shared else ni
        conf.update(num_ufeedback=nfb, wd_ufeedback=0.004, wd_ufeedback_bias=float(rng.choice([0.0, 0.002])),
                    scale_lr_ufeedback=float(rng.choice([1.0, 0.7])), ufeedback_init_sigma=0.01)
        if rng.integers(0, 2) == 0:
            blocks = cases.user_blocks(int(rng.integers(5, min(nu, 600 if big else 40))), nu, ni, nfb, seed, split_every=int(rng.choice([0, 3])),
                                       binary_label=binary)
        else:
            blocks = cases.rank_blocks(int(rng.integers(5, 400 if big else 40)), max(nu, 4), ni, ng if ng >= 2 else 0, seed, graded=not binary, max_fb=int(rng.integers(0, 4)))
            for b in blocks:   # rank_blocks draws feedback ids below num_item
                b.index_ufeedback = b.index_ufeedback % np.uint32(nfb)
                b.index_ufeedback = np.unique(b.index_ufeedback)
                b.value_ufeedback = b.value_ufeedback[:len(b.index_ufeedback)]
        data = dict(train_blocks=blocks)
    plan = dict(rounds=int(rng.integers(1, 4)), chunk=int(rng.choice([0, 0, 7, 64])), window=int(rng.choice([0, 0, 50, 400])),
                resident=bool(rng.integers(0, 2)), knobs={}, single=bool(rng.integers(0, 5) == 0), peek=bool(rng.integers(0, 3) == 0),
                reload=bool(rng.integers(0, 4) == 0))
    if rng.integers(0, 3) == 0:
        plan["knobs"]["use_fused"] = 0
    if rng.integers(0, 3) == 0:
        plan["knobs"]["use_simple_units"] = 0
    if rng.integers(0, 4) == 0:
        plan["knobs"]["rows_without_feedback"] = 0
    # round 2: the device scheduler against the host scheduler, windows of plain instances scheduled on the device
    if rng.integers(0, 2) == 0:
        plan["knobs"]["device_schedule"] = 0
    elif rng.integers(0, 2) == 0:
        plan["knobs"]["device_schedule_min"] = 1
    # round 5: runs of an item's consecutive ratings (svdf_k_runs.hip) and hot rows walked as units (svdf_pivot.cpp) at fuzz sizes -- both are
    # schedule forms of the same pass: they apply to the configurations they are built for and must not change a bit
    if rng.integers(0, 2) == 0:
        plan["knobs"]["runs_min_rows"] = 0
        plan["knobs"]["runs_len"] = int(rng.integers(2, 8))
        if rng.integers(0, 2) == 0:
            plan["knobs"]["runs_sets"] = 2
    elif rng.integers(0, 3) == 0:
        plan["knobs"]["runs_exec"] = 0
    if rng.integers(0, 2) == 0:
        plan["knobs"]["pivot_min"] = int(rng.choice([2, 16, 200]))
        plan["knobs"]["pivot_run"] = int(rng.choice([2, 8, 256]))
    elif rng.integers(0, 3) == 0:
        plan["knobs"]["pivot_exec"] = 0
    if rng.integers(0, 3) == 0:
        plan["knobs"]["chain_width"] = int(rng.choice([0, 4, 128]))
    # the multi-level implicit-feedback solver (extend_type 2) on nested spans, the bilinear solver (15) on plain blocks
    plan["extend"] = 0
    if fmt == 1 and not shared:
        pick = int(rng.integers(0, 5))
        if pick == 0:
            plan["extend"] = 2
            data = dict(train_blocks=cases.nested_blocks(int(rng.integers(5, 60)), nu, ni, nfb, seed))
            if rng.integers(0, 3) == 0:
                extra = extra + [("ufeedback_disable_level", str(int(rng.integers(0, 3))))]
            plan["knobs"].pop("use_simple_units", None)
        elif pick == 1:
            plan["extend"] = 15
            extra = extra + [("num_bi_feedback", str(int(rng.integers(0, 5)))), ("reg_bi_feedback", str(int(rng.integers(0, 6))))]
    # rank pairs through the three-column entry point (random-order trainers, plain data)
    if fmt == 0 and not shared and not side and rng.integers(0, 4) == 0:
        pu, pp, pq = cases.planted_pairs(int(rng.integers(50, 30000 if big else 1500)), nu, max(ni, 2), seed)
        if ni >= 2:
            data = dict(pairs=(pu, pp, pq))
    return fmt, active, [(a, str(b)) for a, b in conf.items()] + extra, data, plan


def run(make, fmt, active, conf, data, plan, is_hip):
    if "pairs" in data:   # the same instances either as three columns (HIP engine) or as CSR rows (everything else)
        pu, pp, pq = data["pairs"]
        csr = sa.pairs_as_csr(pu, pp, pq)
        data = dict(train=csr, pairs=data["pairs"])

    def fresh(model_path=None):
        t = make(fmt, active, plan.get("extend", 0))
        t.seed(11)
        if model_path:   # warm start (svd_feature.cpp:175-182, continue training from a saved model)
            t.load_model(model_path)
        for k, v in conf:
            t.set_param(k, v)
        if not model_path:
            t.init_model()
        t.init_trainer()
        if is_hip:
            for k, v in plan["knobs"].items():
                t.set_knob(k, v)
            if plan["window"]:
                t.set_knob("stage_window", plan["window"])
        return t
    t = fresh()
    ds = None
    peeks = []
    for r in range(plan["rounds"]):
        t.set_round(r)
        if "train" in data:
            d = data["train"]
            if plan["single"]:   # one update(Elem) per instance, now and then a predict(Elem) in between (the reference CLI's calls)
                for j in range(d.num_row):
                    t.update_csr(*d.row(j))
                    if plan["peek"] and j % 37 == 5:
                        peeks.append(t.predict_csr(*d.row((j * 7) % d.num_row)))
            elif is_hip and plan["resident"]:
                ds = ds or (t.dataset_from_pairs(*data["pairs"]) if "pairs" in data else t.dataset_from_csr(d))
                t.train_dataset(ds)
            elif plan["chunk"]:
                for st in range(0, d.num_row, plan["chunk"]):
                    t.update_batch(d.slice_rows(st, st + plan["chunk"]))
            else:
                t.update_batch(d)
        else:
            if is_hip and plan["resident"] and not plan["peek"]:
                ds = ds or t.dataset_from_blocks(data["train_blocks"])
                t.train_dataset(ds)
            else:
                for j, b in enumerate(data["train_blocks"]):
                    t.update_block(b)
                    if plan["peek"] and j % 5 == 2 and b.extend_tag in (0, 2):   # predict(block) between whole users
                        peeks.extend(t.predict_block(data["train_blocks"][(j * 3) % len(data["train_blocks"])]).tolist()
                                     if data["train_blocks"][(j * 3) % len(data["train_blocks"])].extend_tag == 0 else [])
        t.finish_round()
        if plan.get("reload") and r == 0 and plan["rounds"] > 1:   # save, drop the trainer, load into a new one, go on
            fd, path = tempfile.mkstemp(suffix=".model")
            os.close(fd)
            t.save_model(path)
            if ds is not None:
                ds.close()
                ds = None
            t.close()
            t = fresh(path)
            os.unlink(path)
    if "train" in data:
        pred = t.predict_batch(data["train"])
    else:
        pred = np.concatenate([t.predict_block(b) for b in data["train_blocks"]] + [np.zeros(0, np.float32)])
    views = {}
    for v in VIEWS:
        try:
            a = t.view(v)
        except Exception:
            a = None
        views[v] = None if a is None else np.array(a, copy=True)
    t.close()
    return views, np.concatenate([np.array(pred, np.float32, copy=True), np.array(peeks, np.float32)])


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--wide", action="store_true", help="factor widths 257..1024 (general kernels, several registers per lane)")
    ap.add_argument("--big", action="store_true", help="up to 3000 users / 800 items / 40 K instances per data set")
    ap.add_argument("--only", type=int, default=-1, help="run only this iteration (the generator is advanced up to it)")
    ap.add_argument("--reference", action="store_true", help="compare the oracle port with the compiled reference instead of the GPU")
    ap.add_argument("--trace", default="", help="file that always holds the configuration being run (to locate a crash)")
    a = ap.parse_args(argv)
    rng = np.random.default_rng(a.seed)
    stats = dict(iters=0, exact=0, tolerance=0, skipped=0)
    for it in range(a.iters):
        with tempfile.TemporaryDirectory() as tmp:
            fmt, active, conf, data, plan = draw(rng, tmp, a.wide, a.big)
            if a.only >= 0 and it != a.only:
                continue
            if a.trace:
                with open(a.trace, "w") as f:
                    f.write("iteration %d format %d active %d\n%s\n%s\n%s\n" % (it, fmt, active, conf, plan,
                            {k: (len(v) if isinstance(v, list) else v.num_row) for k, v in data.items()}))
            try:
                ov, op = run(lambda f, x, e=0: oracle.OracleTrainer("port", f, x, e), fmt, active, conf, data, plan, False)
            except Exception as e:   # configuration the reference rejects (e.g. an id out of a shrunken range)
                stats["skipped"] += 1
                continue
            if a.reference:
                hv, hp = run(lambda f, x, e=0: oracle.OracleTrainer("reference_full" if e else "reference", f, x, e), fmt, active, conf, data, plan, False)
            elif sa.device_count() == 0:   # dry run of the generator and the oracle half on a box without a GPU
                stats["skipped"] += 1
                continue
            else:
                try:
                    hv, hp = run(lambda f, x, e=0: sa.Trainer(f, x, e), fmt, active, conf, data, plan, True)
                except sa.SvdfError as e:
                    print("iteration %d: engine refused what the oracle ran: %s\n%s %s" % (it, e, conf, plan), file=sys.stderr)
                    sys.exit(1)
            tol = False   # the device restates glibc's expf (svdf_device.h: glibc_expf): sigmoid links compare with == too
            for v in VIEWS:
                if ov[v] is None or hv[v] is None:
                    continue
                ok = np.allclose(hv[v], ov[v], rtol=2e-5, atol=2e-6, equal_nan=True) if tol else same_bits(hv[v], ov[v])
                if not ok:
                    print("iteration %d: %s differs\nformat %d active %d\n%s\n%s" % (it, v, fmt, active, conf, plan), file=sys.stderr)
                    sys.exit(1)
            ok = np.allclose(hp, op, rtol=2e-5, atol=2e-6, equal_nan=True) if tol else same_bits(hp, op)
            if not ok:
                print("iteration %d: predictions differ\nformat %d active %d\n%s\n%s" % (it, fmt, active, conf, plan), file=sys.stderr)
                sys.exit(1)
            stats["iters"] += 1
            stats["tolerance" if tol else "exact"] += 1
    print(json.dumps(stats))
    return stats


if __name__ == "__main__":
    main()
