"""Randomised differential run of the ranker: random models (format, factor width, side tables), candidate sets,
section streams (positives, bans, special samples, duplicate candidates = tied scores, candidates arriving mid-stream), top_k or
position mode, fed to the HIP engine line by line, as one pipelined svdf_ranker_process_rows call, or block by block (user-group
input), against the CPU ranker (the compiled reference when ties are in the draw and it is present, else the C port).
Prints one JSON line.  Not a pytest module (tests/test_gpu_fuzz.py runs a short draw)."""
import argparse
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def one_case(rng, sa, oracle, cases, tmp):
    from svdfeature_amd.data import PlusBlock
    fmt = int(rng.integers(0, 2))
    k = int(rng.choice([1, 3, 4, 7, 16, 31, 32, 48, 64, 100, 128, 200, 260, 300]))
    side = bool(rng.integers(0, 3) == 0)
    nu, ni, ng = int(rng.integers(20, 120)), int(rng.integers(30, 400)), int(rng.integers(1, 6))
    extra = []
    if side:
        fu, fi = os.path.join(tmp, "fu.txt"), os.path.join(tmp, "fi.txt")
        cases.write_side_table(fu, max(1, nu - 5), nu, 3)
        cases.write_side_table(fi, ni, ni, 4)
        extra = [("feature_user", fu), ("feature_item", fi)]
    kw = dict(num_user=nu, num_item=ni, num_global=ng, num_factor=k, wd_global=0.002, learning_rate=0.02,
              ui_init_sigma=float(rng.choice([0.01, 0.05, 0.3])))
    if fmt == 1:
        kw.update(num_ufeedback=ni, wd_ufeedback=0.004, ufeedback_init_sigma=0.05)
    conf = cases.conf_with(cases.BASICMF_CONF, **kw) + extra
    t = oracle.OracleTrainer("port", fmt, 0)
    t.seed(int(rng.integers(1, 1000)))
    for kk, v in conf:
        t.set_param(kk, v)
    t.init_model()
    t.init_trainer()
    if fmt == 0:
        t.update_batch(cases.sparse_feature_rows(300, nu, ni, ng, int(rng.integers(0, 99))))
    else:
        for b in cases.user_blocks(20, nu, ni, ni, int(rng.integers(0, 99))):
            t.update_block(b)
    path = os.path.join(tmp, "rank.model")
    t.save_model(path)
    t.close()
    ncand, nsec = int(rng.integers(8, 700)), int(rng.integers(1, 30))
    items, sections = cases.ranker_stream(ncand, nsec, nu, ni, ng, seed=int(rng.integers(0, 1 << 30)), spec=bool(rng.integers(0, 2)))
    ties = bool(rng.integers(0, 4) == 0) and oracle.have_reference()
    nlate = int(rng.integers(0, 40)) if rng.integers(0, 2) else 0
    late = sa.CSRData.from_rows([(0.0, [], [], [(int(c % 5) if ties else int(rng.integers(0, ni)), 1.0)]) for c in range(nlate)]) if nlate else None
    if ties:   # duplicates inside the first candidate set as well
        dup = sa.CSRData.from_rows([(0.0, [], [], [(int(c % 3), 1.0)]) for c in range(6)])
        items = sa.CSRData.concat([items, dup])
    cut = int(rng.integers(0, nsec + 1))
    parts = [items] + sections[:cut] + ([late] if late is not None else []) + sections[cut:]
    total = items.num_row + nlate
    top_k = int(rng.choice([0, 0, 1, 3, min(10, items.num_row - 4), min(50, items.num_row - 4)]))   # <= ranked candidates of every section
    # random duplicates may tie without being asked to: only the reference's sort is bound to the order inside a tie
    kind = "reference" if oracle.have_reference() else "port"
    mode = str(rng.choice(["lines", "bulk", "blocks"])) if fmt == 1 else str(rng.choice(["lines", "bulk"]))
    outs, errs = [], []
    for who in ("cpu", "gpu"):
        r = oracle.OracleRanker(kind, fmt, 0) if who == "cpu" else sa.Ranker(fmt, 0)
        for kk, v in extra + [("top_k", str(top_k))]:
            r.set_param(kk, v)
        r.load_model(path)
        r.init_ranker(total + int(rng.integers(0, 3)) if who == "gpu" else total + 2)
        res = []
        try:
          if mode == "blocks":
            for s, d in enumerate(parts):
                fb = np.sort(np.random.default_rng(s).choice(ni, size=min(3, ni), replace=False)).astype(np.uint32)
                res.append(r.process_block(PlusBlock(fb, np.full(fb.size, 0.5, np.float32), d, 0)))
          elif mode == "bulk" and who == "gpu":
            res.append(r.process_rows(sa.CSRData.concat(parts)))
          else:
            for d in parts:
                for i in range(d.num_row):
                    res.append(r.process(*d.row(i)))
        except Exception:   # the reference's asserts (e.g. "k can not exceed candidate size"): both sides must refuse
            errs.append(True)
        else:
            errs.append(False)
        outs.append(np.concatenate(res).astype(np.int32) if res else np.zeros(0, np.int32))
        r.close()
    if errs[0] or errs[1]:   # an error stream: only "both refused" is compared (the engine's bulk call reports nothing before it)
        ok = errs[0] == errs[1]
    else:
        ok = np.array_equal(outs[0], outs[1])
    return ok, dict(fmt=fmt, k=k, side=side, ncand=ncand, nsec=nsec, top_k=top_k, ties=ties, nlate=nlate, mode=mode, kind=kind)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    import cases
    import svdfeature_amd as sa
    from oracle import oracle
    oracle.build()
    rng = np.random.default_rng(a.seed)
    stats = dict(iters=0, exact=0, failed=0)
    tmp = tempfile.mkdtemp()
    for it in range(a.iters):
        ok, desc = one_case(rng, sa, oracle, cases, tmp)
        stats["iters"] += 1
        stats["exact" if ok else "failed"] += 1
        if not ok:
            print("MISMATCH", json.dumps(desc), flush=True)
    print(json.dumps(stats))
    return 0 if stats["failed"] == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
