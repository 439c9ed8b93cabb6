#!/usr/bin/env python3
"""Rate of the rank-pair input path: pairs drawn on the host per second by the native sampler, time to
schedule + upload one pass, and the device training rate on it, next to the reference's own generator (oracle/_ref/
ref_pairgen_dump, when present) run over the same user-group buffer file.  Secondary numbers for DESIGN.md; writes one
JSON line.  Lives under tests/ because it runs an oracle binary (perf_* files are not collected by pytest).

    python tests/perf_rank_input.py [--users 100000] [--rows 64] [--items 100000] [--factor 128] [--passes 3]
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import svdfeature_amd as sa  # noqa: E402


def write_candidates(path, users, rows, items, seed):
    """One block per user: `rows` candidate items with label 0/1, one user id and one item id each (the shape of
    demo/pairwiseRank).  Blocks are fixed-size, so the file is one structured array."""
    rng = np.random.default_rng(seed)
    rec = np.dtype([("nfb", "<i4"), ("num_row", "<i4"), ("num_val", "<i4"), ("row_ptr", "<i4", (3 * rows + 1,)),
                    ("label", "<f4", (rows,)), ("index", "<u4", (2 * rows,)), ("value", "<f4", (2 * rows,))])
    a = np.zeros(users, rec)
    a["num_row"] = rows
    a["num_val"] = 2 * rows
    rp = np.zeros(3 * rows + 1, np.int32)
    rp[1::3] = 2 * np.arange(rows)
    rp[2::3] = 2 * np.arange(rows) + 1
    rp[3::3] = 2 * np.arange(rows) + 2
    a["row_ptr"] = rp
    a["label"] = rng.integers(0, 2, (users, rows)).astype(np.float32)
    a["index"][:, 0::2] = np.arange(users, dtype=np.uint32)[:, None]
    a["index"][:, 1::2] = rng.integers(0, items, (users, rows), dtype=np.uint32)
    a["value"] = 1.0
    with open(path, "wb") as f:
        np.array([users, 0, rows, 2 * rows], np.int32).tofile(f)
        a.tofile(f)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--users", type=int, default=100000)
    ap.add_argument("--rows", type=int, default=64)
    ap.add_argument("--items", type=int, default=100000)
    ap.add_argument("--factor", type=int, default=128)
    ap.add_argument("--passes", type=int, default=3)
    ap.add_argument("--knob", action="append", default=[], help="tuning knob name=value, repeatable")
    args = ap.parse_args()
    out = {"users": args.users, "rows_per_user": args.rows, "items": args.items, "factor": args.factor}
    with tempfile.TemporaryDirectory() as tmp:
        src = os.path.join(tmp, "cand.buffer")
        write_candidates(src, args.users, args.rows, args.items, 7)
        conf = [("num_user", args.users), ("num_item", args.items), ("num_global", 0), ("num_factor", args.factor), ("num_ufeedback", 0),
                ("learning_rate", 0.005), ("wd_user", 0.004), ("wd_item", 0.004), ("active_type", 3), ("no_user_bias", 1)]
        # host sampler alone (pairs written to a buffer file on tmpfs)
        h = sa.Trainer(1, 3, device=-2)
        h.seed(10)
        t0 = time.perf_counter()
        pairs = h.rank_sample_buffer_file(src, os.path.join(tmp, "pairs.buffer"))
        out["pairs_per_pass"] = pairs
        out["sampler_to_file_pairs_per_s"] = pairs / (time.perf_counter() - t0)
        h.close()
        os.unlink(os.path.join(tmp, "pairs.buffer"))
        ref = os.path.join(ROOT, "oracle", "_ref", "ref_pairgen_dump")
        if os.path.exists(ref):
            t0 = time.perf_counter()
            subprocess.check_call([ref, src, os.path.join(tmp, "ref.buffer"), "10", "1"], cwd=tmp, stdout=subprocess.DEVNULL)
            out["reference_generator_to_file_pairs_per_s"] = pairs / (time.perf_counter() - t0)
            os.unlink(os.path.join(tmp, "ref.buffer"))
        if sa.device_count() > 0:
            t = sa.Trainer(1, 3)
            t.seed(10)
            for k, v in conf:
                t.set_param(k, str(v))
            t.init_model()
            t.init_trainer()
            for kv in args.knob:
                t.set_knob(kv.split("=")[0], int(kv.split("=")[1]))
            build, train = [], []
            for r in range(args.passes):
                t.set_round(r)
                t0 = time.perf_counter()
                ds = t.dataset_from_rank_buffer_file(src)
                t.synchronize()
                t1 = time.perf_counter()
                t.train_dataset(ds)
                t.finish_round()
                t.synchronize()
                t2 = time.perf_counter()
                build.append(t1 - t0)
                train.append(t2 - t1)
                out["batches_per_pass"] = ds.num_batches
                out["kind"] = ds.kind
                n = ds.num_row
                ds.close()
            out["sample_schedule_upload_s"] = min(build)
            out["train_s"] = min(train)
            out["train_pairs_per_s"] = n / min(train)
            out["end_to_end_pairs_per_s"] = n / (min(build) + min(train))
            # the same with the next pass drawn on the background thread (svdf_rank_prefetch_buffer_file) while this thread
            # issues the current pass's launches
            t.synchronize()
            t0 = time.perf_counter()
            ds = t.dataset_from_rank_buffer_file(src)
            total = 0
            for r in range(args.passes):
                t.set_round(args.passes + r)
                if r + 1 < args.passes:
                    t.rank_prefetch_buffer_file(src)
                t.train_dataset(ds)
                total += ds.num_row
                nxt = t.dataset_from_rank_buffer_file(src) if r + 1 < args.passes else None
                t.finish_round()
                ds.close()
                ds = nxt
            t.synchronize()
            out["overlapped_end_to_end_pairs_per_s"] = total / (time.perf_counter() - t0)
            t.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
