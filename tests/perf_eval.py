"""Prediction / evaluation throughput on a resident data set (svd_feature_infer's test pass): basicMF shape of
BASELINE configs[1], svdf_predict_dataset and svdf_eval_dataset (RMSE reduced on the device) against the C port on a prefix.
Prints one JSON line.  Not a pytest module."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ratings", type=int, default=100_000_000)
    ap.add_argument("--users", type=int, default=1_000_000)
    ap.add_argument("--items", type=int, default=100_000)
    ap.add_argument("--factor", type=int, default=64)
    ap.add_argument("--cpu-sample", type=int, default=5_000_000)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--knob", action="append", default=[], help="tuning knob name=value, repeatable")
    a = ap.parse_args()
    import cases
    import svdfeature_amd as sa
    from oracle import oracle
    oracle.build()
    rng = np.random.default_rng(3)
    u = rng.integers(0, a.users, a.ratings, dtype=np.uint32)
    i = rng.integers(0, a.items, a.ratings, dtype=np.uint32)
    r = rng.integers(1, 6, a.ratings).astype(np.float32)
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=a.users, num_item=a.items, num_factor=a.factor)
    t, o = sa.Trainer(0, 0), oracle.OracleTrainer("port", 0, 0)
    for x in (t, o):
        x.seed(10)
        for k, v in conf:
            x.set_param(k, v)
        x.init_model()
        x.init_trainer()
    for kv in a.knob:
        t.set_knob(kv.split("=")[0], int(kv.split("=")[1]))
    ds = t.dataset_from_triples(u, i, r)
    t.predict_dataset(ds)
    t0 = time.time()
    for _ in range(a.reps):
        pred = t.predict_dataset(ds)
    dt_pred = (time.time() - t0) / a.reps
    t.eval_dataset(ds)
    t0 = time.time()
    for _ in range(a.reps):
        sse, cnt = t.eval_dataset(ds)
    dt_eval = (time.time() - t0) / a.reps
    n = min(a.cpu_sample, a.ratings)
    t0 = time.time()
    cpu = o.predict_batch(sa.CSRData.from_triples(u[:n], i[:n], r[:n]))
    dt_cpu = time.time() - t0
    byts = a.ratings * (2 * a.factor * 4 + 8 + 12)   # two rows, two bias words, the (user, item, label) record
    print(json.dumps({"ratings": a.ratings, "factor": a.factor,
                      "predict_dataset": {"s": dt_pred, "inst_per_s": a.ratings / dt_pred, "note": "scores to HBM + device-to-host copy of the predictions"},
                      "eval_dataset": {"s": dt_eval, "inst_per_s": a.ratings / dt_eval, "GBps": byts / dt_eval / 1e9, "frac_of_8TBps": byts / dt_eval / 8e12,
                                       "rmse": float(np.sqrt(sse / cnt))},
                      "cpu_port": {"sample": n, "inst_per_s": n / dt_cpu},
                      "predictions_identical_on_sample": bool(np.array_equal(pred[:n].view(np.uint32), np.asarray(cpu, np.float32).view(np.uint32)))}))


if __name__ == "__main__":
    main()
