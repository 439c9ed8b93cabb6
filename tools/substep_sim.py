#!/usr/bin/env python3
"""CPU simulation of the one-GPU window step with ORDERED SUB-STEPS on hot items (oracle/svdf_oracle.c: svdo_update_window_substeps) on a
scaled-down Zipf stream: held-out RMSE after P passes of (a) exact sequential SGD, (b) the stale window step with the round-5 rule (no row
more than 128 updates per window), (c) sub-steps of 128 inside far fewer windows.  usage: substep_sim.py [users items ratings passes]"""
import os
import sys
import time
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from benchlib import orders  # noqa: E402
from oracle import oracle  # noqa: E402
import svdfeature_amd.data as sd  # noqa: E402


def main():
    nu, ni, n, passes = [int(x) for x in (sys.argv[1:5] + ["100000", "10000", "10000000", "2"][len(sys.argv) - 1:])]
    u, i, r = orders.synth_zipf_triples(types.SimpleNamespace(Planted=bench.Planted), n + 200000, nu, ni, 4321)
    tu, ti, tl = u[n:], i[n:], r[n:]
    u, i, r = u[:n], i[:n], r[:n]
    cnt = np.bincount(i, minlength=ni).astype(np.float64)
    print("top item %d of %d (%.2f %%)" % (cnt.max(), n, 100 * cnt.max() / n), flush=True)
    args = types.SimpleNamespace(users=nu, items=ni, factor=64)
    test = sd.CSRData.from_triples(tu, ti, tl)

    def trainer():
        o = oracle.OracleTrainer("port", 0, 0)
        o.seed(10)
        for k, v in bench.conf_for(args):
            o.set_param(k, v)
        o.init_model()
        o.init_trainer()
        return o

    def windows(W):
        return [sd.CSRData.from_triples(u[n * w // W:n * (w + 1) // W], i[n * w // W:n * (w + 1) // W], r[n * w // W:n * (w + 1) // W]) for w in range(W)]

    def score(o):
        return bench.rmse(o.predict_batch(test), tl)
    t0 = time.time()
    o = trainer()
    whole = sd.CSRData.from_triples(u, i, r)
    for _ in range(passes):
        o.update_batch(whole)
    seq = score(o)
    print("sequential: rmse %.6f (%.0fs)" % (seq, time.time() - t0), flush=True)
    W5 = int(np.ceil(max((cnt * cnt).sum() / cnt.sum() / 24.0, cnt.max() / 128.0)))
    for name, W, sub in [("round-5 rule (== stale)", W5, 128)] + [("sub-steps of 128", int(x), 128) for x in os.environ.get("SUBSTEP_WINDOWS", "").split(",") if x]:
        t0 = time.time()
        o = trainer()
        ws = windows(W)
        for _ in range(passes):
            for d in ws:
                o.update_window_substeps(d, sub)
        print("%-22s %5d windows (top item %.0f per window): rmse %.6f  d = %+.2e  (%.0fs)" % (name, W, cnt.max() / W, score(o), score(o) - seq, time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
