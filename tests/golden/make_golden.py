#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the COMPILED REFERENCE (oracle/_ref/libsvdf_ref.so).

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

Outputs (data only -- inputs and expected outputs, no reference source):
  ml100k_ua.npz        ML-100K ua.base (in the shuffled order of demo/basicMF/ua.base.basicfeature)
                       and ua.test as 0-based (user, item, rating) arrays
  scenarios.npz        for every scenario in tests/scenarios.py: md5 of the initial and final model
                       file the reference wrote, 256 sampled parameters, predictions, RMSE
  variants.npz         the same digests for tests/scenarios.py's VARIANT_SCENARIOS (extend_type 1 / 2 / 15), written from the
                       reference's default factory (oracle/_ref/libsvdf_ref_full.so); `make_golden.py variants` writes only this
  fixtures/            the reference tree's own committed binary fixtures for the path
                       (demo/basicMF/ua.base.buffer, ua.test.buffer) and its golden prediction
                       (demo/basicMF/eg.pred.txt)
"""
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))
REF = "/root/reference"


def make_ml100k():
    def load(path):
        u, i, r = [], [], []
        for line in open(path):
            t = line.split()
            assert t[1:4] == ["0", "1", "1"]
            r.append(int(t[0]))
            u.append(int(t[4].split(":")[0]))
            i.append(int(t[5].split(":")[0]))
        return np.array(u, np.uint16), np.array(i, np.uint16), np.array(r, np.uint8)
    bu, bi, br = load(os.path.join(REF, "demo/basicMF/ua.base.basicfeature"))
    tu, ti, tr = load(os.path.join(REF, "demo/basicMF/ua.test.basicfeature"))
    np.savez_compressed(os.path.join(HERE, "ml100k_ua.npz"), base_u=bu, base_i=bi, base_r=br, test_u=tu, test_i=ti, test_r=tr)
    print("ml100k: %d train, %d test" % (br.size, tr.size))


def make_scenarios():
    from oracle import oracle
    import scenarios
    oracle.build()
    assert oracle.have_reference(), "oracle/_ref/libsvdf_ref.so missing: run make -C oracle in the build container"
    out = {}
    for name in scenarios.SCENARIOS:
        res = scenarios.run_scenario(name, lambda f, a: oracle.OracleTrainer("reference", f, a))
        for k, v in scenarios.digest(res).items():
            out["%s/%s" % (name, k)] = np.asarray(v)
        print("%-28s rmse=%.6f model_md5=%s" % (name, res["rmse"], out["%s/model_md5" % name]))
    np.savez_compressed(os.path.join(HERE, "scenarios.npz"), **out)


def make_variants():
    """variant solvers (extend_type 1 / 2 / 15) from the reference's DEFAULT factory: oracle/_ref/libsvdf_ref_full.so"""
    from oracle import oracle
    import scenarios
    oracle.build()
    assert oracle.have_reference_full(), "oracle/_ref/libsvdf_ref_full.so missing: run make -C oracle in the build container"
    out = {}
    for name in scenarios.VARIANT_SCENARIOS:
        res = scenarios.run_scenario(name, lambda f, a, e: oracle.OracleTrainer("reference_full", f, a, e))
        for k, v in scenarios.digest(res).items():
            out["%s/%s" % (name, k)] = np.asarray(v)
        print("%-36s rmse=%.6f model_md5=%s len=%d" % (name, res["rmse"], out["%s/model_md5" % name], len(res["model"])))
    np.savez_compressed(os.path.join(HERE, "variants.npz"), **out)


def copy_fixtures():
    dst = os.path.join(HERE, "fixtures")
    os.makedirs(dst, exist_ok=True)
    for f in ("ua.base.buffer", "ua.test.buffer", "eg.pred.txt", "ua.base.example", "ua.test.example"):
        shutil.copyfile(os.path.join(REF, "demo/basicMF", f), os.path.join(dst, f))
        os.chmod(os.path.join(dst, f), 0o644)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "variants":
        make_variants()
        sys.exit(0)
    make_ml100k()
    copy_fixtures()
    make_scenarios()
    make_variants()
