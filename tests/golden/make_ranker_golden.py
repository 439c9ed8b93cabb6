#!/usr/bin/env python3
"""tests/golden/ranker.npz from the COMPILED REFERENCE's ranker (create_svd_ranker of oracle/_ref/libsvdf_ref.so): the int
results ISVDRanker::process returns for tests/test_ranker.py's cases.  Build container only (needs /root/reference)."""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

if __name__ == "__main__":
    from oracle import oracle
    import test_ranker
    oracle.build()
    assert oracle.have_reference()
    out = {}
    for name in test_ranker.RANK_CASES:
        with tempfile.TemporaryDirectory() as tmp:
            out[name] = test_ranker.run_ranker(lambda f: oracle.OracleRanker("reference", f, 0), name, tmp)
        print(name, out[name][:12], out[name].size)
    np.savez_compressed(os.path.join(HERE, "ranker.npz"), **out)
