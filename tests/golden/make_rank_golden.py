#!/usr/bin/env python3
"""Generate tests/golden/rank_input.npz from the COMPILED REFERENCE (build container only, needs oracle/_ref).

  sampler/<case>/...   what the reference's own PairwiseRankGenerator (apex_svd_data.cpp:812-1025, run by
                       oracle/_ref/ref_pairgen_dump) produces from tests/cases.py:rank_blocks over two passes after
                       srand(10): md5 over all generated blocks, row count, and for the first case the generated
                       labels / row_ptr / indices / values themselves
  e2e/model_rN         the NNNN.model files the reference's trainer CLI (oracle/_ref/svd_feature) writes for
                       cases.RANK_E2E_CONF (input_type = 2, active_type = 3) on rank_blocks(150, ..., seed 900)

Data only: inputs come from the seeded generators in tests/cases.py, outputs are what the reference computed.
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))

import cases  # noqa: E402
from svdfeature_amd import data as D  # noqa: E402

REFDIR = os.path.join(ROOT, "oracle", "_ref")


def main():
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for n, (name, graded, keys) in enumerate(cases.RANK_SAMPLER_CASES):
            blocks = cases.rank_blocks(200, 60, 50, 8, 500 + n, graded)
            src = os.path.join(tmp, name + ".in")
            dst = os.path.join(tmp, name + ".out")
            D.write_ugroup_buffer(src, blocks)
            subprocess.check_call([os.path.join(REFDIR, "ref_pairgen_dump"), src, dst, str(cases.RANK_SAMPLER_SEED),
                                   str(cases.RANK_SAMPLER_ROUNDS)] + ["%s=%s" % kv for kv in keys.items()],
                                  cwd=tmp, stdout=subprocess.DEVNULL)
            got = D.read_ugroup_buffer(dst)
            assert len(got) == cases.RANK_SAMPLER_ROUNDS * len(blocks)
            rows = sum(b.data.num_row for b in got)
            out["sampler/%s/md5" % name] = np.asarray(cases.blocks_digest(got))
            out["sampler/%s/num_row" % name] = np.asarray(rows)
            if n == 0:
                out["sampler/%s/label" % name] = np.concatenate([b.data.row_label for b in got])
                out["sampler/%s/row_len" % name] = np.concatenate([np.diff(b.data.row_ptr) for b in got]).astype(np.int32)
                out["sampler/%s/index" % name] = np.concatenate([b.data.feat_index for b in got])
                out["sampler/%s/value" % name] = np.concatenate([b.data.feat_value for b in got])
            print("%-20s rows=%d md5=%s" % (name, rows, out["sampler/%s/md5" % name]))
        # end to end through the reference's trainer CLI
        d = os.path.join(tmp, "e2e")
        os.mkdir(d)
        D.write_ugroup_buffer(os.path.join(d, "train.buffer"), cases.rank_blocks(150, 60, 50, 8, 900))
        with open(os.path.join(d, "run.conf"), "w") as f:
            for k, v in cases.RANK_E2E_CONF:
                f.write("%s = %s\n" % (k, v))
            f.write('buffer_feature = "train.buffer"\nmodel_out_folder = "./"\n')
        subprocess.check_call([os.path.join(REFDIR, "svd_feature"), "run.conf", "num_round=%d" % cases.RANK_E2E_ROUNDS, "silent=1"],
                              cwd=d, stdout=subprocess.DEVNULL)
        for r in range(cases.RANK_E2E_ROUNDS + 1):
            raw = open(os.path.join(d, "%04d.model" % r), "rb").read()
            out["e2e/model_r%d" % r] = np.frombuffer(raw, np.uint8)
        print("e2e: %d rounds, model %d bytes" % (cases.RANK_E2E_ROUNDS, len(raw)))
    np.savez_compressed(os.path.join(HERE, "rank_input.npz"), **out)


if __name__ == "__main__":
    main()
