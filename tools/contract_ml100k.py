"""The N-rank accuracy contract on the one REAL data set in the reference's tree (ML-100K, demo/basicMF: 943 x 1682, skewed users and items), through the
reference's own CLI linked against the engine (oracle/_ref/svd_feature_amd, `amd:gpus = N` in the config file, no amd:window: the staged path cuts its
windows from the data on line) against the unmodified reference binary: held-out RMSE after 5 and 40 rounds, 2 / 4 / 8 virtual ranks, both steps.
python tools/contract_ml100k.py [window_per_target_max ...]   (build container or a box with oracle/_ref)"""
import os
import subprocess
import sys
import tempfile

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import cases
import svdfeature_amd as sa
from svdfeature_amd import data as D

REFDIR = os.path.join("oracle", "_ref")
REF_CLI, AMD_CLI = os.path.abspath(os.path.join(REFDIR, "svd_feature")), os.path.abspath(os.path.join(REFDIR, "svd_feature_amd"))
base, test = cases.ml100k()
conf = cases.conf_with(cases.BASICMF_CONF, num_factor=16)


def run(cli, extra, rounds, tag):
    d = tempfile.mkdtemp(prefix="c100k_" + tag)
    D.write_csr_buffer(os.path.join(d, "train.buffer"), base)
    with open(os.path.join(d, "run.conf"), "w") as f:
        for k, v in conf + extra + [("buffer_feature", '"train.buffer"'), ("model_out_folder", '"./"')]:
            f.write("%s = %s\n" % (k, v))
    p = subprocess.run([cli, "run.conf", "num_round=%d" % rounds, "silent=1"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert p.returncode == 0, p.stdout.decode()[-2000:]
    t = sa.Trainer(0, 0)
    t.load_model(os.path.join(d, "%04d.model" % rounds))
    t.init_trainer()
    r = cases.rmse(t.predict_batch(test), test.row_label)
    t.close()
    return r


SWEEP = [tuple(int(x) for x in s.split(",")) for s in sys.argv[1:]] or [None]
for sw in SWEEP:
  extra_sw = [] if sw is None else [("amd:window_per_target", str(sw[0])), ("amd:window_per_target_max", str(sw[1]))]
  print("== window rule: %s" % ("defaults" if sw is None else "mean %d / max %d updates per row and window" % sw), flush=True)
  worst = 0.0
  for rounds in (5, 40):
    ref = run(REF_CLI, [], rounds, "ref")
    print("rounds %2d: reference CLI rmse %.6f" % (rounds, ref), flush=True)
    for step in ("minibatch", "levels"):
        for gpus in (2, 4, 8):
            r = run(AMD_CLI, [("amd:gpus", str(gpus)), ("amd:step", step)] + extra_sw, rounds, "amd")
            worst = max(worst, abs(r - ref))
            print("   amd:gpus = %d, amd:step = %-9s rmse %.6f (%+.2e)" % (gpus, step, r, r - ref), flush=True)
  print("max |dRMSE| = %.2e (contract 1e-4)" % worst, flush=True)
