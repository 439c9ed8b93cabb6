"""Latency of ONE predict(Elem) call through the C ABI (svdf_predict_csr): the per-instance path of ISVDTrainer::predict."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import cases
import svdfeature_amd as sa

nu, ni, k = 100000, 20000, 64
t = sa.Trainer(0, 0)
for kk, v in cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=k):
    t.set_param(kk, str(v))
t.init_model(); t.init_trainer()
u, i, r = cases.planted_triples(4000, nu, ni, seed=1)
d = sa.CSRData.from_triples(u, i, r)
rows = [d.row(j) for j in range(d.num_row)]
for j in range(200):
    t.predict_csr(*rows[j])
t0 = time.time()
for j in range(200, 4000):
    t.predict_csr(*rows[j])
dt = (time.time() - t0) / 3800
print("predict(Elem): %.1f us per call (python ctypes overhead included)" % (dt * 1e6))
t0 = time.time()
p = t.predict_batch(d)
print("predict_batch of 4000 rows: %.1f us total" % ((time.time() - t0) * 1e6))
