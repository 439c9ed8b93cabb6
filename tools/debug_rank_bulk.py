"""debug helper: bulk (pipelined) ranker rows vs line-by-line, prints the sections that differ"""
import os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import cases
import svdfeature_amd as sa
from oracle import oracle
oracle.build()
top_k, spec = 7, True
nu, ni, ng = 200, 1500, 4
conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_global=ng, num_factor=48, ui_init_sigma=0.1, wd_global=0.001)
t = oracle.OracleTrainer("port", 0, 0)
t.seed(5)
for kk, v in conf:
    t.set_param(kk, v)
t.init_model(); t.init_trainer()
t.update_batch(cases.sparse_feature_rows(3000, nu, ni, ng, 12))
path = os.path.join(tempfile.mkdtemp(), "m.model")
t.save_model(path)
items, sections = cases.ranker_stream(1200, 100, nu, ni, ng, seed=17 + top_k, spec=spec)
late = sa.CSRData.from_rows([(0.0, [], [], [(int(c % 11), 1.0)]) for c in range(40)])
parts = [items] + sections[:50] + [late] + sections[50:]
stream = sa.CSRData.concat(parts)
outs = {}
for name in ("lines", "bulk"):
    r = sa.Ranker(0, 0)
    r.set_param("top_k", str(top_k)); r.load_model(path); r.init_ranker(items.num_row + late.num_row)
    if name == "bulk":
        outs[name] = r.process_rows(stream)
    else:
        outs[name] = np.concatenate([r.process(*stream.row(i)) for i in range(stream.num_row)])
    print(name, "hosted", r.counter(1), len(outs[name]))
a, b = outs["lines"].reshape(-1, top_k), outs["bulk"].reshape(-1, top_k)
for s in range(a.shape[0]):
    if not np.array_equal(a[s], b[s]):
        print("section", s, "lines", a[s], "bulk", b[s], "labels", sections[s].row_label)
