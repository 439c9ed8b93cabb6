import ctypes, os, sys, tempfile, hashlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import svdfeature_amd as sa
from perf_rank_input import write_candidates
libc = ctypes.CDLL(None); libc.rand.restype = ctypes.c_int
users, rows, items, k = 300, 9, 150, 16
tmp = tempfile.mkdtemp(); src = os.path.join(tmp, "c.buffer")
write_candidates(src, users, rows, items, seed=users + rows)
conf = [("num_user", users), ("num_item", items), ("num_global", 0), ("num_factor", k), ("num_ufeedback", 0), ("learning_rate", "0.01"),
        ("wd_user", "0.004"), ("wd_item", "0.004"), ("no_user_bias", 1), ("ui_init_sigma", "0.05")]
res = {}
for mode in (1, 0):
    t = sa.Trainer(1, 3); t.seed(10)
    for a, b in conf: t.set_param(a, str(b))
    t.init_model(); t.init_trainer(); t.set_knob("device_rank", mode)
    log = []
    for r in range(4):
        pk_before = sa.rand_peek(3).tolist()
        ds = t.dataset_from_rank_buffer_file(src)
        pk_after = sa.rand_peek(3).tolist()
        t.train_dataset(ds)
        h = hashlib.md5(t.view("W_item").tobytes()).hexdigest()[:8]
        log.append((ds.num_row, ds.num_batches, pk_before, pk_after, h))
    res[mode] = log
for r in range(4):
    print(r, "dev ", res[1][r]); print(r, "host", res[0][r])
