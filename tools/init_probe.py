"""init_model (SVDModel::rand_init) at the BASELINE shapes: the host loop (knob device_init = 0: libc rand() one draw at a time, the
reference's own cost) against the device path (svdf_k_init.hip); models compared bit for bit, libc's next draws compared."""
import ctypes
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import svdfeature_amd as sa

libc = ctypes.CDLL(None)
libc.rand.restype = ctypes.c_int

SHAPES = [("configs[1] basicMF 1M x 100K k=64", 0, dict(num_user=1_000_000, num_item=100_000, num_global=0, num_factor=64)),
          ("configs[3] SVD++ 40K x 100K k=128", 1, dict(num_user=40_000, num_item=100_000, num_global=0, num_factor=128, num_ufeedback=100_000)),
          ("configs[3] neighbourhood 1M x 100K k=128", 0, dict(num_user=1_000_000, num_item=100_000, num_global=10_000, num_factor=128))]


def run(fmt, kw, device_init):
    t = sa.Trainer(fmt, 0)
    t.set_knob("device_init", device_init)
    t.seed(10)
    for k, v in kw.items():
        t.set_param(k, str(v))
    t0 = time.perf_counter()
    t.init_model()
    t1 = time.perf_counter()
    t.init_trainer()
    t.synchronize()
    t2 = time.perf_counter()
    nxt = [libc.rand() for _ in range(4)]
    views = {n: t.view(n) for n in ("W_user", "W_item", "W_ufeedback")}
    return t1 - t0, t2 - t1, nxt, views, (t.counter(13), t.counter(14))


for name, fmt, kw in SHAPES:
    sa.Trainer(0, 0)   # runtime warm
    h = run(fmt, kw, 0)
    d = run(fmt, kw, 1)
    d2 = run(fmt, kw, 1)
    same = all((h[3][n] is None and d[3][n] is None) or np.array_equal(h[3][n].view(np.uint32), d[3][n].view(np.uint32)) for n in h[3])
    print(json.dumps({"shape": name, "host_init_s": round(h[0], 3), "host_init_trainer_s": round(h[1], 3), "device_init_s": round(d[0], 4),
                      "device_init_second_call_s": round(d2[0], 4), "device_init_trainer_s": round(d[1], 4), "bit_identical": bool(same),
                      "same_next_draws": h[2] == d[2], "values_decided_by_host_libm": d[4][0], "draws": d[4][1]}), flush=True)
