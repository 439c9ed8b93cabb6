"""Rank-pair input drawn ON THE DEVICE (svdf_randstream.cpp + svdf_k_sample.hip): for candidate files of the
demo/pairwiseRank shape the pairs are sampled in HBM from the libc rand() stream the host sampler would have consumed.
Same draws -> byte-identical models and the same rand() position afterwards as the host sampler (knob device_rank = 0),
pass after pass; where oracle/_ref is present also against the reference's own generator feeding the engine."""
import ctypes
import os
import sys
import time

import numpy as np
import pytest

import cases
import svdfeature_amd as sa

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from perf_rank_input import write_candidates  # noqa: E402

pytestmark = pytest.mark.gpu

libc = ctypes.CDLL(None)
libc.rand.restype = ctypes.c_int


def _conf(nu, ni, k, **kw):
    base = [("num_user", nu), ("num_item", ni), ("num_global", 0), ("num_factor", k), ("num_ufeedback", 0), ("learning_rate", "0.01"),
            ("wd_user", "0.004"), ("wd_item", "0.004"), ("no_user_bias", 1), ("ui_init_sigma", "0.05")]
    return base + [(a, b) for a, b in kw.items()]


def _run(src, conf, passes, device_rank, seed=10):
    t = sa.Trainer(1, 3)
    t.seed(seed)
    for k, v in conf:
        t.set_param(k, str(v))
    t.init_model()
    t.init_trainer()
    t.set_knob("device_rank", device_rank)
    rows, batches = [], []
    for r in range(passes):
        t.set_round(r)
        ds = t.dataset_from_rank_buffer_file(src)
        rows.append(ds.num_row)
        batches.append(ds.num_batches)
        assert ds.kind == 2 or ds.num_row == 0
        t.train_dataset(ds)
        t.finish_round()
    views = {n: t.view(n).copy() for n in ("W_user", "W_item", "i_bias")}
    nxt = [libc.rand() for _ in range(4)]   # where libc's generator stands after the passes
    dev_passes = t.counter(7)
    t.close()
    return views, rows, batches, nxt, dev_passes


@pytest.mark.parametrize("keys", [{}, {"rank_sample_num": 5, "rank_sample_max": 4}, {"rank_sample_num": 40}, {"pos_sample_lowerb": "0.5", "neg_sample_upperb": "0.5"}],
                         ids=["default", "num_max", "num40", "bounds"])
@pytest.mark.parametrize("users,rows,k", [(300, 9, 16), (2000, 33, 64), (50, 1, 8)])
def test_device_sampler_equals_host_sampler(users, rows, k, keys, tmp_path):
    src = str(tmp_path / "cand.buffer")
    items = 150
    write_candidates(src, users, rows, items, seed=users + rows)
    conf = _conf(users, items, k, **keys)
    dv, drows, dbat, dnext, dpass = _run(src, conf, 3, 1)
    hv, hrows, hbat, hnext, hpass = _run(src, conf, 3, 0)
    assert dpass == 3 and hpass == 0, "the device path was not taken"
    assert drows == hrows and dbat == hbat
    assert dnext == hnext, "libc rand() stands elsewhere after the device passes"
    for name in dv:
        assert np.array_equal(dv[name].view(np.uint32), hv[name].view(np.uint32)), name


def test_device_sampler_declines_what_it_does_not_cover(tmp_path):
    """rows with global features / several item entries, blocks with feedback, rank_sample_method = 1: host sampler as before"""
    from svdfeature_amd import data as D
    src = str(tmp_path / "rich.buffer")
    D.write_ugroup_buffer(src, cases.rank_blocks(100, 60, 50, 8, 900))
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=60, num_item=50, num_global=8, num_factor=8, num_ufeedback=50)
    t = sa.Trainer(1, 3)
    t.seed(10)
    for k, v in [(a, b) for a, b in conf if a != "base_score"]:
        t.set_param(k, v)
    t.init_model()
    t.init_trainer()
    ds = t.dataset_from_rank_buffer_file(src)
    assert t.counter(7) == 0 and ds.num_row > 0
    src2 = str(tmp_path / "cand.buffer")
    write_candidates(src2, 100, 8, 40, seed=3)
    conf2 = _conf(100, 40, 8, rank_sample_method=1)
    _, _, _, _, dpass = _run(src2, conf2, 1, 1)
    assert dpass == 0


def test_device_sampler_pass_time(tmp_path):
    """100 K users x 64 candidates (3.2 M pairs per pass, k=128): sampling + scheduling of a pass on the device, against the
    host sampler + host scheduler of round 1 (0.44 s)."""
    src = str(tmp_path / "cand.buffer")
    users, rows, items = 100_000, 64, 100_000
    write_candidates(src, users, rows, items, seed=7)
    conf = _conf(users, items, 128)
    out = {}
    for mode in (1, 0):
        t = sa.Trainer(1, 3)
        t.seed(10)
        for k, v in conf:
            t.set_param(k, str(v))
        t.init_model()
        t.init_trainer()
        t.set_knob("device_rank", mode)
        t.dataset_from_rank_buffer_file(src).close()   # first pass: file parse + upload of the candidates
        ts = []
        for _ in range(3):
            t0 = time.time()
            ds = t.dataset_from_rank_buffer_file(src)
            ts.append(time.time() - t0)
            n = ds.num_row
            t1 = time.time()
            t.train_dataset(ds)
            t.synchronize()
            tt = time.time() - t1
            ds.close()
        out[mode] = (min(ts), n, tt)
        t.close()
    print("rank pass build: device %.3f s, host %.3f s (%d pairs); training %.3f s" % (out[1][0], out[0][0], out[1][1], out[1][2]))
    assert out[1][1] == out[0][1]
    assert out[1][0] < 0.15
