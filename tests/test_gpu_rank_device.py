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


GOLD = np.load(os.path.join(cases.GOLDEN, "rank_input.npz"))


def _rich_trainer(keys, device_rank=1):
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=60, num_item=50, num_global=8, num_factor=8, num_ufeedback=50, wd_ufeedback="0.004",
                           ufeedback_init_sigma="0.01", no_user_bias=1, wd_global="0.001")
    t = sa.Trainer(1, 3)
    t.seed(3)
    for k, v in [(a, b) for a, b in conf if a != "base_score"] + list(keys.items()):
        t.set_param(k, str(v))
    t.init_model()
    t.init_trainer()
    t.set_knob("device_rank", device_rank)
    return t


@pytest.mark.parametrize("n", range(len(cases.RANK_SAMPLER_CASES)), ids=[c[0] for c in cases.RANK_SAMPLER_CASES])
def test_general_device_sampler_draws_the_reference_pairs(n, tmp_path):
    """The GENERAL device sampler (svdf_k_gsample.hip: rows with global entries, several user / item entries, blocks with feedback,
    rank_sample_method 0 and 1, pointwise output) against the golden vectors the REFERENCE's own PairwiseRankGenerator produced
    (tests/golden/rank_input.npz, oracle/_ref/ref_pairgen_dump): the generated blocks of two passes after srand(10), byte for byte,
    and libc's generator left where the host sampler leaves it."""
    from svdfeature_amd import data as D
    name, graded, keys = cases.RANK_SAMPLER_CASES[n]
    blocks = cases.rank_blocks(200, 60, 50, 8, 500 + n, graded)
    src = str(tmp_path / "in.buffer")
    D.write_ugroup_buffer(src, blocks)
    nxt = {}
    for mode in (1, 0):
        t = _rich_trainer(keys, mode)
        t.seed(cases.RANK_SAMPLER_SEED)
        got, rows = [], 0
        for r in range(cases.RANK_SAMPLER_ROUNDS):
            out = str(tmp_path / ("pass%d_%d.buffer" % (mode, r)))
            rows += t.rank_sample_buffer_file(src, out)
            got += D.read_ugroup_buffer(out)
        nxt[mode] = [libc.rand() for _ in range(4)]
        assert t.counter(7) == (cases.RANK_SAMPLER_ROUNDS if mode else 0), "device sampler %staken" % ("not " if mode else "")
        assert rows == int(GOLD["sampler/%s/num_row" % name]) == sum(b.data.num_row for b in got)
        if n == 0:
            np.testing.assert_array_equal(np.concatenate([b.data.row_label for b in got]).view(np.uint32), GOLD["sampler/%s/label" % name].view(np.uint32))
            np.testing.assert_array_equal(np.concatenate([np.diff(b.data.row_ptr) for b in got]), GOLD["sampler/%s/row_len" % name])
            np.testing.assert_array_equal(np.concatenate([b.data.feat_index for b in got]), GOLD["sampler/%s/index" % name])
            np.testing.assert_array_equal(np.concatenate([b.data.feat_value for b in got]).view(np.uint32), GOLD["sampler/%s/value" % name].view(np.uint32))
        assert cases.blocks_digest(got) == str(GOLD["sampler/%s/md5" % name])
        t.close()
    assert nxt[1] == nxt[0], "libc rand() stands elsewhere after the device passes"


@pytest.mark.parametrize("keys", [{}, {"rank_sample_method": 1}, {"rank_sample_pointwise": 1}, {"rank_sample_method": 1, "rank_sample_gap": "1.5"}],
                         ids=["posneg", "cmp", "pointwise", "cmp_wide"])
def test_rich_rank_input_trains_identically_from_device_and_host_samplers(keys, tmp_path):
    """input_type = 2 on rich candidate files (global entries, side user entries, several item entries, implicit feedback, graded
    labels): passes drawn in HBM -> byte-identical parameters and rand() position to passes drawn by the host sampler."""
    from svdfeature_amd import data as D
    src = str(tmp_path / "rich.buffer")
    D.write_ugroup_buffer(src, cases.rank_blocks(300, 60, 50, 8, 900, graded=bool(keys.get("rank_sample_method"))))
    res = {}
    for mode in (1, 0):
        t = _rich_trainer(keys, mode)
        t.seed(10)
        rows = []
        for r in range(3):
            t.set_round(r)
            ds = t.dataset_from_rank_buffer_file(src)
            rows.append(ds.num_row)
            t.train_dataset(ds)
            t.finish_round()
        res[mode] = ({v: t.view(v).copy() for v in ("W_user", "W_item", "i_bias", "g_bias", "W_ufeedback")}, rows, [libc.rand() for _ in range(4)], t.counter(7))
        t.close()
    assert res[1][3] == 3 and res[0][3] == 0
    assert res[1][1] == res[0][1] and sum(res[1][1]) > 0
    assert res[1][2] == res[0][2]
    for v in res[1][0]:
        assert np.array_equal(res[1][0][v].view(np.uint32), res[0][0][v].view(np.uint32)), v


def test_big_blocks_through_the_device_sort(tmp_path):
    """blocks of thousands of rows with five distinct labels: the per-block std::sort restatement on the device (svdf_stdsort.h) at the
    sizes where introsort partitions many times; device-drawn pass == host-drawn pass, byte for byte"""
    from svdfeature_amd import data as D
    rng = np.random.default_rng(4)
    blocks = []
    for b in range(6):
        nrow = int(rng.integers(1500, 6000))
        rows = [(float(rng.integers(1, 6)), [], [(b, 1.0)], [(int(rng.integers(0, 50)), 1.0)]) for _ in range(nrow)]
        blocks.append(D.PlusBlock(np.zeros(0, np.uint32), np.zeros(0, np.float32), sa.CSRData.from_rows(rows), 0))
    src = str(tmp_path / "big.buffer")
    D.write_ugroup_buffer(src, blocks)
    out = {}
    for mode in (1, 0):
        t = _rich_trainer({"rank_sample_method": 1}, mode)
        t.seed(10)
        o = str(tmp_path / ("o%d" % mode))
        t.rank_sample_buffer_file(src, o)
        out[mode] = cases.blocks_digest(D.read_ugroup_buffer(o))
        assert t.counter(7) == mode
        t.close()
    assert out[1] == out[0]


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
