"""Exact passes over ratings with HOT rows (svdf_pivot.cpp): runs of up to 64 consecutive ratings of a popular item (or user) are
walked by one wave with that row in registers -- the SVD++ user-unit walker on transposed parameters when the hot side is the items --, the other
ratings go through the contract kernel, all of it levelled together.  update_inner on (user:1, item:1) is symmetric in the two rows, so the model
must equal the level-by-level pass (knob pivot_exec = 0) and the oracle bit for bit."""
import numpy as np
import pytest

import cases
import svdfeature_amd as sa

pytestmark = pytest.mark.gpu
NAMES = ("W_user", "W_item", "u_bias", "i_bias")


def _run(u, i, r, nu, ni, pivot, k=64, passes=2, pivot_min=256, extra=()):
    t = sa.Trainer(0, 0)
    t.seed(10)
    for kk, v in cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=k) + list(extra):
        t.set_param(kk, str(v))
    t.init_model()
    t.init_trainer()
    t.set_knob("pivot_exec", pivot)
    t.set_knob("runs_exec", 0)   # (the plain schedule, not runs of an item's ratings, is what the unit form is compared with)
    t.set_knob("pivot_min", pivot_min)
    ds = t.dataset_from_triples(u, i, r)
    for _ in range(passes):
        t.train_dataset(ds)
    t.synchronize()
    return {n: t.view(n).copy() for n in NAMES}, ds.kind, ds.num_batches, t.counter(22), t, ds


@pytest.mark.parametrize("nu,ni,n,k,swap", [(20000, 300, 200000, 64, False), (5000, 40, 60000, 128, False), (300, 20000, 150000, 64, True),
                                            (2000, 2000, 100000, 32, False), (100000, 3000, 1500000, 64, False)])
def test_hot_rows_walked_as_units_equal_the_level_by_level_pass(nu, ni, n, k, swap):
    u, i, r = cases.planted_triples(n, ni if swap else nu, nu if swap else ni, seed=nu + n, zipf=True)
    if swap:   # the hot side is the USERS: no transposition
        u, i = i, u
    a, kind_a, levels_a, pa, ta, dsa = _run(u, i, r, nu, ni, 0, k)
    b, kind_b, levels_b, pb, tb, dsb = _run(u, i, r, nu, ni, 1, k)
    assert kind_a == 0 and kind_b == 9 and pa == 0 and pb == 2
    assert levels_b < levels_a, (levels_a, levels_b)
    for name in NAMES:
        assert np.array_equal(a[name].view(np.uint32), b[name].view(np.uint32)), (name, levels_a, levels_b)
    # the evaluator scores such a data set like a plain one
    sa_, ca = ta.eval_dataset(dsa)
    sb_, cb = tb.eval_dataset(dsb)
    assert ca == cb == n and abs(sa_ - sb_) <= 1e-9 * abs(sa_)


def test_hot_rows_pass_equals_the_oracle():
    from oracle import oracle
    oracle.build()
    nu, ni, n = 3000, 60, 50000
    u, i, r = cases.planted_triples(n, nu, ni, seed=7, zipf=True)
    got, kind, _, _, _, _ = _run(u, i, r, nu, ni, 1, passes=1)
    assert kind == 9
    o = oracle.OracleTrainer("port", 0, 0)
    o.seed(10)
    for kk, v in cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=64):
        o.set_param(kk, v)
    o.init_model()
    o.init_trainer()
    o.update_batch(sa.CSRData.from_triples(u, i, r))
    for name in NAMES:
        assert np.array_equal(got[name].view(np.uint32), o.view(name).view(np.uint32)), name


def test_configurations_outside_the_symmetric_form_keep_plain_levels():
    nu, ni, n = 3000, 60, 40000
    u, i, r = cases.planted_triples(n, nu, ni, seed=2, zipf=True)
    for extra in ([("reg_method", 1)], [("no_user_bias", 1)], [("user_nonnegative", 1)], [("up:wd", 0.01), ("up:bound", 100), ("up:wd", 0.02), ("up:bound", 3000)]):
        _, kind, _, p, _, _ = _run(u, i, r, nu, ni, 1, passes=1, extra=extra)
        assert kind != 9 and p == 0, extra
    # uniform ratings have no hot row
    u, i, r = cases.planted_triples(n, nu, 4000, seed=2)
    _, kind, _, _, _, _ = _run(u, i, r, nu, 4000, 1, passes=1)
    assert kind == 0


def test_predict_dataset_of_a_hot_row_data_set_reports_in_file_order():
    """dataset_from_triples may pick the unit form by itself (pivot_exec defaults to 1); the documented inference call
    `predict_dataset` (out[num_row], file order) must work on it and equal the plain data set's predictions bit for bit"""
    nu, ni, n = 20000, 300, 200000
    u, i, r = cases.planted_triples(n, nu, ni, seed=11, zipf=True)
    a, kind_a, _, _, ta, dsa = _run(u, i, r, nu, ni, 0)
    b, kind_b, _, _, tb, dsb = _run(u, i, r, nu, ni, 1)
    assert kind_a == 0 and kind_b == 9
    pa, pb = ta.predict_dataset(dsa), tb.predict_dataset(dsb)
    assert pa.shape == pb.shape == (n,)
    assert np.array_equal(pa.view(np.uint32), pb.view(np.uint32))
    # and against the per-row scorer on a sample of file positions
    sel = np.arange(0, n, 997)
    ref = tb.predict_batch(sa.CSRData.from_triples(u[sel], i[sel], r[sel]))
    assert np.array_equal(np.asarray(ref, np.float32).view(np.uint32), pb[sel].view(np.uint32))
