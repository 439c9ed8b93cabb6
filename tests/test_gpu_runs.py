"""Runs of an item's consecutive ratings as the units of the contract workload's schedule (svdf_k_runs.hip / svdf_runs.cpp): formed in
HBM from file-order facts only (a rating may join the run headed at position h iff its user's previous rating lies before h), level-scheduled by the
device scheduler with 1 + R row slots per run, a lane group keeps the item's row across a run.  Same instances, every row's touches in file order, the
contract kernel's arithmetic: the model must equal the level-by-level pass over single instances (knob runs_exec = 0) and the oracle bit for bit."""
import numpy as np
import pytest

import cases
import svdfeature_amd as sa

pytestmark = pytest.mark.gpu
NAMES = ("W_user", "W_item", "u_bias", "i_bias")


def _trainer(nu, ni, knobs, k=64):
    t = sa.Trainer(0, 0)
    t.seed(10)
    for kk, v in cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=k):
        t.set_param(kk, str(v))
    t.init_model()
    t.init_trainer()
    t.set_knob("pivot_exec", 0)
    t.set_knob("runs_min_rows", 0)
    for k, v in knobs:
        t.set_knob(k, v)
    return t


def _run(u, i, r, nu, ni, knobs, passes=2, k=64):
    t = _trainer(nu, ni, knobs, k)
    ds = t.dataset_from_triples(u, i, r)
    for _ in range(passes):
        t.train_dataset(ds)
    t.synchronize()
    return {n: t.view(n).copy() for n in NAMES}, ds, t


@pytest.mark.parametrize("nu,ni,n,zipf,knobs", [
    (60000, 5000, 1500000, False, [("runs_len", 4)]),
    (60000, 5000, 1500000, False, [("runs_len", 7), ("runs_sets", 2)]),
    (20000, 300, 300000, True, [("runs_len", 4), ("runs_block", 256)]),
    (3000, 40, 60000, True, [("runs_len", 2)]),
    (500, 2000, 40000, False, [("runs_len", 3), ("runs_sets", 2), ("runs_block", 128)]),
    (64, 9, 5000, False, [("runs_len", 6)]),
])
def test_runs_equal_the_level_by_level_pass(nu, ni, n, zipf, knobs):
    u, i, r = cases.planted_triples(n, nu, ni, seed=nu + n, zipf=zipf)
    a, dsa, ta = _run(u, i, r, nu, ni, [("runs_exec", 0)])
    b, dsb, tb = _run(u, i, r, nu, ni, [("runs_exec", 1)] + knobs)
    assert dsa.kind == 0 and dsb.kind == 10 and tb.counter(23) == 2
    for name in NAMES:
        assert np.array_equal(a[name].view(np.uint32), b[name].view(np.uint32)), (name, dsa.num_batches, dsb.num_batches)
    # scoring: the evaluator's sum and predict_dataset's file-order predictions
    sa_, ca = ta.eval_dataset(dsa)
    sb_, cb = tb.eval_dataset(dsb)
    assert ca == cb == n and abs(sa_ - sb_) <= 1e-9 * abs(sa_)
    assert np.array_equal(ta.predict_dataset(dsa).view(np.uint32), tb.predict_dataset(dsb).view(np.uint32))


@pytest.mark.parametrize("knobs", [[("runs_len", 4)], [("runs_len", 7), ("runs_sets", 2)], [("runs_len", 2), ("runs_block", 256)]])
def test_runs_at_k_128(knobs):
    """k = 128: 16 lanes x 2 chunks per row, 4 runs per wave"""
    nu, ni, n = 30000, 2500, 600000
    u, i, r = cases.planted_triples(n, nu, ni, seed=33)
    a, dsa, _ = _run(u, i, r, nu, ni, [("runs_exec", 0)], k=128)
    b, dsb, _ = _run(u, i, r, nu, ni, [("runs_exec", 1)] + knobs, k=128)
    assert dsa.kind == 0 and dsb.kind == 10
    for name in NAMES:
        assert np.array_equal(a[name].view(np.uint32), b[name].view(np.uint32)), name


def test_a_user_rating_the_same_item_twice_in_a_row_and_other_repeats():
    rng = np.random.default_rng(3)
    nu, ni, n = 50, 12, 6000
    u = rng.integers(0, nu, n).astype(np.uint32)
    i = rng.integers(0, ni, n).astype(np.uint32)
    u[1::7] = u[0::7][:len(u[1::7])]     # the same (user, item) twice in a row, and the same user on consecutive ratings
    i[1::7] = i[0::7][:len(i[1::7])]
    r = rng.integers(1, 6, n).astype(np.float32)
    a, _, _ = _run(u, i, r, nu, ni, [("runs_exec", 0)], passes=3)
    b, dsb, _ = _run(u, i, r, nu, ni, [("runs_exec", 1), ("runs_len", 7)], passes=3)
    assert dsb.kind == 10
    for name in NAMES:
        assert np.array_equal(a[name].view(np.uint32), b[name].view(np.uint32)), name


def test_runs_pass_equals_the_oracle():
    from oracle import oracle
    oracle.build()
    nu, ni, n = 4000, 300, 80000
    u, i, r = cases.planted_triples(n, nu, ni, seed=12, zipf=True)
    got, ds, _ = _run(u, i, r, nu, ni, [("runs_exec", 1), ("runs_len", 4)], passes=1)
    assert ds.kind == 10
    o = oracle.OracleTrainer("port", 0, 0)
    o.seed(10)
    for kk, v in cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=64):
        o.set_param(kk, v)
    o.init_model()
    o.init_trainer()
    o.update_batch(sa.CSRData.from_triples(u, i, r))
    for name in NAMES:
        assert np.array_equal(got[name].view(np.uint32), o.view(name).view(np.uint32)), name


def test_ids_out_of_range_and_other_configurations():
    t = _trainer(100, 20, [("runs_exec", 1)])
    u = np.array([1, 2, 100], np.uint32); i = np.array([1, 2, 3], np.uint32); r = np.ones(3, np.float32)
    with pytest.raises(sa.SvdfError, match="user feature index exceed bound"):
        t.dataset_from_triples(u, i, r)
    with pytest.raises(sa.SvdfError, match="item feature index exceed bound"):
        t.dataset_from_triples(np.array([1], np.uint32), np.array([20], np.uint32), np.ones(1, np.float32))
    # another width / link: the plain schedule
    t2 = sa.Trainer(0, 0)
    t2.seed(10)
    for kk, v in cases.conf_with(cases.BASICMF_CONF, num_user=100, num_item=20, num_factor=32):
        t2.set_param(kk, str(v))
    t2.init_model(); t2.init_trainer(); t2.set_knob("runs_min_rows", 0); t2.set_knob("pivot_exec", 0)
    assert t2.dataset_from_triples(u[:2], i[:2], r[:2]).kind == 0
