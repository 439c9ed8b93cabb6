"""User-run units of rank pairs (svdf_punit.cpp, svdf_k_wave.hip: k_pair_units; round 6).  PairwiseRankGenerator emits a user's
pairs back to back (apex_svd_data.cpp:946-965); on that order up to 24 consecutive pairs of one user with pairwise distinct items become one unit walked by
a wave with the user's row in registers, units levelled like instances.  Nothing of update_inner (apex_svd_base.h:456-462) changes: the model must equal
the level-by-level pass (knob pair_units = 0) and the oracle bit for bit; predictions come back in file order."""
import numpy as np
import pytest

import cases
import svdfeature_amd as sa

pytestmark = pytest.mark.gpu
NAMES = ("W_user", "W_item", "i_bias", "u_bias")


def _grouped_pairs(nu, ni, per_user, seed, shuffle_users=True):
    rng = np.random.default_rng(seed)
    users = rng.permutation(nu).astype(np.uint32) if shuffle_users else np.arange(nu, dtype=np.uint32)
    u = np.repeat(users, per_user)
    p = rng.integers(0, ni, len(u)).astype(np.uint32)
    q = ((p + 1 + rng.integers(0, ni - 1, len(u))) % ni).astype(np.uint32)
    return u, p, q


def _run(cols, nu, ni, k, units, active=3, extra=(), passes=2, cap=None):
    t = sa.Trainer(0, active)
    t.seed(10)
    for kk, v in cases.conf_with(cases.PAIR_CONF, num_user=nu, num_item=ni, num_factor=k) + list(extra):
        t.set_param(kk, str(v))
    t.init_model()
    t.init_trainer()
    t.set_knob("pair_units", units)
    if cap:
        t.set_knob("pair_unit_cap", cap)
    ds = t.dataset_from_pairs(*cols)
    for _ in range(passes):
        t.train_dataset(ds)
    t.synchronize()
    return {n: (None if t.view(n) is None else t.view(n).copy()) for n in NAMES}, ds, t


@pytest.mark.parametrize("nu,ni,per_user,k,extra,cap", [
    (300, 2000, 60, 128, (), None), (120, 40, 90, 128, (), None), (200, 500, 50, 64, (), 7), (80, 300, 200, 24, (("no_user_bias", "0"),), None),
    (150, 800, 33, 200, (("reg_method", "1"),), 64), (100, 600, 100, 256, (("no_user_bias", "0"), ("wd_user_bias", "0.01")), 1), (60, 50, 300, 100, (), 200)])
def test_units_equal_the_level_by_level_pass_and_the_oracle(nu, ni, per_user, k, extra, cap):
    from oracle import oracle
    oracle.build()
    cols = _grouped_pairs(nu, ni, per_user, nu + k)
    a, dsa, ta = _run(cols, nu, ni, k, 0, extra=extra)
    b, dsb, tb = _run(cols, nu, ni, k, 1, extra=extra, cap=cap)
    assert dsa.kind == 2 and dsb.kind == 11 and tb.counter(29) == 2 and ta.counter(29) == 0
    assert dsb.num_batches < dsa.num_batches or cap == 1   # fewer levels than one per dependent pair
    for n in NAMES:
        if a[n] is None:
            continue
        assert np.array_equal(a[n].view(np.uint32), b[n].view(np.uint32)), n
    o = oracle.OracleTrainer("port", 0, 3)
    o.seed(10)
    for kk, v in cases.conf_with(cases.PAIR_CONF, num_user=nu, num_item=ni, num_factor=k) + list(extra):
        o.set_param(kk, str(v))
    o.init_model()
    o.init_trainer()
    csr = sa.pairs_as_csr(*cols)
    for _ in range(2):
        o.update_batch(csr)
    for n in NAMES:
        if b[n] is None:
            continue
        assert np.array_equal(b[n].view(np.uint32), o.view(n).view(np.uint32)), n
    # scores in file order, and the evaluator
    pa, pb = ta.predict_dataset(dsa), tb.predict_dataset(dsb)
    assert np.array_equal(pa.view(np.uint32), pb.view(np.uint32))
    sa_, ca = ta.eval_dataset(dsa)
    sb_, cb = tb.eval_dataset(dsb)
    assert ca == cb == len(cols[0]) and abs(sa_ - sb_) <= 1e-9 * max(abs(sa_), 1.0)


def test_other_links_and_streams_that_are_not_user_grouped():
    nu, ni = 200, 400
    cols = _grouped_pairs(nu, ni, 40, 5)
    for active, extra in ((0, ()), (2, (("base_score", "0.5"),)), (5, ())):
        extra = tuple(extra) + (("active_type", str(active)),)
        a, dsa, _ = _run(cols, nu, ni, 64, 0, active=active, extra=extra, passes=1)
        b, dsb, _ = _run(cols, nu, ni, 64, 1, active=active, extra=extra, passes=1)
        assert dsa.kind == 2 and dsb.kind == 11
        for n in NAMES:
            if a[n] is not None:
                assert np.array_equal(a[n].view(np.uint32), b[n].view(np.uint32)), (active, n)
    # a random-order stream keeps the plain level schedule
    rng = np.random.default_rng(1)
    perm = rng.permutation(len(cols[0]))
    _, ds, t = _run(tuple(c[perm] for c in cols), nu, ni, 64, 1, passes=1)
    assert ds.kind == 2 and t.counter(29) == 0
    # configurations outside the walker too (lazy decay)
    _, ds, _ = _run(cols, nu, ni, 64, 1, extra=(("reg_method", "4"),), passes=1)
    assert ds.kind != 11


def test_bad_ids_raise_the_reference_messages():
    nu, ni = 50, 60
    u, p, q = _grouped_pairs(nu, ni, 30, 2)
    t = sa.Trainer(0, 3)
    t.seed(10)
    for kk, v in cases.conf_with(cases.PAIR_CONF, num_user=nu, num_item=ni, num_factor=64):
        t.set_param(kk, str(v))
    t.init_model()
    t.init_trainer()
    bad = p.copy(); bad[7] = ni
    with pytest.raises(sa.SvdfError, match="item feature index exceed bound"):
        t.dataset_from_pairs(u, bad, q)
    bad = u.copy(); bad[3] = nu
    with pytest.raises(sa.SvdfError, match="user feature index exceed bound"):
        t.dataset_from_pairs(bad, p, q)
    same = q.copy(); same[11] = p[11]
    with pytest.raises(sa.SvdfError, match="must differ"):
        t.dataset_from_pairs(u, p, same)


def test_staged_windows_of_grouped_pairs_take_the_unit_walker_too():
    """the per-instance route (svdf_update_csr / _batch: what the reference's CLI drives, svd_feature.cpp:220-248 with input_type = 2): a staged window whose
    rows are all user-grouped rank pairs in the generator's shape is walked as user-run units at the flush; same bits as with the knob off; windows of any
    other shape (a label that is not 1, a non-unit value) keep the level-by-level flush"""
    nu, ni = 150, 900
    cols = _grouped_pairs(nu, ni, 70, 77)
    csr = sa.pairs_as_csr(*cols)
    out = []
    for units in (0, 1):
        t = sa.Trainer(0, 3)
        t.seed(10)
        for kk, v in cases.conf_with(cases.PAIR_CONF, num_user=nu, num_item=ni, num_factor=128):
            t.set_param(kk, str(v))
        t.init_model()
        t.init_trainer()
        t.set_knob("pair_units", units)
        t.set_knob("stage_window", 4000)        # several flushes per pass
        for _ in range(2):
            t.update_batch(csr)
            t.finish_round()
        out.append(({n: t.view(n).copy() for n in ("W_user", "W_item", "i_bias")}, t.counter(29), t.counter(3)))
    assert out[0][1] == 0 and out[1][1] >= 2 and out[1][2] == out[0][2]
    for n in out[0][0]:
        assert np.array_equal(out[0][0][n].view(np.uint32), out[1][0][n].view(np.uint32)), n
    # another shape in the window: the plain flush
    d = sa.pairs_as_csr(*cols)
    d.row_label[5] = 0.0
    t = sa.Trainer(0, 3)
    t.seed(10)
    for kk, v in cases.conf_with(cases.PAIR_CONF, num_user=nu, num_item=ni, num_factor=128):
        t.set_param(kk, str(v))
    t.init_model()
    t.init_trainer()
    t.update_batch(d)
    t.finish_round()
    assert t.counter(29) == 0
