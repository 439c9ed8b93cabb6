"""Runs of NARROW conflict-free levels inside one launch (k_fewrow_slots_chain): a rank pass in the reference's own file order
(user-grouped pairs, apex_svd_data.cpp:946-965) is tens of thousands of levels of a few dozen pairs.  The chained launch executes the same instances in
the same level order, so the model must equal the level-by-level pass (knob chain_width = 0) bit for bit -- and, through it, the oracle."""
import numpy as np
import pytest

import cases
import svdfeature_amd as sa

pytestmark = pytest.mark.gpu


def _grouped_pairs(nu, ni, per_user, seed):
    rng = np.random.default_rng(seed)
    u = np.repeat(np.arange(nu, dtype=np.uint32), per_user)
    p = rng.integers(0, ni, len(u)).astype(np.uint32)
    q = ((p + 1 + rng.integers(0, ni - 1, len(u))) % ni).astype(np.uint32)
    return u, p, q


def _run(cols, nu, ni, k, chain_width, passes=2, triples=False):
    t = sa.Trainer(0, 0 if triples else 3)
    t.seed(10)
    for kk, v in cases.conf_with(cases.BASICMF_CONF if triples else cases.PAIR_CONF, num_user=nu, num_item=ni, num_factor=k):
        t.set_param(kk, v)
    t.init_model()
    t.init_trainer()
    t.set_knob("chain_width", chain_width)
    t.set_knob("pivot_exec", 0)   # (ratings with hot rows would otherwise be walked as units: svdf_pivot.cpp)
    t.set_knob("pair_units", 0)   # (user-grouped pairs would otherwise be walked as user-run units: svdf_punit.cpp, tests/test_gpu_punit.py)
    ds = t.dataset_from_triples(*cols) if triples else t.dataset_from_pairs(*cols)
    for _ in range(passes):
        t.train_dataset(ds)
    names = ("W_user", "W_item", "i_bias") + (("u_bias",) if triples else ())
    return {n: t.view(n).copy() for n in names}, ds.num_batches, t.counter(15)


@pytest.mark.parametrize("nu,ni,per_user,width", [(60, 300, 400, 96), (200, 50, 150, 96), (30, 1000, 700, 1 << 20), (500, 400, 40, 16)])
def test_chained_levels_equal_the_level_by_level_pass(nu, ni, per_user, width):
    cols = _grouped_pairs(nu, ni, per_user, nu + per_user)
    a, levels, chained0 = _run(cols, nu, ni, 128, 0)
    b, levels_b, chained = _run(cols, nu, ni, 128, width)
    assert levels == levels_b and chained0 == 0 and chained > 0
    for n in a:
        assert np.array_equal(a[n].view(np.uint32), b[n].view(np.uint32)), n


def test_shapes_without_a_chain_form_keep_the_level_loop():
    """k = 64 rank pairs and plain ratings have no chained kernel: the knob changes nothing, nothing is counted"""
    cols = _grouped_pairs(40, 200, 100, 3)
    a, _, c0 = _run(cols, 40, 200, 64, 0)
    b, _, c1 = _run(cols, 40, 200, 64, 1 << 20)
    assert c0 == 0 and c1 == 0
    for n in a:
        assert np.array_equal(a[n].view(np.uint32), b[n].view(np.uint32)), n


def test_chained_pass_equals_the_oracle():
    from oracle import oracle
    oracle.build()
    nu, ni = 50, 120
    cols = _grouped_pairs(nu, ni, 200, 9)
    got, _, chained = _run(cols, nu, ni, 128, 96, passes=1)
    assert chained > 0
    o = oracle.OracleTrainer("port", 0, 3)
    o.seed(10)
    for kk, v in cases.conf_with(cases.PAIR_CONF, num_user=nu, num_item=ni, num_factor=128):
        o.set_param(kk, v)
    o.init_model()
    o.init_trainer()
    o.update_batch(sa.pairs_as_csr(*cols))
    for n in ("W_user", "W_item", "i_bias"):
        assert np.array_equal(got[n].view(np.uint32), o.view(n).view(np.uint32)), n


@pytest.mark.parametrize("nu,ni,n,width", [(3000, 40, 60000, 96), (500, 8, 20000, 1 << 20), (20000, 300, 200000, 16)])
def test_chained_levels_of_plain_ratings_equal_the_level_by_level_pass(nu, ni, n, width):
    """round 5: the contract kernel's chained form (k_basicmf_slots_chain): ratings over few / Zipf-popular items have a long tail of narrow levels"""
    u, i, r = cases.planted_triples(n, nu, ni, seed=nu + ni, zipf=True)
    cols = (u.astype(np.uint32), i.astype(np.uint32), r)
    a, levels, chained0 = _run(cols, nu, ni, 64, 0, triples=True)
    b, levels_b, chained = _run(cols, nu, ni, 64, width, triples=True)
    assert levels == levels_b and chained0 == 0 and chained > 0
    for n_ in a:
        assert np.array_equal(a[n_].view(np.uint32), b[n_].view(np.uint32)), n_


def test_chained_ratings_pass_equals_the_oracle():
    from oracle import oracle
    oracle.build()
    nu, ni, n = 800, 12, 15000
    u, i, r = cases.planted_triples(n, nu, ni, seed=4, zipf=True)
    got, _, chained = _run((u.astype(np.uint32), i.astype(np.uint32), r), nu, ni, 64, 96, passes=1, triples=True)
    assert chained > 0
    o = oracle.OracleTrainer("port", 0, 0)
    o.seed(10)
    for kk, v in cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=64):
        o.set_param(kk, v)
    o.init_model()
    o.init_trainer()
    o.update_batch(sa.CSRData.from_triples(u, i, r))
    for n_ in ("W_user", "W_item", "u_bias", "i_bias"):
        assert np.array_equal(got[n_].view(np.uint32), o.view(n_).view(np.uint32)), n_
