"""Window data sets of plain ratings / rank pairs (kind 5) regrouped ON THE DEVICE (svdf_k_wbuild.hip) against the host builder of the same engine
(knob device_window = 0): the step's result depends on the order of a user's instances and on the order of an item's contribution slots, so equal
models bit for bit = equal regrouping; tests/test_gpu_window.py additionally holds the device-built windows against the oracle-backed simulation."""
import numpy as np
import pytest

import cases
import svdfeature_amd as sa

pytestmark = pytest.mark.gpu


def _conf(nu, ni, k, pairs):
    base = cases.PAIR_CONF if pairs else cases.BASICMF_CONF
    return cases.conf_with(base, num_user=nu, num_item=ni, num_factor=k)


def _train(conf, cols, pairs, device_window, window, passes=2):
    t = sa.Trainer(0, 3 if pairs else 0)
    t.seed(10)
    for k, v in conf + [("amd:step", "minibatch"), ("amd:window", str(window))]:
        t.set_param(k, v)
    t.init_model()
    t.init_trainer()
    t.set_knob("device_window", device_window)
    ds = t.dataset_from_pairs(*cols) if pairs else t.dataset_from_triples(*cols)
    assert ds.kind == 8
    for _ in range(passes):
        t.train_dataset(ds)
    names = ("W_user", "W_item", "i_bias") + (() if pairs else ("u_bias",))
    return {n: t.view(n).copy() for n in names}, ds.num_batches


def _same(a, b):
    for n in a:
        assert np.array_equal(a[n].view(np.uint32), b[n].view(np.uint32)), n


def _skewed(n, nu, ni, seed):
    """heavy users, many users with equal counts (ties in the launch order), users and items that never occur"""
    rng = np.random.default_rng(seed)
    u = np.where(rng.random(n) < 0.3, rng.integers(0, max(nu // 50, 1), n), rng.integers(0, nu, n)).astype(np.uint32)
    i = (rng.zipf(1.3, n) % ni).astype(np.uint32)
    return u, i, rng


@pytest.mark.parametrize("k,n,nu,ni,window", [(64, 60000, 3000, 500, 7000), (128, 20000, 400, 90, 20000), (16, 5000, 5000, 5000, 999), (64, 300, 40, 30, 1)])
def test_ratings_windows_device_equals_host(k, n, nu, ni, window):
    u, i, rng = _skewed(n, nu, ni, 5)
    r = rng.integers(1, 6, n).astype(np.float32)
    conf = _conf(nu, ni, k, False)
    host, wh = _train(conf, (u, i, r), False, 0, window)
    dev, wd = _train(conf, (u, i, r), False, 1, window)
    assert wh == wd
    _same(host, dev)


@pytest.mark.parametrize("k,n,nu,ni,window", [(128, 40000, 2000, 300, 9000), (64, 8000, 100, 40, 8000), (128, 500, 60, 50, 3)])
def test_pair_windows_device_equals_host(k, n, nu, ni, window):
    u, p, rng = _skewed(n, nu, ni, 6)
    q = ((p + 1 + rng.integers(0, ni - 1, n)) % ni).astype(np.uint32)
    conf = _conf(nu, ni, k, True)
    host, wh = _train(conf, (u, p, q), True, 0, window)
    dev, wd = _train(conf, (u, p, q), True, 1, window)
    assert wh == wd
    _same(host, dev)


def test_bound_errors_come_from_the_device_builder_too():
    conf = _conf(20, 10, 64, False)
    t = sa.Trainer(0, 0)
    for k, v in conf:
        t.set_param(k, v)
    t.init_model()
    t.init_trainer()
    u = np.array([0, 1, 25], np.uint32)
    i = np.array([0, 1, 2], np.uint32)
    r = np.ones(3, np.float32)
    with pytest.raises(sa.SvdfError, match="user feature index exceed bound"):
        t.dataset_window_from_triples(u, i, r)
    with pytest.raises(sa.SvdfError, match="item feature index exceed bound"):
        t.dataset_window_from_triples(i, np.array([0, 10, 2], np.uint32), r)
    t2 = sa.Trainer(0, 3)
    for k, v in _conf(20, 10, 128, True):
        t2.set_param(k, v)
    t2.init_model()
    t2.init_trainer()
    with pytest.raises(sa.SvdfError, match="positive and negative item must differ"):
        t2.dataset_window_from_pairs(i, i, i)
    with pytest.raises(sa.SvdfError, match="item feature index exceed bound"):
        t2.dataset_window_from_pairs(i, i, np.array([5, 12, 3], np.uint32))


@pytest.mark.parametrize("shape", ["one_user", "one_item", "one_instance", "every_user_once", "two_items_pairs", "sorted_by_item"])
def test_degenerate_windows_device_equals_host(shape):
    rng = np.random.default_rng(11)
    nu, ni, n, k = 200, 60, 3000, 64
    pairs = shape == "two_items_pairs"
    if shape == "one_user":
        u = np.full(n, 7, np.uint32); i = rng.integers(0, ni, n).astype(np.uint32)
    elif shape == "one_item":
        u = rng.integers(0, nu, n).astype(np.uint32); i = np.full(n, 3, np.uint32)
    elif shape == "one_instance":
        n = 1; u = np.array([5], np.uint32); i = np.array([9], np.uint32)
    elif shape == "every_user_once":
        n = nu; u = rng.permutation(nu).astype(np.uint32); i = rng.integers(0, ni, n).astype(np.uint32)
    elif shape == "two_items_pairs":
        ni = 2; u = rng.integers(0, nu, n).astype(np.uint32); i = rng.integers(0, 2, n).astype(np.uint32)
    else:   # file order already item-major, users descending: both regroupings have to move everything
        i = np.sort(rng.integers(0, ni, n)).astype(np.uint32); u = (nu - 1 - (np.arange(n) % nu)).astype(np.uint32)
    if pairs:
        cols = (u, i, (1 - i).astype(np.uint32))
    else:
        cols = (u, i, rng.integers(1, 6, n).astype(np.float32))
    conf = _conf(nu, ni, 128 if pairs else k, pairs)
    for window in (max(n // 3, 1), n):
        host, wh = _train(conf, cols, pairs, 0, window)
        dev, wd = _train(conf, cols, pairs, 1, window)
        assert wh == wd
        _same(host, dev)
