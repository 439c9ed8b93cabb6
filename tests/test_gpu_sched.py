"""Device-side level scheduling (svdf_k_sched.hip) against the host scheduler: identical conflict-free batches (same level
boundaries, same order inside every batch), hence byte-identical training results; bounds errors carry the reference's
messages; deep dependency chains and degenerate shapes drain."""
import time

import numpy as np
import pytest

import cases
import svdfeature_amd as sa
from oracle import oracle

pytestmark = pytest.mark.gpu


def _ready(fmt, act, conf, device_schedule=1, sort_batches=1):
    t = sa.Trainer(fmt, act)
    t.seed(10)
    for k, v in conf:
        t.set_param(k, v)
    t.init_model()
    t.init_trainer()
    t.set_knob("device_schedule", device_schedule)
    t.set_knob("sort_batches", sort_batches)
    return t


def _order_of(t, ds, n):
    """the data set's unit order, recovered through predictions: every instance predicts a distinct value ... simpler: the
    engine un-permutes predictions with the same order array it trained with, so predictions in file order must match."""
    return t.predict_dataset(ds)


@pytest.mark.parametrize("sort_batches", [0, 1, 2])
@pytest.mark.parametrize("n,nu,ni", [(200_000, 5000, 700), (50_000, 40, 30), (3, 10, 10), (1, 5, 5)])
def test_device_schedule_equals_host_schedule_triples(n, nu, ni, sort_batches):
    u, i, r = cases.planted_triples(n, nu, ni, seed=n % 97)
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=16)
    d, h = _ready(0, 0, conf, 1, sort_batches), _ready(0, 0, conf, 0, sort_batches)
    dd, dh = d.dataset_from_triples(u, i, r), h.dataset_from_triples(u, i, r)
    assert dd.num_batches == dh.num_batches and dd.max_batch == dh.max_batch and dd.num_row == dh.num_row == n
    for _ in range(2):
        d.train_dataset(dd)
        h.train_dataset(dh)
    for name in ("W_user", "W_item", "u_bias", "i_bias"):
        assert np.array_equal(d.view(name).view(np.uint32), h.view(name).view(np.uint32)), name
    assert np.array_equal(d.predict_dataset(dd).view(np.uint32), h.predict_dataset(dh).view(np.uint32))
    o = oracle.OracleTrainer("port", 0, 0)
    o.seed(10)
    for k, v in conf:
        o.set_param(k, v)
    o.init_model()
    o.init_trainer()
    for _ in range(2):
        o.update_batch(sa.CSRData.from_triples(u, i, r))
    for name in ("W_user", "W_item", "u_bias", "i_bias"):
        assert np.array_equal(d.view(name).view(np.uint32), o.view(name).view(np.uint32)), name


def test_device_schedule_pairs_and_a_deep_chain():
    """pairs (3 rows per unit) and a worst-case stream: every instance shares one item, so the DAG is one chain of n levels"""
    nu, ni, n = 3000, 400, 120_000
    u, p, q = cases.planted_pairs(n, nu, ni, seed=5)
    conf = cases.conf_with(cases.PAIR_CONF, num_user=nu, num_item=ni, num_factor=32, learning_rate=0.05, ui_init_sigma=0.1)
    d, h = _ready(0, 3, conf, 1), _ready(0, 3, conf, 0)
    dd, dh = d.dataset_from_pairs(u, p, q), h.dataset_from_pairs(u, p, q)
    assert dd.num_batches == dh.num_batches and dd.max_batch == dh.max_batch
    d.train_dataset(dd)
    h.train_dataset(dh)
    for name in ("W_user", "W_item", "i_bias"):
        assert np.array_equal(d.view(name).view(np.uint32), h.view(name).view(np.uint32)), name
    m = 3000
    cu = np.arange(m, dtype=np.uint32) % nu
    ci = np.zeros(m, np.uint32)
    cr = np.ones(m, np.float32)
    conf0 = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=8)
    c1, c0 = _ready(0, 0, conf0, 1), _ready(0, 0, conf0, 0)
    for c in (c1, c0):
        c.set_knob("pivot_exec", 0)   # (a row this hot would otherwise be walked as units, svdf_pivot.cpp: here the plain schedulers are compared)
    e1, e0 = c1.dataset_from_triples(cu, ci, cr), c0.dataset_from_triples(cu, ci, cr)
    assert e1.num_batches == e0.num_batches == m and e1.max_batch == 1
    c1.train_dataset(e1)
    c0.train_dataset(e0)
    assert np.array_equal(c1.view("W_item").view(np.uint32), c0.view("W_item").view(np.uint32))


def test_device_schedule_reports_the_reference_bound_errors():
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=10, num_item=8, num_factor=4)
    t = _ready(0, 0, conf)
    with pytest.raises(sa.SvdfError, match="user feature index exceed bound"):
        t.dataset_from_triples(np.array([1, 10], np.uint32), np.array([1, 2], np.uint32), np.ones(2, np.float32))
    with pytest.raises(sa.SvdfError, match="item feature index exceed bound"):
        t.dataset_from_triples(np.array([1, 2], np.uint32), np.array([1, 8], np.uint32), np.ones(2, np.float32))
    ds = t.dataset_from_triples(np.array([1, 2], np.uint32), np.array([1, 7], np.uint32), np.ones(2, np.float32))
    assert ds.num_batches == 1 and ds.num_row == 2


def test_device_schedule_build_time_at_the_contract_size():
    """BASELINE configs[1] (100 M ratings): the schedule build (upload + levels + gathers) must stay under 0.5 s -- the host
    scheduler needs 1.1 s for the levels alone -- and give the 1872-level schedule of the host path."""
    import bench
    nu, ni, n = 1_000_000, 100_000, 100_000_000
    u, i, r = bench.synth_triples(n, nu, ni)
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=64)
    t = _ready(0, 0, conf)
    t.set_knob("runs_exec", 0)   # the plain schedule first (one instance per unit); the default -- runs of an item's ratings -- below
    t.dataset_from_triples(u[:1000], i[:1000], r[:1000]).close()   # warm-up of the code path
    t0 = time.time()
    ds = t.dataset_from_triples(u, i, r)
    dt = time.time() - t0
    print("device schedule of 100M ratings: %.3f s, %d batches, largest %d" % (dt, ds.num_batches, ds.max_batch))
    assert 1800 <= ds.num_batches <= 1950 and ds.num_row == n and ds.max_batch <= ni
    assert dt < 0.5
    # round 5: the default schedule of this configuration, runs of up to 4 consecutive ratings of an item (svdf_k_runs.hip): two more radix
    # sorts, the run formation and the level schedule over ~30 M runs with 5 row slots each -- inside the same budget
    t.set_knob("runs_exec", 1)
    t0 = time.time()
    dr = t.dataset_from_triples(u, i, r)
    dtr = time.time() - t0
    print("runs schedule of the same: %.3f s, %d levels, kind %d" % (dtr, dr.num_batches, dr.kind))
    assert dr.kind == 10 and dr.num_batches < 1000 and dtr < 0.6
    dr.close()
    h = _ready(0, 0, conf, 0)
    h.set_knob("runs_exec", 0)
    t0 = time.time()
    dh = h.dataset_from_triples(u, i, r)
    print("host schedule of the same: %.3f s" % (time.time() - t0))
    assert dh.num_batches == ds.num_batches and dh.max_batch == ds.max_batch


@pytest.mark.parametrize("unit_values", [True, False])
def test_staged_windows_scheduled_on_the_device_match_the_oracle(unit_values):
    """The per-instance / per-batch update path: full staging windows of plain (user, item) instances are scheduled on the GPU
    too (one window at a time, no level state carried over); several windows, chunked calls, predictions in between -- byte for
    byte the oracle.  device_schedule_min = 1 sends even tiny windows through the device scheduler."""
    nu, ni, n = 4000, 600, 150_000
    u, i, r = cases.planted_triples(n, nu, ni, seed=12)
    d = sa.CSRData.from_triples(u, i, r)
    if not unit_values:
        rng = np.random.default_rng(3)
        d.feat_value[:] = rng.choice(np.array([1.0, 0.5, 0.25, 1.5], np.float32), size=d.feat_value.size)
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=32)
    t = _ready(0, 0, conf)
    t.set_knob("device_schedule_min", 1)
    t.set_knob("stage_window", 40_000)
    o = oracle.OracleTrainer("port", 0, 0)
    o.seed(10)
    for k, v in conf:
        o.set_param(k, v)
    o.init_model()
    o.init_trainer()
    for st in range(0, n, 7001):
        t.update_batch(d.slice_rows(st, st + 7001))
        o.update_batch(d.slice_rows(st, st + 7001))
        if st % 5 == 0:
            probe = d.slice_rows(st, st + 50)
            assert np.array_equal(t.predict_batch(probe).view(np.uint32), o.predict_batch(probe).view(np.uint32))
    t.finish_round()
    assert t.counter(3) >= 3   # several windows were flushed
    for name in ("W_user", "W_item", "u_bias", "i_bias"):
        assert np.array_equal(t.view(name).view(np.uint32), o.view(name).view(np.uint32)), name


def _fewrow_global_rows(n, nu, ni, ng, seed, absent=True):
    """<= 2 user ids, <= 2 item ids, 0..4 distinct global ids per instance, real-valued weights (neighbourhood / time-bias shape)"""
    rng = np.random.default_rng(seed)
    rows = []
    for r in range(n):
        g = [(int(x), float(np.float32(rng.uniform(-1, 1)))) for x in rng.choice(ng, int(rng.integers(0, 5)), replace=False)]
        u = [(int(x), float(np.float32(rng.choice([1.0, 0.5, 0.25])))) for x in rng.choice(nu, int(rng.integers(1, 3)), replace=False)]
        i = [(int(x), float(np.float32(rng.choice([1.0, -1.0, 0.5])))) for x in rng.choice(ni, int(rng.integers(1, 3)), replace=False)]
        if absent and r % 31 == 7:
            u = u[:1]
        rows.append((float(rng.integers(1, 6)), g, u, i))
    return sa.CSRData.from_rows(rows)


@pytest.mark.parametrize("sort_batches", [0, 1, 2])
@pytest.mark.parametrize("n,nu,ni,ng", [(60_000, 3000, 500, 64), (4000, 30, 20, 6), (2, 5, 5, 5)])
def test_device_schedule_few_row_instances_with_global_features(n, nu, ni, ng, sort_batches):
    """Rows with global features (the neighbourhood shape) scheduled on the device (dataset_fewrow_on_device) against the
    host scheduler: same batches, byte-identical parameters and predictions, both identical to the oracle."""
    d = _fewrow_global_rows(n, nu, ni, ng, seed=n % 89)
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_global=ng, num_factor=16, wd_global=0.004)
    dev, host = _ready(0, 0, conf, 1, sort_batches), _ready(0, 0, conf, 0, sort_batches)
    dd, dh = dev.dataset_from_csr(d), host.dataset_from_csr(d)
    assert dd.kind == 2 and dh.kind == 2
    assert dd.num_batches == dh.num_batches and dd.max_batch == dh.max_batch and dd.num_row == dh.num_row == n
    o = oracle.OracleTrainer("port", 0, 0)
    o.seed(10)
    for k, v in conf:
        o.set_param(k, v)
    o.init_model()
    o.init_trainer()
    for _ in range(2):
        dev.train_dataset(dd)
        host.train_dataset(dh)
        o.update_batch(d)
    for name in ("W_user", "W_item", "u_bias", "i_bias", "g_bias"):
        assert np.array_equal(dev.view(name).view(np.uint32), host.view(name).view(np.uint32)), name
        assert np.array_equal(dev.view(name).view(np.uint32), o.view(name).view(np.uint32)), name
    assert np.array_equal(dev.predict_dataset(dd).view(np.uint32), o.predict_batch(d).view(np.uint32))
    assert np.array_equal(host.predict_dataset(dh).view(np.uint32), o.predict_batch(d).view(np.uint32))


def test_device_schedule_few_row_shape_limits_fall_back_to_the_host_scheduler():
    """Five global ids on a row, or a repeated global id, do not fit the inline slots: the host scheduler takes the data set
    and the result is the oracle's either way."""
    nu, ni, ng = 200, 80, 12
    base = _fewrow_global_rows(3000, nu, ni, ng, seed=3)
    five = sa.CSRData.from_rows([(3.0, [(0, .5), (1, .5), (2, .5), (3, .5), (4, .5)], [(1, 1.0)], [(2, 1.0)])])
    twice = sa.CSRData.from_rows([(2.0, [(7, .5), (7, .25)], [(3, 1.0)], [(4, 1.0)])])
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_global=ng, num_factor=16, wd_global=0.004)
    for extra in (five, twice):
        d = sa.CSRData.concat([base, extra])
        t = _ready(0, 0, conf, 1, 1)
        ds = t.dataset_from_csr(d)
        o = oracle.OracleTrainer("port", 0, 0)
        o.seed(10)
        for k, v in conf:
            o.set_param(k, v)
        o.init_model()
        o.init_trainer()
        t.train_dataset(ds)
        o.update_batch(d)
        for name in ("W_user", "W_item", "u_bias", "i_bias", "g_bias"):
            assert np.array_equal(t.view(name).view(np.uint32), o.view(name).view(np.uint32)), name
        assert np.array_equal(t.predict_dataset(ds).view(np.uint32), o.predict_batch(d).view(np.uint32))


@pytest.mark.parametrize("k,active,extra", [(128, 0, ()), (64, 0, ()), (128, 2, (("base_score", "0.5"),)), (64, 0, (("reg_global", "1"), ("num_regfree_global", "3"))),
                                            (128, 0, (("no_user_bias", "1"),))])
def test_specialised_kernel_for_inline_global_slots_equals_k_fused_and_the_oracle(k, active, extra):
    """k_fewrow_gslots (one user id, one item id, up to four inline global ids; k = 64 / 128) against k_fused (knob fewrow_gslots = 0) and
    the oracle: identical bytes.  Rows with 0..4 global entries, so absent inline slots are exercised too."""
    rng = np.random.default_rng(k + active)
    nu, ni, ng, n = 900, 150, 40, 20000
    rows = []
    for _ in range(n):
        g = sorted(int(x) for x in rng.choice(ng, size=int(rng.integers(0, 5)), replace=False))
        rows.append((float(rng.integers(0, 2)) if active == 2 else float(rng.integers(1, 6)), [(x, float(rng.uniform(0.1, 1.0))) for x in g],
                     [(int(rng.integers(0, nu)), float(rng.choice([1.0, 0.5])))], [(int(rng.integers(0, ni)), 1.0)]))
    d = sa.CSRData.from_rows(rows)
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_global=ng, num_factor=k, wd_global=0.004) + [(a, b) for a, b in extra]
    fast, slow = _ready(0, active, conf, 1, 1), _ready(0, active, conf, 1, 1)
    slow.set_knob("fewrow_gslots", 0)
    df, dsl = fast.dataset_from_csr(d), slow.dataset_from_csr(d)
    assert df.kind == 2 and dsl.kind == 2
    o = oracle.OracleTrainer("port", 0, active)
    o.seed(10)
    for kk, v in conf:
        o.set_param(kk, v)
    o.init_model()
    o.init_trainer()
    for _ in range(2):
        fast.train_dataset(df)
        slow.train_dataset(dsl)
        o.update_batch(d)
    for name in ("W_user", "W_item", "u_bias", "i_bias", "g_bias"):
        assert np.array_equal(fast.view(name).view(np.uint32), slow.view(name).view(np.uint32)), name
        assert np.array_equal(fast.view(name).view(np.uint32), o.view(name).view(np.uint32)), name


# ---- user units (SVD++ blocks): Engine::schedule_units against device_schedule_units ---------------------------------------------
def _svdpp_conf(nu, ni, k, **kw):
    return cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=k, num_ufeedback=ni, wd_ufeedback=0.004,
                           wd_ufeedback_bias=0.002, scale_lr_ufeedback=0.7, ufeedback_init_sigma=0.01, learning_rate=0.01, **kw)


def _unit_trainers(conf, knobs=()):
    out = []
    for dev in (1, 0):
        t = sa.Trainer(1, 0)
        t.seed(21)
        for k, v in conf:
            t.set_param(k, v)
        t.init_model()
        t.init_trainer()
        t.set_knob("device_schedule", dev)
        t.set_knob("device_schedule_min", 1)
        for k, v in knobs:
            t.set_knob(k, v)
        out.append(t)
    return out


def _same_unit_schedule(blocks, conf, passes=2, knobs=()):
    d, h = _unit_trainers(conf, knobs)
    dd, dh = d.dataset_from_blocks(blocks), h.dataset_from_blocks(blocks)
    assert dd.kind == dh.kind == 3
    for what in (0, 1, 2, 5, 6, 7):   # rows, levels, widest level, units, fast-path units, digest of level_ptr / level_mid / order
        assert dd.info(what) == dh.info(what), what
    for _ in range(passes):
        d.train_dataset(dd)
        h.train_dataset(dh)
    for name in ("W_user", "W_item", "u_bias", "i_bias", "W_ufeedback", "ufeedback_bias"):
        assert np.array_equal(d.view(name).view(np.uint32), h.view(name).view(np.uint32)), name
    assert np.array_equal(d.predict_dataset(dd).view(np.uint32), h.predict_dataset(dh).view(np.uint32))
    return d, dd


@pytest.mark.parametrize("k", [16, 128])
def test_device_unit_schedule_equals_host_schedule(k):
    """split users, users without feedback, an item rated twice inside a unit (row_fresh), a feedback id listed twice and rows with a
    non-unit value (both: not on the fast path) -- same levels, same fast-path split, same order, same bits; and the oracle"""
    nu, ni = 900, 350
    blocks = cases.user_blocks(700, nu, ni, ni, seed=k, max_rows=30, max_fb=25, split_every=6)
    for b in blocks[::9]:
        if b.data.num_row >= 2 and b.extend_tag == 0:
            b.data.feat_index[3] = b.data.feat_index[1]
            if b.data.num_row >= 20:
                b.data.feat_index[2 * 19 + 1] = b.data.feat_index[1]
    for b in blocks[4::11]:
        if b.num_ufeedback >= 2 and b.extend_tag == 0:
            b.index_ufeedback[1] = b.index_ufeedback[0]
    for b in blocks[5::13]:
        if b.data.num_row >= 2:
            b.data.feat_value[2] = 0.5
    conf = _svdpp_conf(nu, ni, k)
    d, dd = _same_unit_schedule(blocks, conf)
    assert 0 < dd.num_simple_units < dd.num_units
    o = oracle.OracleTrainer("port", 1, 0)
    o.seed(21)
    for kk, v in conf:
        o.set_param(kk, v)
    o.init_model()
    o.init_trainer()
    for _ in range(2):
        for b in blocks:
            o.update_block(b)
    for name in ("W_user", "W_item", "u_bias", "i_bias", "W_ufeedback", "ufeedback_bias"):
        assert np.array_equal(d.view(name).view(np.uint32), o.view(name).view(np.uint32)), name


def test_device_unit_schedule_degenerate_shapes():
    """one user; every user the same single item (one chain of units); no feedback at all but one block; the generic path only"""
    nu, ni = 300, 40
    one = cases.user_blocks(1, nu, ni, ni, seed=3, max_rows=12, max_fb=6)
    _same_unit_schedule(one, _svdpp_conf(nu, ni, 32))
    chain = cases.user_blocks(200, nu, ni, ni, seed=4, max_rows=3, max_fb=4)
    for b in chain:
        b.data.feat_index[1::2] = 7
    d, dd = _same_unit_schedule(chain, _svdpp_conf(nu, ni, 32))
    assert dd.num_batches == len(chain)
    wide = cases.user_blocks(250, nu, 5000, 5000, seed=5, max_rows=4, max_fb=3)
    _same_unit_schedule(wide, _svdpp_conf(nu, 5000, 64))
    _same_unit_schedule(wide, _svdpp_conf(nu, 5000, 64), knobs=(("use_simple_units", 0),))


def test_device_unit_schedule_at_scale_is_fast():
    """40 K users x 100 rows (BASELINE configs[3]'s shape at the bench's old size): identical schedule, and the device build is reported"""
    rng = np.random.default_rng(11)
    nu, ni, per = 40_000, 100_000, 100
    n = nu * per
    item = rng.integers(0, ni, n, dtype=np.uint32)
    user = np.repeat(np.arange(nu, dtype=np.uint32), per)
    feat_index = np.empty(2 * n, np.uint32)
    feat_index[0::2] = user
    feat_index[1::2] = item
    row_ptr = np.empty(3 * n + 1, np.int64)
    row_ptr[0::3] = 2 * np.arange(n + 1)[: n + 1]
    row_ptr[1::3] = 2 * np.arange(n)
    row_ptr[2::3] = 2 * np.arange(n) + 1
    fbn = 100
    fb_index = np.concatenate([np.sort(rng.choice(ni, fbn, replace=False)) for _ in range(nu)]).astype(np.uint32)
    blocks = sa.BlockArrays(np.zeros(nu, np.int32), np.arange(nu + 1, dtype=np.int64) * fbn, fb_index, np.full(nu * fbn, 0.1, np.float32),
                            np.arange(nu + 1, dtype=np.int64) * per, rng.integers(1, 6, n).astype(np.float32), row_ptr, feat_index,
                            np.ones(2 * n, np.float32))
    conf = _svdpp_conf(nu, ni, 128)
    d, h = _unit_trainers(conf)
    t0 = time.perf_counter()
    dd = d.dataset_from_blocks(blocks)
    t1 = time.perf_counter()
    dh = h.dataset_from_blocks(blocks)
    t2 = time.perf_counter()
    print("dataset_from_blocks, 40 K users x 100: %.3f / %.3f s with the schedule on the device / host; the schedule itself %.1f ms (uploads included) / %.1f ms, %d levels" % (t1 - t0, t2 - t1, d.counter(24) / 1e3, h.counter(24) / 1e3, dd.num_batches))
    assert d.counter(25) == 1 and h.counter(25) == 0
    for what in (0, 1, 2, 5, 6, 7):
        assert dd.info(what) == dh.info(what), what
    assert dd.num_simple_units == dd.num_units == nu
