"""`amd:step = auto` (svdf_dataset.cpp: Engine::auto_step): opt-in, decided per resident data set from the level schedule the
engine builds anyway -- exact conflict-free levels when they are wide enough to stream, the window-minibatch step when the data's dependency depth
binds (levels x unit latency > 2 x bytes at the streaming rate).  Default (key absent) stays exact.

The guard: on the reference's own FILE order of a rank pass (user-grouped pairs, apex_svd_data.cpp:946-965; the stream of
tools/chain_probe.py, 99 % of whose pairs form one dependency chain) the exact pass is slower than the CPU reference path; the path `auto` picks must
not be."""
import time

import numpy as np
import pytest

import cases
import svdfeature_amd as sa

pytestmark = pytest.mark.gpu


def _trainer(conf, active=0, fmt=0, extra=()):
    t = sa.Trainer(fmt, active)
    t.seed(10)
    for k, v in list(conf) + list(extra):
        t.set_param(k, str(v))
    t.init_model()
    t.init_trainer()
    return t


def _grouped_pairs(nu, ni, per_user, seed):
    rng = np.random.default_rng(seed)
    u = np.repeat(np.arange(nu, dtype=np.uint32), per_user)
    p = rng.integers(0, ni, len(u)).astype(np.uint32)
    q = ((p + 1 + rng.integers(0, ni - 1, len(u))) % ni).astype(np.uint32)
    return u, p, q


def test_wide_levels_stay_exact_and_equal_the_oracle():
    from oracle import oracle
    oracle.build()
    nu, ni, n = 200000, 40000, 3000000
    u, i, r = cases.planted_triples(n, nu, ni, seed=3)
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=64)
    t = _trainer(conf, extra=[("amd:step", "auto")])
    ds = t.dataset_from_triples(u, i, r)
    assert t.counter(16) == 1 and ds.kind in (0, 10) and t.counter(17) == ds.num_batches   # exact levels kept (as runs of an item's ratings at this size)
    assert t.counter(18) <= 2 * t.counter(19)
    t.train_dataset(ds)
    o = oracle.OracleTrainer("port", 0, 0)
    o.seed(10)
    for k, v in conf:
        o.set_param(k, v)
    o.init_model()
    o.init_trainer()
    o.update_batch(sa.CSRData.from_triples(u, i, r))
    for name in ("W_user", "W_item", "u_bias", "i_bias"):
        assert np.array_equal(t.view(name).view(np.uint32), o.view(name).view(np.uint32)), name


def test_default_is_exact_whatever_the_depth():
    nu, ni = 60, 200
    cols = _grouped_pairs(nu, ni, 300, 5)
    t = _trainer(cases.conf_with(cases.PAIR_CONF, num_user=nu, num_item=ni, num_factor=128), active=3)
    ds = t.dataset_from_pairs(*cols)
    assert ds.kind == 11 and t.counter(16) == 0      # exact: user-run units (round 6, svdf_punit.cpp); kind 2 with the knob pair_units = 0
    t.set_knob("pair_units", 0)
    assert t.dataset_from_pairs(*cols).kind == 2


def test_deep_data_takes_the_window_step_and_equals_amd_step_minibatch():
    nu, ni = 120, 400
    cols = _grouped_pairs(nu, ni, 500, 7)
    conf = cases.conf_with(cases.PAIR_CONF, num_user=nu, num_item=ni, num_factor=128)
    models = []
    for step in ("auto", "minibatch"):
        t = _trainer(conf, active=3, extra=[("amd:step", step)])
        ds = t.dataset_from_pairs(*cols)
        assert ds.kind == 8
        if step == "auto":
            assert t.counter(16) == 2 and t.counter(18) > 2 * t.counter(19) and t.counter(20) >= 1
        for _ in range(2):
            t.train_dataset(ds)
        models.append({n: t.view(n).copy() for n in ("W_user", "W_item", "i_bias")})
    for n in models[0]:
        assert np.array_equal(models[0][n].view(np.uint32), models[1][n].view(np.uint32)), n


def test_shapes_outside_the_window_step_keep_exact_levels_and_say_so():
    """lazy decay (reg_method 4) is outside the window step: the deep data set keeps its exact levels, decision 3"""
    nu, ni = 80, 60
    u, i, r = cases.planted_triples(30000, nu, ni, seed=2)
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=32, reg_method=4)
    t = _trainer(conf, extra=[("amd:step", "auto")])
    ds = t.dataset_from_triples(u, i, r)
    assert t.counter(16) == 3 and ds.kind != 8
    t.train_dataset(ds)
    # rows with global features and user-group blocks decide the same way
    from test_gpu_wunit import _rows_with_globals
    d = _rows_with_globals(20000, 300, 200, 8, 3, seed=1, fixed=True)
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=300, num_item=200, num_global=8, num_factor=32, wd_global=0.001)
    t = _trainer(conf, extra=[("amd:step", "auto")])
    ds = t.dataset_from_csr(d)
    assert t.counter(16) == 2 and ds.kind == 8
    blocks = cases.user_blocks(400, 500, 150, 150, seed=6, max_rows=20, max_fb=12)
    pconf = cases.conf_with(cases.BASICMF_CONF, num_user=500, num_item=150, num_factor=32, num_ufeedback=150, wd_ufeedback=0.004)
    t = _trainer(pconf, fmt=1, extra=[("amd:step", "auto")])
    ds = t.dataset_from_blocks(sa.BlockArrays.from_blocks(blocks))
    assert t.counter(16) == 2 and ds.kind == 8
    t.train_dataset(ds)


def test_guard_the_path_auto_picks_is_not_slower_than_the_cpu_reference():
    """tools/chain_probe.py's stream at a size the CPU path finishes in seconds: 943 users x 1 682 items (the demo's shape), 600 pairs per user in
    the generator's file order, k = 128.  CPU = the compiled reference (oracle/_ref) when it travelled, else the pinned C port; one thread."""
    from oracle import oracle
    oracle.build()
    nu, ni, per_user, k = 943, 1682, 600, 128
    cols = _grouped_pairs(nu, ni, per_user, 1)
    n = len(cols[0])
    conf = cases.conf_with(cases.PAIR_CONF, num_user=nu, num_item=ni, num_factor=k)
    kind = "reference" if oracle.have_reference() else "port"
    o = oracle.OracleTrainer(kind, 0, 3)
    o.seed(10)
    for kk, v in conf:
        o.set_param(kk, v)
    o.init_model()
    o.init_trainer()
    csr = sa.pairs_as_csr(*cols)
    t0 = time.perf_counter()
    o.update_batch(csr)
    cpu_rate = n / (time.perf_counter() - t0)
    rates = {}
    for step in (None, "auto"):
        t = _trainer(conf, active=3, extra=[("amd:step", step)] if step else [])
        ds = t.dataset_from_pairs(*cols)
        t.train_dataset(ds)
        t.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            t.train_dataset(ds)
        t.synchronize()
        rates[step] = 3 * n / (time.perf_counter() - t0)
        if step == "auto":
            assert t.counter(16) == 2, "auto must leave the exact levels on a 99 % sequential stream"
    print("pairs/s: CPU (%s, 1 thread) %.3g, exact levels %.3g, auto (window step) %.3g" % (kind, cpu_rate, rates[None], rates["auto"]))
    assert rates["auto"] >= cpu_rate, (rates, cpu_rate)


def test_a_hot_row_bounds_the_window_and_the_step_stays_finite():
    """round 5: the window rule bounded the MEAN number of updates a shared row meets per window (sum c^2 / sum c at 24); on Zipf-popular items the hot
    row then met hundreds, all computed against its window-start value, and the pass diverged to NaN at the configs[1] size.  Round 5 bounded every row
    at window_per_target_max (128) per window; round 6 applies a hot row in ordered sub-steps of window_hot_sub (128) instead and bounds it at
    window_hot_max (2 048) per window (tests/test_gpu_window_hot.py); window_hot_sub = 0 restores the round-5 rule."""
    nu, ni, n = 50000, 2000, 2000000
    u, i, r = cases.planted_triples(n, nu, ni, seed=11, zipf=True)
    top = int(np.bincount(i, minlength=ni).max())
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=64)
    e = _trainer(conf)
    de = e.dataset_from_triples(u, i, r)
    for _ in range(3):
        e.train_dataset(de)
    for sub in (128, 0):
        t = _trainer(conf, extra=[("amd:step", "minibatch")])
        t.set_knob("window_hot_sub", sub)
        ds = t.dataset_from_triples(u, i, r)
        assert ds.kind == 8 and ds.num_batches >= -(-top // (128 if sub == 0 else 2048)), (ds.num_batches, top)
        for _ in range(3):
            t.train_dataset(ds)
        for name in ("W_item", "W_user", "i_bias"):
            a, b = t.view(name), e.view(name)
            assert np.isfinite(a).all(), name
            assert np.abs(a - b).max() < 0.05, (name, float(np.abs(a - b).max()))
    # the knobs move the bounds
    t2 = _trainer(conf, extra=[("amd:step", "minibatch")])
    t2.set_knob("window_hot_sub", 0)
    t2.set_knob("window_per_target_max", 32)
    assert t2.dataset_from_triples(u, i, r).num_batches >= -(-top // 32)
    t3 = _trainer(conf, extra=[("amd:step", "minibatch")])
    t3.set_knob("window_hot_max", 256)
    assert t3.dataset_from_triples(u, i, r).num_batches >= -(-top // 256)


def test_streams_beyond_the_probe_size_are_judged_on_their_first_rows():
    """a deep stream of more than 8 M rows: auto schedules the first 2 M only (a full level schedule of a user-grouped rank pass of 200 M pairs has
    56 M levels and takes 85 s to build), extrapolates, and builds the window sequence straight away"""
    nu, ni = 943, 1682
    cols = _grouped_pairs(nu, ni, 9000, 3)
    conf = cases.conf_with(cases.PAIR_CONF, num_user=nu, num_item=ni, num_factor=128)
    t = _trainer(conf, active=3, extra=[("amd:step", "auto")])
    ds = t.dataset_from_pairs(*cols)
    n = len(cols[0])
    assert n > 8_000_000 and ds.kind == 8 and t.counter(16) == 2
    assert 0.02 * n < t.counter(17) < 1.2 * n  # extrapolated level count: user-run units since round 6 (svdf_punit.cpp: up to 24 pairs per level), ~n before
    t.train_dataset(ds)
    t.synchronize()


def test_rank_pair_passes_from_a_candidate_file_follow_the_decision_of_their_first_pass(tmp_path):
    """input_type = 2 (the reference's own generator order, re-drawn every round): the first pass goes through the level schedule and decides, the later
    passes of the same file go straight to the chosen builder; the pairs drawn are the same as without the key (same libc rand() stream)."""
    from svdfeature_amd import data as D
    src = str(tmp_path / "train.buffer")
    D.write_ugroup_buffer(src, cases.rank_blocks(600, 60, 50, 0, 901, side_user=False, max_fb=0))
    conf = [(k, v) for k, v in cases.RANK_E2E_CONF if k not in ("num_global", "wd_global")] + [("num_global", "0")]
    out = {}
    for step in (None, "auto"):
        t = _trainer(conf + ([("amd:step", step)] if step else []), active=3, fmt=1)
        kinds, rows = [], 0
        for r in range(3):
            t.set_round(r)
            ds = t.dataset_from_rank_buffer_file(src)
            kinds.append(ds.kind)
            rows += ds.info(0)
            t.train_dataset(ds)
            t.finish_round()
            ds.close()
        out[step] = (kinds, rows, {n: t.view(n).copy() for n in ("W_user", "W_item", "i_bias")}, t.counter(16))
    assert out["auto"][0] == [8, 8, 8] and out["auto"][3] == 2, out["auto"][0]
    assert 8 not in out[None][0] and out[None][1] == out["auto"][1] and out[None][1] > 1000   # the same draws either way
    for n in out[None][2]:
        a, b = out[None][2][n], out["auto"][2][n]
        assert np.isfinite(b).all() and np.abs(a - b).max() < 0.05, n


def test_user_group_passes_beyond_the_probe_size_are_judged_on_a_prefix_of_whole_users():
    """SVD++ blocks, 9 M rows: auto stages and level-schedules only the first users (>= 2 M rows, ending at a unit boundary -- one user here is split
    into START / MIDDLE / END right where the prefix would end), extrapolates the level count, and builds the window sequence without the full
    exact data set; same windows as amd:step = minibatch builds."""
    rng = np.random.default_rng(5)
    nu, ni, per = 90_000, 20_000, 100
    n = nu * per
    feat_index = np.empty(2 * n, np.uint32)
    feat_index[0::2] = np.repeat(np.arange(nu, dtype=np.uint32), per)
    feat_index[1::2] = rng.integers(0, ni, n, dtype=np.uint32)
    row_ptr = np.empty(3 * n + 1, np.int64)
    row_ptr[0::3] = 2 * np.arange(n + 1)
    row_ptr[1::3] = 2 * np.arange(n)
    row_ptr[2::3] = 2 * np.arange(n) + 1
    fbn = 20
    fb_index = (rng.integers(0, ni // fbn, (nu, fbn)) + np.arange(fbn) * (ni // fbn)).astype(np.uint32).ravel()   # distinct ids inside a list
    # user 20 000 (rows 2 000 000 .. 2 000 099: where a 2 M-row prefix would end) comes as three blocks
    cut = 20_000
    tags = np.zeros(nu + 2, np.int32)
    tags[cut], tags[cut + 1], tags[cut + 2] = 1, 3, 2   # START, MIDDLE, END (svdpp_tag: 0 default, 1 start, 2 end, 3 middle)
    brp = np.concatenate([np.arange(cut + 1) * per, [cut * per + 30, cut * per + 70], np.arange(cut + 1, nu + 1) * per]).astype(np.int64)
    fbp = np.concatenate([np.arange(cut + 1) * fbn, [(cut + 1) * fbn, (cut + 1) * fbn], np.arange(cut + 2, nu + 2) * fbn]).astype(np.int64)
    # (START carries the list, MIDDLE none, END the same list again)
    fb_full = np.concatenate([fb_index[:(cut + 1) * fbn], fb_index[cut * fbn:(cut + 1) * fbn], fb_index[(cut + 1) * fbn:]])
    blocks = sa.BlockArrays(tags, fbp, fb_full, np.full(fb_full.size, fbn ** -0.5, np.float32), brp, rng.integers(1, 6, n).astype(np.float32), row_ptr,
                            feat_index, np.ones(2 * n, np.float32))
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=64, num_ufeedback=ni, wd_ufeedback=0.004)
    t = _trainer(conf, fmt=1, extra=[("amd:step", "auto")])
    t0 = time.perf_counter()
    ds = t.dataset_from_blocks(blocks)
    dt = time.perf_counter() - t0
    assert ds.kind == 8 and t.counter(16) == 2 and ds.num_row == n
    assert 0.5 * (nu / 8) < t.counter(17) < 2.0 * nu   # extrapolated levels: a handful of users per level
    m = _trainer(conf, fmt=1, extra=[("amd:step", "minibatch")])
    t0 = time.perf_counter()
    dm = m.dataset_from_blocks(blocks)
    dt_m = time.perf_counter() - t0
    assert dm.kind == 8 and dm.num_batches == ds.num_batches
    print("9 M rows of user blocks: auto %.2f s (prefix probe + windows), minibatch %.2f s (windows only), %d windows" % (dt, dt_m, ds.num_batches))
    t.train_dataset(ds)
    m.train_dataset(dm)
    for name in ("W_user", "W_item", "W_ufeedback"):
        assert np.array_equal(t.view(name).view(np.uint32), m.view(name).view(np.uint32)), name


def test_the_default_step_warns_when_the_dependency_depth_binds(capfd):
    """the DEFAULT step stays exact whatever the data order, but it no longer does so silently -- on the reference's own
    pair order (user-grouped pairs: one dependency chain) the engine runs the `amd:step = auto` estimator on the schedule it has built anyway and says
    on stderr what the pass will cost and which key trades bit parity for the streaming rate.  Wide data sets get no line."""
    nu, ni = 600, 2000
    cols = _grouped_pairs(nu, ni, 600, 9)                      # 360 K pairs, ~99 % of them one chain
    conf = cases.conf_with(cases.PAIR_CONF, num_user=nu, num_item=ni, num_factor=128)
    t = _trainer(conf, active=3)
    capfd.readouterr()
    ds = t.dataset_from_pairs(*cols)
    err = capfd.readouterr().err
    assert ds.kind in (2, 11) and t.counter(16) == 0            # exact (user-run units since round 6), no auto decision taken
    assert t.counter(26) == 1 and t.counter(27) > 10 * t.counter(28)
    assert "default (exact) step" in err and "amd:step = auto" in err and "conflict-free levels" in err
    # the same stream under amd:step = auto: no guard line (the caller has chosen), the decision line instead
    t2 = _trainer(conf, active=3, extra=[("amd:step", "auto")])
    capfd.readouterr()
    ds2 = t2.dataset_from_pairs(*cols)
    err2 = capfd.readouterr().err
    assert ds2.kind == 8 and t2.counter(26) == 0 and "default (exact) step" not in err2 and "amd:step = auto" in err2
    # wide levels: silent
    u, i, r = cases.planted_triples(3000000, 200000, 40000, seed=3)
    t3 = _trainer(cases.conf_with(cases.BASICMF_CONF, num_user=200000, num_item=40000, num_factor=64))
    capfd.readouterr()
    t3.dataset_from_triples(u, i, r)
    assert t3.counter(26) == 0 and "default (exact) step" not in capfd.readouterr().err
