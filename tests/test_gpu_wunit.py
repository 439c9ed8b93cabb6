"""Window-minibatch step for USER UNITS on one MI355X (svdf_k_wunit.hip): user-group (SVD++) blocks and rows with
global features.  N trainers play the N ranks (HipShard(minibatch=True) windows, explicit sum in rank order instead of the collective) and
must equal the oracle-backed simulation of tests/multi_rank_utils.py -- every block the reference's SVDPPFeature::update
(/root/reference/solvers/base-solver/apex_svd_base.h:568-582) / every row its update_inner (:456-462) on (the user's private state, the
window-start shared rows), shared-row changes summed per row in file order (oracle/svdf_oracle.c: svdo_update_block_stale, pinned to the
compiled reference in tests/test_window_blocks.py) -- bit for bit.  Plus the one-GPU opt-in `amd:step = minibatch` (window sequences)."""
import numpy as np
import pytest

import cases
import svdfeature_amd as sa
from multi_rank_utils import simulate
from svdfeature_amd import BlockArrays, CSRData
from svdfeature_amd.multi_gpu import HipShard, shard_block_windows, shard_csr_windows
from test_window_blocks import SVDPP_EXTRA, _blocks_with_globals

pytestmark = pytest.mark.gpu


def _trainer(conf, fmt=0, active=0, extra=(), knobs=()):
    t = sa.Trainer(fmt, active)
    t.seed(10)
    for k, v in list(conf) + list(extra):
        t.set_param(k, str(v))
    t.init_model()
    t.init_trainer()
    for k, v in knobs:
        t.set_knob(k, v)
    return t


def _run_ranks(conf, data, world, windows, passes, fmt, active=0, half=False, knobs=()):
    import torch
    dev = torch.device("cuda", 0)
    ranks = []
    for rk in range(world):
        ad = HipShard(_trainer(conf, fmt, active, knobs=knobs), torch, dev, minibatch=True)
        ad.set_wire_half(half)
        sh = shard_block_windows(data, rk, world, windows) if isinstance(data, BlockArrays) else shard_csr_windows(data, rk, world, windows)
        ranks.append((ad, ad.make_windows(sh)))
    for _ in range(passes):
        for w in range(windows):
            ds_ = []
            for ad, wins in ranks:
                ad.train(wins[w])
                d = ad.delta_get()
                ad.stream.synchronize()
                ds_.append(d.clone())
            total = ds_[0]
            for d in ds_[1:]:
                total = total + d
            torch.cuda.synchronize()
            for ad, _ in ranks:
                ad.delta_set(total)
    for ad, _ in ranks:
        ad.t.synchronize()
    return [ad for ad, _ in ranks], ranks[0][1]


def _check(ranks, sim, names):
    for ad, s in zip(ranks, sim):
        for name in names:
            a, b = ad.t.view(name), s.t.view(name)
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), name


SVDPP_NAMES = ("W_item", "i_bias", "W_ufeedback", "ufeedback_bias", "W_user", "u_bias")


@pytest.mark.parametrize("k,world,windows,knobs", [(16, 1, 3, ()), (16, 2, 3, ()), (64, 3, 2, ()), (128, 2, 2, ()), (100, 2, 3, ()), (7, 4, 2, ()), (256, 2, 2, ()),
                                                   (64, 2, 2, (("wunit_fast", 0),)), (128, 1, 2, (("wunit_fast", 0),)), (64, 2, 2, (("wunit_fast", 1),)), (128, 2, 2, (("wunit_fast", 1),)),
                                                   (192, 2, 2, ())])
def test_user_group_blocks_on_simulated_ranks_equal_the_oracle_simulation(k, world, windows, knobs):
    """SVD++ blocks: DEFAULT blocks and START / MIDDLE / END spans, users without feedback, users with several blocks in one window; the
    lane-group kernel at every width, the slot kernel with its row ring at k = 64 / 128 (k_wunit_fast, knob wunit_fast = 1) and one wave per
    unit at k = 64 / 128 / 192 / 256 (k_wunit_wave, the default where it applies)"""
    nu, ni = 260, 90
    blocks = cases.user_blocks(300, nu, ni, ni, seed=k + world, max_rows=9, max_fb=6, split_every=4)
    blocks += cases.user_blocks(120, nu, ni, ni, seed=k + world + 50, max_rows=4, max_fb=3)   # the same users again: several segments per unit
    ba = BlockArrays.from_blocks(blocks)
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=k, num_ufeedback=ni) + SVDPP_EXTRA
    ranks, wins = _run_ranks(conf, ba, world, windows, 2, fmt=1, knobs=knobs)
    assert wins[0].kind == 7
    _check(ranks, simulate(conf, ba, None, None, world, windows, 2, fmt=1, minibatch=True), SVDPP_NAMES)


@pytest.mark.parametrize("k,active,extra", [(64, 2, (("base_score", "0.5"),)), (128, 0, (("reg_method", "1"),)), (64, 0, (("reg_method", "3"),)), (128, 0, (("no_user_bias", "1"),)),
                                            (64, 0, (("user_nonnegative", "1"),)), (128, 0, (("scale_lr_ufeedback", "0.5"), ("wd_ufeedback_bias", "0.01"))),
                                            (64, 0, (("ip:wd", "0.1"), ("ip:bound", "30"), ("ip:wd", "0.002"), ("ip:bound", "100000")))])
@pytest.mark.parametrize("fast", [1, 2])
def test_slot_kernel_links_and_regularisers(k, active, extra, fast):
    """the configurations k_wunit_fast (fast = 1) and k_wunit_wave (fast = 2) cover beyond the usual one (long units: up to 40 rows and 30
    feedback ids per user)"""
    nu, ni = 150, 120
    ba = BlockArrays.from_blocks(cases.user_blocks(140, nu, ni, ni, seed=k + active, max_rows=40, max_fb=30, split_every=6, binary_label=(active == 2)))
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=k, num_ufeedback=ni) + SVDPP_EXTRA + list(extra)
    ranks, _ = _run_ranks(conf, ba, 2, 2, 2, fmt=1, active=active, knobs=(("wunit_fast", fast),))
    _check(ranks, simulate(conf, ba, None, None, 2, 2, 2, fmt=1, active=active, minibatch=True), SVDPP_NAMES)


@pytest.mark.parametrize("k,fast", [(64, 2), (128, 2), (256, 2), (128, 1)])
def test_units_longer_than_one_record_block(k, fast):
    """units of up to 200 rows and 170 feedback ids: the wave kernel reads records and feedback entries 64 at a time and fetches item rows eight
    ahead -- the block boundaries, the partial last block and the partial last group"""
    nu, ni = 40, 300
    ba = BlockArrays.from_blocks(cases.user_blocks(60, nu, ni, ni, seed=k + 7, max_rows=200, max_fb=170, split_every=5))
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=k, num_ufeedback=ni) + SVDPP_EXTRA
    ranks, _ = _run_ranks(conf, ba, 2, 2, 2, fmt=1, knobs=(("wunit_fast", fast),))
    _check(ranks, simulate(conf, ba, None, None, 2, 2, 2, fmt=1, minibatch=True), SVDPP_NAMES)


@pytest.mark.parametrize("k", [64, 128])
def test_plain_rows_through_the_unit_kernels(k):
    """(user, item, rating) rows as a CSR window of a random-order trainer: the slot kernel without feedback and without global entries"""
    nu, ni, n = 700, 150, 20000
    u, i, r = cases.planted_triples(n, nu, ni, seed=k)
    u[:2000] = u[:2000] % 5   # a few long units
    d = CSRData.from_triples(u, i, r)
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=k)
    ranks, wins = _run_ranks(conf, d, 2, 3, 2, fmt=0)
    assert wins[0].kind == 7
    _check(ranks, simulate(conf, u, i, r, 2, 3, 2, minibatch=True), ("W_item", "i_bias", "W_user", "u_bias"))


@pytest.mark.parametrize("active,extra", [(2, (("base_score", "0.5"),)), (0, (("reg_method", "1"), ("reg_global", "1"))),
                                          (0, (("reg_method", "2"), ("wd_user", "0.5"), ("wd_item", "0.5"))), (0, (("no_user_bias", "1"),)),
                                          (0, (("user_nonnegative", "1"),)), (0, (("scale_lr_ufeedback", "0.5"), ("wd_ufeedback_bias", "0.01"))),
                                          (0, (("num_regfree_global", "2"), ("gp:wd", "0.1"), ("gp:bound", "3"), ("gp:wd", "0.002"), ("gp:bound", "100")))])
def test_blocks_with_global_entries_two_item_entries_links_and_regularisers(active, extra):
    nu, ni, ng = 120, 60, 7
    blocks = _blocks_with_globals(200, nu, ni, ng, seed=11 + active)
    if active == 2:
        for b in blocks:
            b.data.row_label[:] = (b.data.row_label > 3).astype(np.float32)
    ba = BlockArrays.from_blocks(blocks)
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=20, num_global=ng, num_ufeedback=ni, wd_global="0.002") + SVDPP_EXTRA + list(extra)
    ranks, _ = _run_ranks(conf, ba, 2, 3, 2, fmt=1, active=active)
    _check(ranks, simulate(conf, ba, None, None, 2, 3, 2, fmt=1, active=active, minibatch=True), SVDPP_NAMES + ("g_bias",))


def _rows_with_globals(n, nu, ni, ng, per_row, seed, fixed=True):
    rng = np.random.default_rng(seed)
    rows = []
    for _ in range(n):
        m = per_row if fixed else int(rng.integers(0, per_row + 1))
        gl = [(int(g), float(rng.uniform(0.1, 1.0))) for g in sorted(rng.choice(ng, size=m, replace=False))]
        items = [(int(rng.integers(0, ni)), 1.0)]
        if not fixed and rng.random() < 0.3:
            x = int(rng.integers(0, ni))
            if x != items[0][0]:
                items = sorted(items + [(x, -1.0)])
        rows.append((float(rng.integers(1, 6)), gl, [(int(rng.integers(0, nu)), 1.0 if fixed else float(rng.choice([1.0, 0.5])))], items))
    return CSRData.from_rows(rows)


@pytest.mark.parametrize("k,world,fixed,knobs", [(16, 1, True, ()), (128, 2, True, ()), (64, 2, True, ()), (128, 2, True, (("wunit_fast", 0),)), (64, 3, False, ()), (10, 2, False, ())])
def test_rows_with_global_features_on_simulated_ranks_equal_the_oracle_simulation(k, world, fixed, knobs):
    """the neighbourhood shape (4 global entries + user + item, fixed layout) and ragged rows (0..4 global entries, sometimes two item entries,
    user values != 1) on a random-order trainer"""
    nu, ni, ng, n = 500, 120, 40, 9000
    d = _rows_with_globals(n, nu, ni, ng, 4, seed=k + world, fixed=fixed)
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=k, num_global=ng, wd_global="0.001")
    ranks, wins = _run_ranks(conf, d, world, 4, 2, fmt=0, knobs=knobs)
    assert wins[0].kind == 7
    _check(ranks, simulate(conf, d, None, None, world, 4, 2, minibatch=True), ("W_item", "i_bias", "g_bias", "W_user", "u_bias"))


def test_fp16_wire_stays_within_the_rounding_of_the_wire_format():
    nu, ni = 200, 80
    ba = BlockArrays.from_blocks(cases.user_blocks(250, nu, ni, ni, seed=21, max_rows=8, max_fb=5))
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=32, num_ufeedback=ni) + SVDPP_EXTRA
    ranks, _ = _run_ranks(conf, ba, 2, 3, 2, fmt=1, half=True)
    sim = simulate(conf, ba, None, None, 2, 3, 2, fmt=1, minibatch=True)
    for name in ("W_item", "W_ufeedback", "i_bias"):
        np.testing.assert_allclose(ranks[0].t.view(name), sim[0].t.view(name), rtol=0, atol=2e-4)


@pytest.mark.parametrize("shape", ["blocks", "rows", "triples", "pairs"])
def test_one_gpu_opt_in_minibatch_step_is_the_one_rank_simulation(shape):
    """`amd:step = minibatch` on a single-GPU handle (opt-in, NOT the reference's semantics): resident data sets become window sequences
    (kind 8), one pass = per window the users' exact walks + the per-row sums added in place; equals the one-rank oracle simulation with
    the same window cuts bit for bit.  Without the key the same calls build the exact level-scheduled data sets."""
    if shape == "blocks":
        nu, ni, windows = 200, 70, 4
        ba = BlockArrays.from_blocks(cases.user_blocks(320, nu, ni, ni, seed=4, max_rows=6, max_fb=5, split_every=5))
        conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=24, num_ufeedback=ni) + SVDPP_EXTRA
        t = _trainer(conf, 1, 0, [("amd:step", "minibatch"), ("amd:window", -(-ba.num_row // windows))])
        ds = t.dataset_from_blocks(ba)
        from svdfeature_amd.multi_gpu import block_window_bounds
        sim = simulate(conf, ba, None, None, 1, ds.num_batches, 2, fmt=1, minibatch=True)
        names = SVDPP_NAMES
        exact = _trainer(conf, 1, 0).dataset_from_blocks(ba)
    elif shape == "rows":
        nu, ni, ng, n, windows = 400, 100, 30, 8000, 5
        d = _rows_with_globals(n, nu, ni, ng, 4, seed=6)
        conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=32, num_global=ng, wd_global="0.001")
        t = _trainer(conf, 0, 0, [("amd:step", "minibatch"), ("amd:window", n // windows)])
        ds = t.dataset_from_csr(d)
        sim = simulate(conf, d, None, None, 1, windows, 2, minibatch=True)
        names = ("W_item", "i_bias", "g_bias", "W_user", "u_bias")
        exact = _trainer(conf, 0, 0).dataset_from_csr(d)
    elif shape == "pairs":
        from svdfeature_amd.multi_gpu import Pairs
        nu, ni, n, windows = 600, 150, 24000, 4
        pu, pp, pq = cases.planted_pairs(n, nu, ni, seed=3)
        conf = cases.conf_with(cases.PAIR_CONF, num_user=nu, num_item=ni, num_factor=128, learning_rate=0.05, ui_init_sigma=0.1)
        t = _trainer(conf, 0, 3, [("amd:step", "minibatch"), ("amd:window", n // windows)])
        ds = t.dataset_from_pairs(pu, pp, pq)
        sim = simulate(conf, Pairs(pu, pp, pq), None, None, 1, windows, 2, active=3, minibatch=True)
        names = ("W_item", "i_bias", "W_user")
        exact = _trainer(conf, 0, 3).dataset_from_pairs(pu, pp, pq)
    else:
        nu, ni, n, windows = 900, 200, 30000, 6
        u, i, r = cases.planted_triples(n, nu, ni, seed=12)
        conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=64)
        t = _trainer(conf, 0, 0, [("amd:step", "minibatch"), ("amd:window", n // windows)])
        ds = t.dataset_from_triples(u, i, r)
        sim = simulate(conf, u, i, r, 1, windows, 2, minibatch=True)
        names = ("W_item", "i_bias", "W_user", "u_bias")
        exact = _trainer(conf, 0, 0).dataset_from_triples(u, i, r)
    assert ds.kind == 8 and exact.kind != 8
    for _ in range(2):
        t.train_dataset(ds)
    t.synchronize()
    for name in names:
        assert np.array_equal(t.view(name).view(np.uint32), sim[0].t.view(name).view(np.uint32)), name


@pytest.mark.parametrize("shape,k,fast,contrib,extra", [
    ("blocks", 128, 2, "fp32", ()), ("blocks", 64, 2, "bf16", ()), ("blocks", 128, 2, "fp32", (("no_user_bias", "1"),)), ("blocks", 128, 1, "bf16", ()),
    ("blocks", 64, 1, "fp32", (("no_user_bias", "1"),)), ("blocks", 24, 2, "fp32", ()), ("blocks", 40, 0, "bf16", (("reg_method", "1"),)), ("blocks", 256, 2, "fp32", ()),
    ("rows", 128, 2, "fp32", ()), ("rows", 64, 1, "bf16", ()), ("rows", 32, 2, "fp32", ()), ("rows", 128, 0, "fp32", ()), ("rows_ragged", 48, 2, "bf16", ())])
def test_single_contributions_of_a_window_are_applied_in_place_with_the_same_bits(shape, k, fast, contrib, extra):
    """one-GPU window sequences: a shared row that meets exactly one contribution in a window gets no slot -- the unit applies it where it
    computes it, with the sum kernel's operations (apply_single).  Many more items than rows per window, so most contributions are single:
    every unit kernel (lane groups, slots, one wave per unit), fp32 and bf16 contribution rows, with and without user bias == the one-rank
    oracle simulation bit for bit == the same pass with every contribution through a slot (knob wunit_inplace = 0) == the same passes with
    the feedback rows' contributions written as rows by the walk (knob wunit_defer_fb = 0; default: k_wunit_sum forms them from the segments'
    deltas against the rows it updates)"""
    import multi_rank_utils
    windows = 5
    if shape == "blocks":
        nu, ni = 300, 4000
        data = BlockArrays.from_blocks(cases.user_blocks(280, nu, ni, ni, seed=k + fast, max_rows=30, max_fb=24, split_every=6))
        conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=k, num_ufeedback=ni) + SVDPP_EXTRA + list(extra)
        fmt, names, window = 1, SVDPP_NAMES, -(-data.num_row // windows)
    else:
        nu, ni, ng, n = 900, 6000, 30, 6000
        data = _rows_with_globals(n, nu, ni, ng, 4, seed=k + fast, fixed=(shape == "rows"))
        conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=k, num_global=ng, wd_global="0.001") + list(extra)
        fmt, names, window = 0, ("W_item", "i_bias", "g_bias", "W_user", "u_bias"), n // windows
    got = []
    # (in place, deferred feedback scatter): the default, every contribution through a slot, feedback contributions as rows written by the walk
    for inplace, defer in ((1, 1), (0, 1), (1, 0), (0, 0)):
        t = _trainer(conf, fmt, 0, [("amd:step", "minibatch"), ("amd:window", window), ("amd:contrib", contrib)],
                     knobs=(("wunit_fast", fast), ("wunit_inplace", inplace), ("wunit_defer_fb", defer)))
        ds = t.dataset_from_blocks(data) if fmt == 1 else t.dataset_from_csr(data)
        assert ds.kind == 8
        for _ in range(2):
            t.train_dataset(ds)
        t.synchronize()
        got.append({name: t.view(name).copy() for name in names})
        nwin = ds.num_batches
    multi_rank_utils.CONTRIB_BF16 = contrib == "bf16"
    try:
        sim = simulate(conf, data, None, None, 1, nwin, 2, fmt=fmt, minibatch=True)
    finally:
        multi_rank_utils.CONTRIB_BF16 = False
    for name in names:
        for other in got[1:]:
            assert np.array_equal(got[0][name].view(np.uint32), other[name].view(np.uint32)), name
        assert np.array_equal(got[0][name].view(np.uint32), sim[0].t.view(name).view(np.uint32)), name


def test_malformed_windows_are_refused_with_messages():
    nu, ni = 50, 20
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=8, num_ufeedback=ni) + SVDPP_EXTRA
    t = _trainer(conf, 1, 0)
    from svdfeature_amd.data import PlusBlock, TAG_DEFAULT, TAG_START
    rows2 = CSRData.from_rows([(3.0, [], [(1, 1.0)], [(2, 1.0)]), (4.0, [], [(5, 1.0)], [(3, 1.0)])])
    with pytest.raises(sa.SvdfError, match="one user"):
        t.dataset_window_from_blocks(BlockArrays.from_blocks([PlusBlock(np.array([1], np.uint32), np.array([1.0], np.float32), rows2, TAG_DEFAULT)]))
    one = CSRData.from_rows([(3.0, [], [(1, 1.0)], [(2, 1.0)])])
    with pytest.raises(sa.SvdfError, match="START..END"):
        t.dataset_window_from_blocks(BlockArrays.from_blocks([PlusBlock(np.array([1], np.uint32), np.array([1.0], np.float32), one, TAG_START)]))
    with pytest.raises(sa.SvdfError, match="listed twice"):
        t.dataset_window_from_blocks(BlockArrays.from_blocks([PlusBlock(np.array([1, 1], np.uint32), np.array([0.5, 0.5], np.float32), one, TAG_DEFAULT)]))
    r = _trainer(cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=8, num_global=4), 0, 0)
    with pytest.raises(sa.SvdfError, match="exactly one user"):
        r.dataset_window_from_csr(CSRData.from_rows([(3.0, [], [(1, 1.0), (2, 1.0)], [(2, 1.0)])]))
    with pytest.raises(sa.SvdfError, match="listed twice"):
        r.dataset_window_from_csr(CSRData.from_rows([(3.0, [(1, 0.5), (1, 0.5)], [(1, 1.0)], [(2, 1.0)])]))


@pytest.mark.parametrize("shape,k", [("blocks", 128), ("blocks", 24), ("rows", 64), ("triples", 64), ("triples", 40)])
def test_bf16_contribution_rows_equal_the_oracle_simulation_with_the_same_rounding(shape, k):
    """`amd:contrib = bf16` (opt-in): contribution rows stored as bfloat16, sums in fp32 -- every kernel that writes or sums contributions
    (k_window_users / _slots, k_window_items, k_wunit_walk / _fast, k_wunit_sum) against the oracle simulation that rounds the same way"""
    import multi_rank_utils
    from svdfeature_amd.multi_gpu import shard_windows
    import torch
    dev = torch.device("cuda", 0)
    world, windows = 2, 3
    if shape == "blocks":
        nu, ni = 200, 80
        data = BlockArrays.from_blocks(cases.user_blocks(260, nu, ni, ni, seed=k, max_rows=12, max_fb=8, split_every=5))
        conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=k, num_ufeedback=ni) + SVDPP_EXTRA
        fmt, names = 1, SVDPP_NAMES
    elif shape == "rows":
        nu, ni, ng = 400, 100, 30
        data = _rows_with_globals(8000, nu, ni, ng, 4, seed=k)
        conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=k, num_global=ng, wd_global="0.001")
        fmt, names = 0, ("W_item", "i_bias", "g_bias", "W_user", "u_bias")
    else:
        nu, ni = 900, 200
        tri = cases.planted_triples(30000, nu, ni, seed=k)
        conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=k)
        fmt, names = 0, ("W_item", "i_bias", "W_user", "u_bias")
    ranks = []
    for rk in range(world):
        ad = HipShard(_trainer(conf, fmt, 0, [("amd:contrib", "bf16")]), torch, dev, minibatch=True)
        ad.set_wire_half(False)
        if shape == "blocks":
            sh = shard_block_windows(data, rk, world, windows)
        elif shape == "rows":
            sh = shard_csr_windows(data, rk, world, windows)
        else:
            sh = shard_windows(tri[0], tri[1], tri[2], rk, world, windows)
        ranks.append((ad, ad.make_windows(sh)))
    for _ in range(2):
        for w in range(windows):
            ds_ = []
            for ad, wins in ranks:
                ad.train(wins[w])
                d = ad.delta_get()
                ad.stream.synchronize()
                ds_.append(d.clone())
            total = ds_[0] + ds_[1]
            torch.cuda.synchronize()
            for ad, _ in ranks:
                ad.delta_set(total)
    for ad, _ in ranks:
        ad.t.synchronize()
    multi_rank_utils.CONTRIB_BF16 = True
    try:
        sim = simulate(conf, data, None, None, world, windows, 2, fmt=fmt, minibatch=True) if shape != "triples" else simulate(conf, tri[0], tri[1], tri[2], world, windows, 2, minibatch=True)
    finally:
        multi_rank_utils.CONTRIB_BF16 = False
    _check([ad for ad, _ in ranks], sim, names)


def test_window_sequences_validate_the_callers_pointer_arrays():
    """amd:step = minibatch returned into the window builders before any of the pointer checks of the level path ran; a
    decreasing row_ptr / fb_ptr / block_row_ptr or a negative start must raise the usual messages instead of indexing out of bounds."""
    nu, ni, ng = 50, 30, 4
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_global=ng, num_factor=8)
    t = sa.Trainer(0, 0)
    t.seed(10)
    for k, v in conf + [("amd:step", "minibatch")]:
        t.set_param(k, str(v))
    t.init_model()
    t.init_trainer()
    d = _rows_with_globals(200, nu, ni, ng, 2, seed=1, fixed=True)
    for edit, msg in ((lambda p: p.__setitem__(4, p[5] + 3), "non-decreasing"), (lambda p: p.__isub__(p[-1] + 7), "negative")):
        bad = sa.CSRData(d.row_label, d.row_ptr.copy(), d.feat_index, d.feat_value)
        edit(bad.row_ptr)
        with pytest.raises(sa.SvdfError, match=msg):
            t.dataset_from_csr(bad)
    pconf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=8, num_ufeedback=ni)
    t = sa.Trainer(1, 0)
    t.seed(10)
    for k, v in pconf + [("amd:step", "minibatch")]:
        t.set_param(k, str(v))
    t.init_model()
    t.init_trainer()
    blocks = cases.user_blocks(20, nu, ni, ni, seed=2, max_rows=4, max_fb=3)
    for field in ("fb_ptr", "block_row_ptr", "row_ptr"):
        ba = sa.BlockArrays.from_blocks(blocks)
        arr = getattr(ba, field).copy()
        arr[3] = arr[4] + 2
        setattr(ba, field, arr)
        with pytest.raises(sa.SvdfError, match="non-decreasing"):
            t.dataset_from_blocks(ba)
