"""Window-minibatch step of the N-rank path on one MI355X (svdf_k_window.hip): N trainers play the N
ranks (HipShard(minibatch=True) windows, explicit sum in rank order instead of the collective) and must equal the
oracle-backed simulation of tests/multi_rank_utils.py -- every instance the reference's update_inner on (current user side,
window-start item side), item-side changes summed per item in file order -- bit for bit."""
import numpy as np
import pytest

import cases
import svdfeature_amd as sa
from multi_rank_utils import simulate
from svdfeature_amd.multi_gpu import HipShard, ShardedTrainer, shard_windows, shard_windows_parts

pytestmark = pytest.mark.gpu
NAMES = ("W_item", "i_bias", "W_user", "u_bias")


def _trainer(conf, active=0, knobs=()):
    t = sa.Trainer(0, active)
    t.seed(10)
    for k, v in conf:
        t.set_param(k, v)
    t.init_model()
    t.init_trainer()
    for k, v in knobs:
        t.set_knob(k, v)
    return t


def _run_ranks(conf, u, i, r, world, windows, passes, active=0, knobs=(), half=False, parts=1, num_item=None):
    import torch
    dev = torch.device("cuda", 0)
    ranks = []
    for rk in range(world):
        ad = HipShard(_trainer(conf, active, knobs), torch, dev, parts=parts, minibatch=True)
        ad.set_wire_half(half)
        sh = shard_windows(u, i, r, rk, world, windows) if parts == 1 else shard_windows_parts(u, i, r, rk, world, windows, num_item, parts)
        ranks.append((ad, ad.make_windows(sh)))
    for _ in range(passes):
        for w in range(windows):
            for part in range(parts):
                ds_ = []
                for ad, wins in ranks:
                    ad.train(wins[w] if parts == 1 else wins[w][part])
                    d = ad.delta_get() if parts == 1 else ad.delta_get(part)
                    ad.stream.synchronize()
                    ds_.append(d.clone())
                total = ds_[0]
                for d in ds_[1:]:
                    total = total + d
                torch.cuda.synchronize()
                for ad, _ in ranks:
                    if parts == 1:
                        ad.delta_set(total)
                    else:
                        ad.delta_set(total, part)
    for ad, _ in ranks:
        ad.t.synchronize()
    return [ad for ad, _ in ranks]


def _check(ranks, sim):
    for ad, s in zip(ranks, sim):
        for name in NAMES:
            a, b = ad.t.view(name), s.t.view(name)
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), name


@pytest.mark.parametrize("k,world,knobs", [(16, 2, ()), (64, 3, ()), (64, 3, (("window_slots", 0),)), (64, 2, (("window_groups", 2),)),
                                           (128, 2, ()), (128, 2, (("window_groups", 2),)), (100, 2, ()), (7, 4, ()), (256, 2, ())])
def test_simulated_ranks_equal_the_oracle_simulation(k, world, knobs):
    """fp32 wire: bit for bit, every kernel variant (lane-group kernel at any width, 8 / 16-lane slot kernels at k = 64 / 128 with
    one or two user sets per wave), users with 1 ... ~40 instances per window, items without any."""
    nu, ni, n = 1500, 700, 60000
    u, i, r = cases.planted_triples(n, nu, ni, seed=k + world)
    u[:3000] = u[:3000] % 7          # a few heavy users: long sequential walks, ragged waves
    i[i == 5] = 6                    # an item nobody rates
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=k)
    ranks = _run_ranks(conf, u, i, r, world, 4, 2, knobs=knobs)
    _check(ranks, simulate(conf, u, i, r, world, 4, 2, minibatch=True))
    assert ranks[0].make_windows([(u[:10], i[:10], r[:10])])[0].kind == 5


@pytest.mark.parametrize("active,extra", [(2, (("base_score", "0.5"),)), (0, (("reg_method", "1"),)), (0, (("reg_method", "2"), ("wd_user", "0.5"), ("wd_item", "0.5"))),
                                          (0, (("no_user_bias", "1"),)), (0, (("user_nonnegative", "1"),)), (0, (("up:wd", "0.1"), ("up:bound", "100"), ("up:wd", "0.002"), ("up:bound", "100000")))])
def test_other_links_and_regularisers(active, extra):
    nu, ni, n = 800, 300, 30000
    u, i, r = cases.planted_triples(n, nu, ni, seed=3)
    if active == 2:
        r = (r > 3).astype(np.float32)
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=24) + list(extra)
    ranks = _run_ranks(conf, u, i, r, 2, 3, 2, active=active)
    _check(ranks, simulate(conf, u, i, r, 2, 3, 2, active=active, minibatch=True))


def test_sharded_trainer_with_one_rank_and_empty_windows():
    """multi_gpu.ShardedTrainer over HipShard(minibatch=True) with one rank (the exchange step runs without a collective), a
    window without instances in the middle."""
    import torch
    nu, ni, n = 900, 250, 20000
    u, i, r = cases.planted_triples(n, nu, ni, seed=8)
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=64)
    ad = HipShard(_trainer(conf), torch, torch.device("cuda", 0), minibatch=True)
    ad.set_wire_half(False)
    sh = shard_windows(u, i, r, 0, 1, 3)
    sh.insert(1, (u[:0], i[:0], r[:0]))
    st = ShardedTrainer(ad, ad.make_windows(sh), 1, None)
    for _ in range(2):
        st.train_pass()
    ad.t.synchronize()
    sim = simulate(conf, u, i, r, 1, 3, 2, minibatch=True)
    _check([ad], sim)


def test_fp16_wire_stays_close_and_item_range_pieces_are_exact():
    """(i) fp16 wire format: the deltas are rounded once per window -- parameters within 1e-3 relative of the fp32-wire run;
    (ii) item-range pieces (svdf_item_delta_select): piece by piece == the simulation of the piece-wise schedule."""
    nu, ni, n = 1200, 400, 50000
    u, i, r = cases.planted_triples(n, nu, ni, seed=21)
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=64)
    full = _run_ranks(conf, u, i, r, 2, 4, 2)
    half = _run_ranks(conf, u, i, r, 2, 4, 2, half=True)
    a, b = full[0].t.view("W_item"), half[0].t.view("W_item")
    assert np.abs(a - b).max() <= 1e-3 * np.abs(a).max() and not np.array_equal(a, b)
    from multi_rank_utils import simulate_parts
    pieces = _run_ranks(conf, u, i, r, 2, 4, 2, parts=2, num_item=ni)
    _check(pieces, simulate_parts(conf, u, i, r, 2, 4, 2, 2, ni, minibatch=True))


def test_refused_configurations():
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=50, num_item=20, num_factor=8)
    t = _trainer(conf + [("reg_method", "4")])
    u, i, r = cases.planted_triples(100, 50, 20, seed=1)
    with pytest.raises(sa.SvdfError, match="window data sets"):
        t.dataset_window_from_triples(u, i, r)
    t = _trainer(conf)
    with pytest.raises(sa.SvdfError, match="item feature index exceed bound"):
        t.dataset_window_from_triples(u[:3], np.array([20, 0, 1], np.uint32), r[:3])
    ds = t.dataset_window_from_triples(u, i, r)
    import torch
    buf = torch.zeros(20 * 9, device="cuda")
    with pytest.raises(sa.SvdfError, match="train this window data set first"):
        t.window_delta_pack(ds, buf.data_ptr())


@pytest.mark.parametrize("k,world", [(16, 2), (128, 3), (100, 2)])
def test_rank_pairs_on_simulated_ranks_equal_the_oracle_simulation(k, world):
    """BASELINE configs[4] shape through the window-minibatch step: rank pairs (user, positive, negative) = two signed item entries per
    instance, sigmoid rank loss (glibc's expf restated on the device), no user bias; two contribution slots per pair, summed per item
    in file order -- N simulated ranks == the oracle simulation, bit for bit."""
    import torch
    from svdfeature_amd.multi_gpu import Pairs, shard_pair_windows
    nu, ni, n, windows, passes = 1200, 300, 40000, 4, 2
    u, p, q = cases.planted_pairs(n, nu, ni, seed=k)
    conf = cases.conf_with(cases.PAIR_CONF, num_user=nu, num_item=ni, num_factor=k, learning_rate=0.05, ui_init_sigma=0.1)
    dev = torch.device("cuda", 0)
    ranks = []
    for rk in range(world):
        ad = HipShard(_trainer(conf, 3), torch, dev, minibatch=True)
        ad.set_wire_half(False)
        ranks.append((ad, ad.make_windows(shard_pair_windows(u, p, q, rk, world, windows))))
    assert ranks[0][1][0].kind == 5
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
    sim = simulate(conf, Pairs(u, p, q), None, None, world, windows, passes, active=3, minibatch=True)
    for (ad, _), s in zip(ranks, sim):
        ad.t.synchronize()
        for name in ("W_item", "i_bias", "W_user"):
            assert np.array_equal(ad.t.view(name).view(np.uint32), s.t.view(name).view(np.uint32)), name
    with pytest.raises(sa.SvdfError, match="must differ"):
        ranks[0][0].t.dataset_window_from_pairs(u[:3], p[:3], p[:3])


@pytest.mark.parametrize("k,world,chunks,ni", [(64, 3, 2, 701), (16, 2, 1, 701), (128, 4, 2, 701), (16, 4, 1, 3)])   # last: fewer items than blocks
def test_stratified_schedule_on_simulated_ranks_equals_the_simulation(k, world, chunks, ni):
    """multi_gpu.StratifiedTrainer's schedule with N trainers on one GPU (block hand-overs = device copies): in-place per-item sums into
    the owned item block (svdf_window_delta_apply_local), svdf_item_block_get / _set -- == the oracle simulation bit for bit"""
    import torch
    from multi_rank_utils import simulate_stratified
    from svdfeature_amd.multi_gpu import StratifiedTrainer, stratified_plan
    nu, n, passes, per_item = 1500, 60000 if ni > 100 else 3000, 2, 9.0
    u, i, r = cases.planted_triples(n, nu, ni, seed=k + world)
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=k)
    dev = torch.device("cuda", 0)
    ranks = []
    for rk in range(world):
        ad = HipShard(_trainer(conf), torch, dev, minibatch=True)
        ad.set_wire_half(False)
        plan = [[ad.make_windows(sub) for sub in chunk] for chunk in stratified_plan(u, i, r, rk, world, chunks, ni, per_item)]
        ranks.append((ad, plan))
    for _ in range(passes):
        for c in range(chunks):
            for s in range(world):
                for rk, (ad, plan) in enumerate(ranks):
                    for w in plan[c][s]:
                        ad.train(w)
                        ad.apply_local(w, (rk + s) % world, world)
                outs = []
                for rk, (ad, _) in enumerate(ranks):
                    blk = ad.block_get((rk + s) % world, world)
                    ad.stream.synchronize()          # the copy-out kernel runs on the adaptor's stream, clone() on torch's
                    outs.append(blk.clone())
                torch.cuda.synchronize()
                for rk, (ad, _) in enumerate(ranks):
                    ad.block_set((rk + s + 1) % world, world, outs[(rk + 1) % world])
    for b in range(world):
        blk = ranks[b][0].block_get(b, world)
        ranks[b][0].stream.synchronize()
        blk = blk.clone()
        for rk, (ad, _) in enumerate(ranks):
            if rk != b:
                ad.block_set(b, world, blk)
    sim = simulate_stratified(conf, u, i, r, world, chunks, passes, ni, per_item)
    for (ad, _), s_ in zip(ranks, sim):
        ad.t.synchronize()
        for name in NAMES:
            assert np.array_equal(ad.t.view(name).view(np.uint32), s_.t.view(name).view(np.uint32)), name
    # one rank through the trainer class itself (no process group)
    ad = HipShard(_trainer(conf), torch, dev, minibatch=True)
    plan = [[ad.make_windows(sub) for sub in chunk] for chunk in stratified_plan(u, i, r, 0, 1, chunks, ni, per_item)]
    st = StratifiedTrainer(ad, plan, 1, 0, None)
    for _ in range(passes):
        st.train_pass()
    ad.t.synchronize()
    one = simulate_stratified(conf, u, i, r, 1, chunks, passes, ni, per_item)
    for name in NAMES:
        assert np.array_equal(ad.t.view(name).view(np.uint32), one[0].t.view(name).view(np.uint32)), name


@pytest.mark.parametrize("shape", ["one_user", "one_item", "one_instance", "few_heavy_users", "every_user_once"])
@pytest.mark.parametrize("k", [10, 64])
def test_degenerate_window_shapes(shape, k):
    """shapes at the edges of the window layout: ONE user holding every instance of the window (one lane group walks thousands of
    instances), ONE item receiving every contribution (one per-item sum over thousands of slots), a single instance, three users with
    most of the mass next to hundreds with one instance, every user exactly once -- 2 simulated ranks == the oracle simulation"""
    rng = np.random.default_rng(5)
    nu, ni, n = 600, 90, 6000
    u, i, r = cases.planted_triples(n, nu, ni, seed=21)
    if shape == "one_user":
        u[:] = 7
    elif shape == "one_item":
        i[:] = 11
    elif shape == "one_instance":
        u, i, r = u[:1], i[:1], r[:1]
    elif shape == "few_heavy_users":
        heavy = rng.random(n) < 0.9
        u[heavy] = rng.choice(np.array([3, 4, 500], np.uint32), int(heavy.sum()))
    else:
        u, i, r = np.arange(nu, dtype=np.uint32), i[:nu], r[:nu]
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=k)
    windows = 1 if len(r) < 10 else 3
    ranks = _run_ranks(conf, u, i, r, 2, windows, 2)
    _check(ranks, simulate(conf, u, i, r, 2, windows, 2, minibatch=True))


def test_apply_local_refuses_a_window_with_items_outside_the_active_block():
    """stratified schedule: a window whose instances touch items outside the selected item block would silently lose those updates"""
    nu, ni = 100, 40
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=16)
    t = _trainer(conf)
    u, i, r = cases.planted_triples(500, nu, ni, seed=1)
    ds = t.dataset_window_from_triples(u, i, r)          # items of every block
    t.train_dataset(ds)
    t.item_delta_select(1, 4)
    with pytest.raises(sa.SvdfError, match="outside the active item block"):
        t.window_delta_apply_local(ds)
    t.item_delta_select(0, 1)
    t.window_delta_apply_local(ds)                        # the whole range: fine


@pytest.mark.parametrize("k,ni,contrib", [(16, 9, "fp32"), (64, 9, "bf16"), (128, 9, "fp32"), (200, 9, "fp32"), (256, 5, "bf16"), (16, 300, "fp32"), (8, 500, "fp32"),
                                          (64, 30000, "fp32"), (128, 30000, "bf16"), (16, 30000, "fp32"), (256, 30000, "fp32"),
                                          (64, 2000, "fp32"), (128, 2000, "bf16"), (16, 2000, "fp32"), (200, 2000, "fp32"), (8, 1000, "fp32")])
def test_long_contribution_lists_are_summed_by_the_whole_workgroup_in_slot_order(k, ni, contrib):
    """items that meet far more than 16 contributions per window (a Zipf-popular catalogue): k_window_items queues such lists and the whole workgroup
    loads their slots into LDS, one lane group adds them in slot order -- lists of ~600 slots (several LDS chunks) at every width, and more long lists
    than a workgroup's queue holds (ni = 300 / 500 at narrow widths: the rest take the lane-group path), and sparse windows (ni = 30 000: a thread per
    item finds the few items with slots, the lane groups share them).  Wire buffers of simulated ranks AND the
    in-place sums of a one-GPU window sequence, fp32 and bf16 contribution rows, against the oracle simulation bit for bit."""
    import multi_rank_utils
    nu, n, windows = 2000, 24000, 4
    u, i, r = cases.planted_triples(n, nu, ni, seed=k + ni)
    if ni == 1000:    # a hundred neighbouring items with ~40 slots each among 6 per item on average: more long lists than one workgroup's queue holds
        i[: 2 * n // 3] = i[: 2 * n // 3] % 100
    if ni >= 2000:    # lists are LONG relative to their window's mean (more than 16 slots and more than 4 x the mean: dense windows of uniform data keep the lane-group loops);
                      # ni = 2 000: the scan kernel's cooperative form (3 slots per item on average, two hot items); ni = 30 000: a SPARSE window (6 000 ratings over 30 000 items: the in-place sums take k_window_items_sparse) with two hot items (2 000 / 850 slots per window)
        i[::3] = 7
        i[1::7] = 11
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=k, learning_rate=0.0005)
    multi_rank_utils.CONTRIB_BF16 = contrib == "bf16"
    try:
        sim2 = simulate(conf, u, i, r, 2, windows, 2, minibatch=True)
        sim1 = simulate(conf, u, i, r, 1, windows, 2, minibatch=True)
    finally:
        multi_rank_utils.CONTRIB_BF16 = False
    conf_c = conf + [("amd:contrib", contrib)]
    _check(_run_ranks(conf_c, u, i, r, 2, windows, 2), sim2)
    t = _trainer(conf_c + [("amd:step", "minibatch"), ("amd:window", str(n // windows))])
    t.set_knob("window_hot_sub", 0)   # (round 6: lists of more than 128 slots would otherwise move in ordered sub-steps -- tests/test_gpu_window_hot.py; here: the stale sums)
    ds = t.dataset_from_triples(u, i, r)
    assert ds.kind == 8 and ds.num_batches == windows
    for _ in range(2):
        t.train_dataset(ds)
    t.synchronize()
    for name in NAMES:
        assert np.array_equal(t.view(name).view(np.uint32), sim1[0].t.view(name).view(np.uint32)), name
