"""Window-minibatch step of the N-rank path on the CPU: the checker step against the compiled
reference, the adaptor protocol across two gloo processes, rank-count independence and the accuracy contract.  The HIP
kernels (svdf_k_window.hip) are compared with the same simulation in tests/test_gpu_window.py."""
import os
import sys

import numpy as np
import pytest

import cases
from multi_rank_utils import OracleShard, make_oracle, merged_predict, simulate
from oracle import oracle
from svdfeature_amd import CSRData
from svdfeature_amd.multi_gpu import ShardedTrainer, shard_windows
from test_multi_rank import _free_port

NU, NI = 3000, 400
CONF = cases.conf_with(cases.BASICMF_CONF, num_user=NU, num_item=NI, num_factor=16)


def _make(kind, conf, fmt=0, active=0, seed=10):
    t = oracle.OracleTrainer(kind, fmt, active)
    t.seed(seed)
    for k, v in conf:
        t.set_param(k, v)
    t.init_model()
    t.init_trainer()
    return t


@pytest.mark.skipif(not oracle.have_reference(), reason="compiled reference (oracle/_ref) not present")
@pytest.mark.parametrize("active,extra", [(0, []), (2, [("base_score", "0.5")]), (0, [("reg_method", "1")]), (0, [("reg_method", "2"), ("wd_user", "0.5"), ("wd_item", "0.5")]),
                                          (0, [("no_user_bias", "1")])])
def test_checker_step_equals_the_compiled_reference(active, extra):
    """svdo_update_csr_batch_stale of the C port == the same step driven through the reference's own classes (every row one
    ISVDTrainer::update on a trainer whose item side was put back through save_model / load_model), bit for bit: deltas,
    user side, and the untouched item side.  Rows with two item entries and a global entry included."""
    nu, ni, ng, n = 40, 15, 6, 120
    rng = np.random.default_rng(7 + active)
    rows = []
    for _ in range(n):
        items = rng.choice(ni, size=int(rng.integers(1, 3)), replace=False)
        rows.append((float(rng.integers(0, 2) if active == 2 else rng.integers(1, 6)),
                     [(int(rng.integers(0, ng)), float(rng.uniform(0.2, 1.0)))] if rng.random() < 0.5 else [],
                     [(int(rng.integers(0, nu)), 1.0)], [(int(x), float(rng.choice([1.0, -1.0, 0.5]))) for x in sorted(items)]))
    d = CSRData.from_rows(rows)
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=7, num_global=ng) + extra
    got = {}
    for kind in ("port", "reference"):
        t = _make(kind, conf, 0, active)
        init_item = t.view("W_item").copy()
        delta = t.update_batch_stale(d)
        delta = t.update_batch_stale(d, delta)     # a second window's worth accumulates into the same arrays
        np.testing.assert_array_equal(t.view("W_item"), init_item)   # the item side does not move
        got[kind] = delta + (t.view("W_user"), t.view("u_bias"), t.view("i_bias"), t.view("g_bias"))
    for a, b in zip(got["port"], got["reference"]):
        np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32))
    assert np.abs(got["port"][0]).max() > 0 and np.abs(got["port"][2]).max() > 0


def test_one_instance_per_window_is_the_sequential_reference():
    """A window of ONE instance: the stale step followed by the add IS update_inner, bit for bit up to (q + d) - q rounding --
    so compare through the definition instead: delta == (W_item after a plain update) - (W_item before)."""
    u, i, r = cases.planted_triples(300, 50, 20, seed=2)
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=50, num_item=20, num_factor=8)
    a, b = make_oracle(conf), make_oracle(conf)
    for t in range(300):
        d = CSRData.from_triples(u[t:t + 1], i[t:t + 1], r[t:t + 1])
        before = b.view("W_item").copy(), b.view("i_bias").copy()
        b.update_batch(d)
        dW, db, _ = a.update_batch_stale(d)
        np.testing.assert_array_equal(dW, b.view("W_item") - before[0])
        np.testing.assert_array_equal(db, b.view("i_bias") - before[1])
        np.testing.assert_array_equal(a.view("W_user"), b.view("W_user"))
        a.set_view("W_item", b.view("W_item"))
        a.set_view("i_bias", b.view("i_bias"))


def test_result_does_not_depend_on_the_number_of_ranks():
    """Every instance sees (its user's exact state, the window-start item side), whichever rank runs it: N ranks differ from one
    rank only by the order of the fp32 additions of the per-rank sums."""
    u, i, r = cases.planted_triples(60000, NU, NI, seed=11)
    one = simulate(CONF, u, i, r, 1, 6, 2, minibatch=True)
    for world in (2, 5):
        many = simulate(CONF, u, i, r, world, 6, 2, minibatch=True)
        np.testing.assert_allclose(many[0].t.view("W_item"), one[0].t.view("W_item"), rtol=0, atol=2e-6)
        np.testing.assert_allclose(many[0].t.view("i_bias"), one[0].t.view("i_bias"), rtol=0, atol=2e-6)
        wu = np.zeros_like(one[0].t.view("W_user"))
        for rk in range(world):
            wu[rk::world] = many[rk].t.view("W_user")[rk::world]
        np.testing.assert_allclose(wu, one[0].t.view("W_user"), rtol=0, atol=2e-6)
        for rk in range(1, world):   # the replicated side is identical on every rank
            np.testing.assert_array_equal(many[rk].t.view("W_item"), many[0].t.view("W_item"))


def _worker(rank, world, port, windows, passes, outdir):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    u, i, r = cases.planted_triples(40000, NU, NI, seed=9)
    a = OracleShard(make_oracle(CONF), torch, minibatch=True)
    wins = a.make_windows(shard_windows(u, i, r, rank, world, windows))
    st = ShardedTrainer(a, wins, world, dist)
    assert st.minibatch
    for _ in range(passes):
        st.train_pass()
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), W_item=a.t.view("W_item"), i_bias=a.t.view("i_bias"),
             W_user=a.t.view("W_user"), u_bias=a.t.view("u_bias"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_gloo_ranks_match_the_simulation_bit_for_bit(tmp_path):
    """multi_gpu.ShardedTrainer over a window-minibatch adaptor in two gloo processes == the single-process simulation."""
    import torch.multiprocessing as mp
    world, windows, passes = 2, 4, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, windows, passes, str(tmp_path)), nprocs=world, join=True)
    u, i, r = cases.planted_triples(40000, NU, NI, seed=9)
    sim = simulate(CONF, u, i, r, world, windows, passes, minibatch=True)
    for rk in range(world):
        z = np.load(str(tmp_path / ("rank%d.npz" % rk)))
        for name in ("W_item", "i_bias", "W_user", "u_bias"):
            np.testing.assert_array_equal(z[name].view(np.uint32), sim[rk].t.view(name).view(np.uint32))


def test_one_rank_runs_the_exchange_step_without_a_collective():
    """world == 1: the item side still only moves through delta_get / delta_set."""
    u, i, r = cases.planted_triples(20000, NU, NI, seed=4)
    a = OracleShard(make_oracle(CONF), minibatch=True)
    st = ShardedTrainer(a, a.make_windows(shard_windows(u, i, r, 0, 1, 5)), 1, None)
    for _ in range(2):
        st.train_pass()
    sim = simulate(CONF, u, i, r, 1, 5, 2, minibatch=True)
    for name in ("W_item", "i_bias", "W_user", "u_bias"):
        np.testing.assert_array_equal(a.t.view(name).view(np.uint32), sim[0].t.view(name).view(np.uint32))
    assert np.abs(a.t.view("W_item") - make_oracle(CONF).view("W_item")).max() > 0


@pytest.mark.parametrize("world", [2, 8])
def test_rmse_contract_of_the_window_minibatch_step(world):
    """north_star: RMSE within 1e-4 of the reference after equal epochs.  1 M ratings, 20 K x 2 K (500 ratings per item per
    pass), k = 16, 5 passes; windows by bench.py's rule for this mode: at most 32 updates per item per window, whatever the
    number of ranks (tools/minibatch_calibration.py at BASELINE configs[2] density: 32 windows 6.3e-5, 24 windows 9.2e-5)."""
    nu, ni, n = 20000, 2000, 1_000_000
    u, i, r = cases.planted_triples(n + 100_000, nu, ni, seed=5)
    tu, ti, tr = u[n:], i[n:], r[n:]
    u, i, r = u[:n], i[:n], r[:n]
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=16)
    windows = int(np.ceil(n / ni / 32.0))
    ref = cases.rmse(merged_predict(simulate(conf, u, i, r, 1, 1, 5), 1, tu, ti, tr), tr)
    got = cases.rmse(merged_predict(simulate(conf, u, i, r, world, windows, 5, minibatch=True), world, tu, ti, tr), tr)
    assert abs(got - ref) <= 1e-4


# ---- rank pairs (BASELINE configs[4]) through the same step: two item entries per instance, sigmoid rank loss, no user bias
def test_pairs_two_ranks_equal_one_rank_and_keep_the_pairwise_contract():
    """(i) window-minibatch step on rank pairs: 4 ranks == 1 rank up to the order of fp32 additions; (ii) the pairwise analogue of
    the RMSE contract (held-out pair accuracy within 3e-3, mean margin within 2 % of the sequential reference after equal passes):
    400 K pairs, 5 K x 500, k = 16, 3 passes at 10x the demo learning rate, 50 windows = 32 item updates per item per window."""
    from svdfeature_amd import pairs_as_csr
    from svdfeature_amd.multi_gpu import Pairs
    nu, ni, n = 5000, 500, 400_000
    u, p, q = cases.planted_pairs(n + 40_000, nu, ni, seed=8)
    tu, tp, tq = u[n:], p[n:], q[n:]
    u, p, q = u[:n], p[:n], q[:n]
    conf = cases.conf_with(cases.PAIR_CONF, num_user=nu, num_item=ni, num_factor=16, learning_rate=0.05, ui_init_sigma=0.1)

    def margins(ranks, world):
        out = np.zeros(len(tu), np.float32)
        for rk, a in enumerate(ranks):
            m = (tu % world) == rk
            out[m] = a.t.predict_batch(pairs_as_csr(tu[m], tp[m], tq[m]))
        return out
    windows = int(np.ceil(2.0 * n / ni / 32.0))
    seq = margins(simulate(conf, Pairs(u, p, q), None, None, 1, 1, 3, active=3), 1)
    one = simulate(conf, Pairs(u, p, q), None, None, 1, windows, 3, active=3, minibatch=True)
    four = simulate(conf, Pairs(u, p, q), None, None, 4, windows, 3, active=3, minibatch=True)
    np.testing.assert_allclose(four[0].t.view("W_item"), one[0].t.view("W_item"), rtol=0, atol=5e-6)
    par = margins(four, 4)
    assert cases.pair_accuracy(seq) > 0.85
    assert abs(cases.pair_accuracy(par) - cases.pair_accuracy(seq)) <= 3e-3
    assert abs(float(par.mean()) - float(seq.mean())) <= 0.02 * abs(float(seq.mean()))


# ---- STRATIFIED schedule (multi_gpu.StratifiedTrainer): item blocks owned exclusively and handed around, no all-reduce
def _worker_strat(rank, world, port, chunks, passes, per_item, outdir, bpr=1):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from svdfeature_amd.multi_gpu import StratifiedTrainer, stratified_plan
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    u, i, r = cases.planted_triples(40000, NU, NI, seed=9)
    a = OracleShard(make_oracle(CONF), torch, minibatch=True)
    plan = [[a.make_windows(sub) for sub in chunk] for chunk in stratified_plan(u, i, r, rank, world, chunks, NI, per_item, bpr)]
    st = StratifiedTrainer(a, plan, world, rank, dist, blocks_per_rank=bpr)
    for _ in range(passes):
        st.train_pass()
    st.gather_blocks()
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), W_item=a.t.view("W_item"), i_bias=a.t.view("i_bias"),
             W_user=a.t.view("W_user"), u_bias=a.t.view("u_bias"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,bpr", [(2, 1), (3, 1), (2, 2), (3, 2)])
def test_stratified_gloo_ranks_match_the_simulation_bit_for_bit(world, bpr, tmp_path):
    """StratifiedTrainer across gloo processes (point-to-point block hand-overs, broadcasts at the end) == the one-process simulation
    whose hand-overs are array copies; the item side is identical on every rank after gather_blocks"""
    import torch.multiprocessing as mp
    from multi_rank_utils import simulate_stratified
    chunks, passes, per_item = 2, 2, 12.0
    mp.spawn(_worker_strat, args=(world, _free_port(), chunks, passes, per_item, str(tmp_path), bpr), nprocs=world, join=True)
    u, i, r = cases.planted_triples(40000, NU, NI, seed=9)
    sim = simulate_stratified(CONF, u, i, r, world, chunks, passes, NI, per_item, blocks_per_rank=bpr)
    for rk in range(world):
        z = np.load(str(tmp_path / ("rank%d.npz" % rk)))
        for name in ("W_item", "i_bias", "W_user", "u_bias"):
            np.testing.assert_array_equal(z[name].view(np.uint32), sim[rk].t.view(name).view(np.uint32))
    z0 = np.load(str(tmp_path / "rank0.npz"))
    for rk in range(1, world):
        np.testing.assert_array_equal(z0["W_item"], np.load(str(tmp_path / ("rank%d.npz" % rk)))["W_item"])


def test_stratified_schedule_uses_every_instance_once_and_one_rank_is_the_plain_step():
    from multi_rank_utils import simulate_stratified
    from svdfeature_amd.multi_gpu import stratified_plan
    u, i, r = cases.planted_triples(30000, NU, NI, seed=3)
    world, chunks = 4, 3
    seen = []
    for rk in range(world):
        for chunk in stratified_plan(u, i, r, rk, world, chunks, NI, 8.0):
            assert len(chunk) == world
            for s, sub in enumerate(chunk):
                for (wu, wi, wr) in sub:
                    lo, hi = NI * ((rk + s) % world) // world, NI * ((rk + s) % world + 1) // world
                    assert np.all(wu % world == rk) and np.all((wi >= lo) & (wi < hi))
                    seen.append(wu.astype(np.int64) * NI + wi)
    np.testing.assert_array_equal(np.sort(np.concatenate(seen)), np.sort(u.astype(np.int64) * NI + i))
    # one rank, one chunk, windows of W instances: the window-minibatch step with the sums added in place
    one = simulate_stratified(CONF, u, i, r, 1, 1, 2, NI, per_item=30000 / NI / 5.0)
    ref = simulate(CONF, u, i, r, 1, 5, 2, minibatch=True)
    for name in ("W_item", "i_bias", "W_user", "u_bias"):
        np.testing.assert_array_equal(one[0].t.view(name).view(np.uint32), ref[0].t.view(name).view(np.uint32))


@pytest.mark.parametrize("world", [2, 4, 8])
def test_rmse_contract_of_the_stratified_schedule(world):
    """|dRMSE| <= 1e-4 against the sequential reference after equal passes: 1 M ratings, 20 K x 2 K, k = 16, 5 passes, 4 file-order
    chunks per pass, <= 32 updates per item per window (bench.py's rule; tools/stratified_calibration.py at configs[2] density: one
    chunk per pass moves the RMSE by the order noise of a reshuffle, +1.6e-4 at 4 ranks; four chunks +5.5e-5 / -4e-6 at 4 / 8 ranks)."""
    from multi_rank_utils import simulate_stratified
    nu, ni, n = 20000, 2000, 1_000_000
    u, i, r = cases.planted_triples(n + 100_000, nu, ni, seed=5)
    tu, ti, tr = u[n:], i[n:], r[n:]
    u, i, r = u[:n], i[:n], r[:n]
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=16)
    ref = cases.rmse(merged_predict(simulate(conf, u, i, r, 1, 1, 5), 1, tu, ti, tr), tr)
    got = cases.rmse(merged_predict(simulate_stratified(conf, u, i, r, world, 4, 5, ni, 32.0), world, tu, ti, tr), tr)
    assert abs(got - ref) <= 1e-4
    if world == 8:   # two item blocks per rank (the hand-over hides behind a step of training): same contract
        got2 = cases.rmse(merged_predict(simulate_stratified(conf, u, i, r, world, 4, 5, ni, 32.0, blocks_per_rank=2), world, tu, ti, tr), tr)
        assert abs(got2 - ref) <= 1e-4


def test_all_rank_stratified_plan_equals_the_per_rank_plans():
    from svdfeature_amd.multi_gpu import stratified_plan, stratified_plan_all_ranks
    u, i, r = cases.planted_triples(20000, 700, 90, seed=5)
    for world, chunks, P in ((1, 2, 1), (3, 2, 1), (4, 3, 2)):
        allp = stratified_plan_all_ranks(u, i, r, world, chunks, 90, 7.0, P)
        for rk in range(world):
            one = stratified_plan(u, i, r, rk, world, chunks, 90, 7.0, P)
            assert len(one) == len(allp[rk])
            for ca, cb in zip(one, allp[rk]):
                assert len(ca) == len(cb)
                for sa_, sb in zip(ca, cb):
                    assert len(sa_) == len(sb)
                    for wa, wb in zip(sa_, sb):
                        for x, y in zip(wa, wb):
                            np.testing.assert_array_equal(x, y)


def test_substep_checker_reduces_to_the_stale_step_when_no_item_is_hot():
    """oracle/svdf_oracle.c: svdo_update_window_substeps (the checker of the one-GPU window step with ordered sub-steps, round 6) against the stale-step
    checker it generalises: with `sub` at least the largest per-item count of the window every item takes ONE sub-step, computed against the window-start
    row -- exactly svdo_update_csr_batch_stale followed by W_item += dW, i_bias += db; bit for bit.  With a small `sub` the hot items move differently
    (and stay finite where thousands of stale changes summed at once do not)."""
    import cases
    from oracle import oracle
    from svdfeature_amd.data import CSRData
    oracle.build()
    nu, ni, n = 400, 60, 6000
    u, i, r = cases.planted_triples(n, nu, ni, seed=9, zipf=True)
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=24)
    d = CSRData.from_triples(u, i, r)
    top = int(np.bincount(i, minlength=ni).max())

    def make():
        o = oracle.OracleTrainer("port", 0, 0)
        o.seed(10)
        for k, v in conf:
            o.set_param(k, v)
        o.init_model()
        o.init_trainer()
        return o
    a, b, c = make(), make(), make()
    a.update_window_substeps(d, top)
    dW, db, dg = b.update_batch_stale(d)
    b.set_view("W_item", b.view("W_item") + dW)
    b.set_view("i_bias", b.view("i_bias") + db)
    for name in ("W_item", "i_bias", "W_user", "u_bias"):
        assert np.array_equal(a.view(name).view(np.uint32), b.view(name).view(np.uint32)), name
    c.update_window_substeps(d, 16)
    assert top > 64 and np.isfinite(c.view("W_item")).all()
    assert np.array_equal(c.view("W_user").view(np.uint32), a.view("W_user").view(np.uint32))      # the user side does not depend on the sub-step size
    assert not np.array_equal(c.view("W_item").view(np.uint32), a.view("W_item").view(np.uint32))
