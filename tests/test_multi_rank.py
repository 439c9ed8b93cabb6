"""N>1 path on CPU: world_size-2 gloo processes run svdfeature_amd.multi_gpu.ShardedTrainer (the same
code the MI355X ranks run) with the C oracle as the per-rank compute engine, and must reproduce the
single-process simulation of the algorithm bit for bit.  Also pins the accuracy contract of the
exchange: |RMSE(sharded) - RMSE(sequential reference order)| <= 1e-4 after equal passes."""
import os
import socket
import sys

import numpy as np
import pytest

import cases
from multi_rank_utils import OracleShard, make_oracle, merged_predict, simulate
from svdfeature_amd.multi_gpu import ShardedTrainer, defer_tails, shard_by_user, shard_windows, window_bounds

NU, NI = 3000, 400
CONF = cases.conf_with(cases.BASICMF_CONF, num_user=NU, num_item=NI, num_factor=16)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, windows, passes, outdir):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    u, i, r = cases.planted_triples(40000, NU, NI, seed=9)
    a = OracleShard(make_oracle(CONF), torch)
    wins = a.make_windows(shard_windows(u, i, r, rank, world, windows))
    st = ShardedTrainer(a, wins, world, dist)
    for _ in range(passes):
        st.train_pass()
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), W_item=a.t.view("W_item"), i_bias=a.t.view("i_bias"),
             W_user=a.t.view("W_user"), u_bias=a.t.view("u_bias"))
    dist.barrier()
    dist.destroy_process_group()


def test_sharding_helpers():
    u = np.array([0, 1, 2, 3, 4, 5, 6, 7], np.uint32)
    i = np.arange(8, dtype=np.uint32)
    r = np.arange(8, dtype=np.float32)
    a = shard_by_user(u, i, r, 1, 2)
    assert list(a[0]) == [1, 3, 5, 7] and list(a[1]) == [1, 3, 5, 7]
    assert window_bounds(10, 3) == [0, 3, 6, 10]
    parts = [shard_windows(u, i, r, rk, 2, 3) for rk in range(2)]
    # every instance lands in exactly one (rank, window), order preserved inside a shard
    seen = sorted(int(x) for p in parts for (_, ii, _) in p for x in ii)
    assert seen == list(range(8))
    assert shard_by_user(u, i, r, 0, 1)[0] is u


def test_two_gloo_ranks_match_single_process_simulation(tmp_path):
    import torch.multiprocessing as mp
    world, windows, passes = 2, 4, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, windows, passes, str(tmp_path)), nprocs=world, join=True)
    u, i, r = cases.planted_triples(40000, NU, NI, seed=9)
    sim = simulate(CONF, u, i, r, world, windows, passes)
    for rk in range(world):
        z = np.load(str(tmp_path / ("rank%d.npz" % rk)))
        for name in ("W_item", "i_bias", "W_user", "u_bias"):
            np.testing.assert_array_equal(z[name].view(np.uint32), sim[rk].t.view(name).view(np.uint32))
    # replicated parameters are identical on both ranks; user rows are owned by exactly one rank
    z0, z1 = np.load(str(tmp_path / "rank0.npz")), np.load(str(tmp_path / "rank1.npz"))
    np.testing.assert_array_equal(z0["W_item"], z1["W_item"])
    init = make_oracle(CONF).view("W_user")
    np.testing.assert_array_equal(z0["W_user"][1::2], init[1::2])   # rank 0 never touches odd users
    np.testing.assert_array_equal(z1["W_user"][0::2], init[0::2])


def test_world_one_is_the_sequential_reference_bit_for_bit():
    u, i, r = cases.planted_triples(20000, NU, NI, seed=4)
    a = simulate(CONF, u, i, r, 1, 5, 2)[0].t
    b = make_oracle(CONF)
    from svdfeature_amd import CSRData
    for _ in range(2):
        b.update_batch(CSRData.from_triples(u, i, r))
    for name in ("W_item", "i_bias", "W_user", "u_bias"):
        np.testing.assert_array_equal(a.view(name).view(np.uint32), b.view(name).view(np.uint32))


@pytest.mark.parametrize("world,windows", [(2, 8), (8, 16)])
def test_rmse_contract_of_the_exchange(world, windows):
    """north_star: RMSE within 1e-4 of the reference after equal epochs.  1M ratings, 20k x 2k (500 ratings
    per item per pass), k=16, 5 passes, windows chosen by bench.py's rule (about 64 ratings per item per
    window at 2 ranks, 42 at 3-4 ranks, 32 beyond).  At BASELINE configs[2] density (1000 per item) the same rule gives
    16 / 32 windows: measured 6.4e-5 (2 ranks), 4.3e-5 (4 ranks), 5.3e-5 (8 ranks) on a 10M-rating replica
    (DESIGN.md section 6)."""
    nu, ni, n = 20000, 2000, 1_000_000
    u, i, r = cases.planted_triples(n + 100_000, nu, ni, seed=5)
    tu, ti, tr = u[n:], i[n:], r[n:]
    u, i, r = u[:n], i[:n], r[:n]
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=16)
    ref = cases.rmse(merged_predict(simulate(conf, u, i, r, 1, 1, 5), 1, tu, ti, tr), tr)
    got = cases.rmse(merged_predict(simulate(conf, u, i, r, world, windows, 5), world, tu, ti, tr), tr)
    assert abs(got - ref) <= 1e-4


def test_deferred_tails_use_every_instance_once_and_keep_the_rmse_contract():
    """multi_gpu.defer_tails (bench.py applies it for N > 1): the short trailing batches of a window move to the
    next window.  Per rank the multiset of instances of a pass is unchanged, an instance moves by at most one
    window, the last window keeps its tail; accuracy stays within the 1e-4 contract."""
    nu, ni, n = 20000, 2000, 1_000_000
    u, i, r = cases.planted_triples(n + 100_000, nu, ni, seed=5)
    tu, ti, tr = u[n:], i[n:], r[n:]
    u, i, r = u[:n], i[:n], r[:n]
    world, windows = 8, 16
    before = shard_windows(u, i, r, 3, world, windows)
    after = defer_tails(before, nu, ni, 0.05)
    key = lambda ws: np.sort(np.concatenate([x[0].astype(np.int64) * ni + x[1] for x in ws]))
    np.testing.assert_array_equal(key(before), key(after))
    moved = sum(abs(len(a[2]) - len(b[2])) for a, b in zip(before, after))
    assert 0 < moved < 0.1 * sum(len(b[2]) for b in before)
    cum_b, cum_a = np.cumsum([len(x[2]) for x in before]), np.cumsum([len(x[2]) for x in after])
    assert np.all(cum_a <= cum_b) and cum_a[-1] == cum_b[-1] and np.all(cum_a[1:] >= cum_b[:-1])
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=16)
    ref = cases.rmse(merged_predict(simulate(conf, u, i, r, 1, 1, 5), 1, tu, ti, tr), tr)
    got = cases.rmse(merged_predict(simulate(conf, u, i, r, world, windows, 5, defer=0.05), world, tu, ti, tr), tr)
    assert abs(got - ref) <= 1e-4


def test_fp16_wire_format_keeps_the_rmse_contract():
    """ShardedTrainer(half_delta=True) semantics (deltas rounded to fp16, summed in fp16 like RCCL does) in the
    single-process simulation: still within 1e-4 of the sequential reference."""
    from svdfeature_amd.multi_gpu import shard_windows as sw
    nu, ni, n = 20000, 2000, 1_000_000
    u, i, r = cases.planted_triples(n + 100_000, nu, ni, seed=5)
    tu, ti, tr = u[n:], i[n:], r[n:]
    u, i, r = u[:n], i[:n], r[:n]
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=16)
    world, windows, passes = 8, 16, 5
    ranks = [OracleShard(make_oracle(conf)) for _ in range(world)]
    wins = [a.make_windows(sw(u, i, r, rk, world, windows)) for rk, a in enumerate(ranks)]
    for _ in range(passes):
        for w in range(windows):
            for rk, a in enumerate(ranks):
                a.delta_begin()
                a.train(wins[rk][w])
            total = None
            for a in ranks:
                d = a.delta_get().astype(np.float16)
                total = d if total is None else (total + d).astype(np.float16)
            for a in ranks:
                a.delta_set(total.astype(np.float32))
    ref = cases.rmse(merged_predict(simulate(conf, u, i, r, 1, 1, passes), 1, tu, ti, tr), tr)
    got = cases.rmse(merged_predict(ranks, world, tu, ti, tr), tr)
    assert abs(got - ref) <= 1e-4


def test_defer_tails_edge_cases():
    """Empty shards, a single window and a disabled threshold leave the windows alone; tiny windows whose batches are all
    'short' keep at least their first batch."""
    e = (np.zeros(0, np.uint32), np.zeros(0, np.uint32), np.zeros(0, np.float32))
    one = (np.array([0, 1, 2], np.uint32), np.array([0, 1, 2], np.uint32), np.ones(3, np.float32))
    out = defer_tails([e, one, e], 10, 10, 0.05)
    assert [len(x[2]) for x in out] == [0, 3, 0]
    assert [len(x[2]) for x in defer_tails([one], 10, 10, 0.05)] == [3]
    assert [len(x[2]) for x in defer_tails([one, one], 10, 10, 0.0)] == [3, 3]
    # one hot item: every instance conflicts with the previous one -> 6 batches of 1; the first batch always stays
    hot = (np.arange(6, dtype=np.uint32), np.zeros(6, np.uint32), np.ones(6, np.float32))
    out = defer_tails([hot, one], 10, 10, 0.5)
    assert len(out[0][2]) >= 1 and len(out[0][2]) + len(out[1][2]) == 9
    assert list(out[1][0][:len(out[1][2]) - 3]) == list(hot[0][len(out[0][2]):])   # deferred instances come first, in order


# ---- rank pairs and user-group data on N ranks (BASELINE configs[4]) ---------------------------------------------------
PNU, PNI = 2000, 300
PCONF = cases.conf_with(cases.PAIR_CONF, num_user=PNU, num_item=PNI, num_factor=16)
SNU, SNI = 600, 200
SCONF = cases.conf_with(cases.BASICMF_CONF, num_user=SNU, num_item=SNI, num_factor=16, num_ufeedback=SNI, wd_ufeedback=0.004,
                        ufeedback_init_sigma=0.01)


def _svdpp_pass():
    from svdfeature_amd import BlockArrays
    return BlockArrays.from_blocks(cases.user_blocks(500, SNU, SNI, SNI, seed=12, max_rows=9, max_fb=6, split_every=5))


def _worker_cfg4(rank, world, port, kind, windows, passes, outdir):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from svdfeature_amd.multi_gpu import shard_block_windows, shard_pair_windows
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if kind == "pairs":
        u, p, q = cases.planted_pairs(30000, PNU, PNI, seed=6)
        a = OracleShard(make_oracle(PCONF, active=3), torch)
        wins = a.make_windows(shard_pair_windows(u, p, q, rank, world, windows))
    else:
        a = OracleShard(make_oracle(SCONF, fmt=1), torch)
        wins = a.make_windows(shard_block_windows(_svdpp_pass(), rank, world, windows))
    st = ShardedTrainer(a, wins, world, dist)
    for _ in range(passes):
        st.train_pass()
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), **{n: a.t.view(n) for n in VIEWS7 if a.t.view(n) is not None})
    dist.barrier()
    dist.destroy_process_group()


VIEWS7 = ("W_item", "i_bias", "W_user", "u_bias", "W_ufeedback", "ufeedback_bias", "g_bias")


@pytest.mark.parametrize("kind", ["pairs", "svdpp"])
def test_two_gloo_ranks_pairs_and_user_groups_match_the_simulation(kind, tmp_path):
    """configs[4] data shapes on 2 ranks (gloo, the oracle as each rank's engine, the ShardedTrainer code the MI355X ranks
    run): rank pairs sharded by user, and SVD++ user blocks (START/MIDDLE/END spans kept together, W_ufeedback and its bias
    exchanged with the item side) -- bit for bit the single-process simulation."""
    import torch.multiprocessing as mp
    from svdfeature_amd.multi_gpu import Pairs
    world, windows, passes = 2, 3, 2
    mp.spawn(_worker_cfg4, args=(world, _free_port(), kind, windows, passes, str(tmp_path)), nprocs=world, join=True)
    if kind == "pairs":
        sim = simulate(PCONF, Pairs(*cases.planted_pairs(30000, PNU, PNI, seed=6)), None, None, world, windows, passes, active=3)
    else:
        sim = simulate(SCONF, _svdpp_pass(), None, None, world, windows, passes, fmt=1)
    zs = [np.load(str(tmp_path / ("rank%d.npz" % rk))) for rk in range(world)]
    for rk in range(world):
        for name in zs[rk].files:
            np.testing.assert_array_equal(zs[rk][name].view(np.uint32), sim[rk].t.view(name).view(np.uint32))
    for name in ("W_item", "i_bias") + (("W_ufeedback", "ufeedback_bias") if kind == "svdpp" else ()):
        np.testing.assert_array_equal(zs[0][name], zs[1][name])          # replicated ranges agree after the last exchange
    assert not np.array_equal(zs[0]["W_user"], zs[1]["W_user"])          # user rows are private


def test_block_sharding_helpers():
    from svdfeature_amd import BlockArrays
    from svdfeature_amd.multi_gpu import block_window_bounds, shard_block_windows
    blocks = cases.user_blocks(60, 80, 40, 40, seed=3, max_rows=6, max_fb=4, split_every=3)
    ba = BlockArrays.from_blocks(blocks)
    # flat form round-trips
    back = ba.to_blocks()
    assert len(back) == len(blocks)
    for a, b in zip(back, blocks):
        assert a.extend_tag == b.extend_tag and np.array_equal(a.index_ufeedback, b.index_ufeedback)
        assert np.array_equal(a.data.row_ptr, b.data.row_ptr - b.data.row_ptr[0]) and np.array_equal(a.data.feat_index, b.data.feat_index[b.data.row_ptr[0]:b.data.row_ptr[-1]])
    # a block's user: first user entry of its first row; spans stay together
    bu = ba.block_user()
    for j, b in enumerate(blocks):
        assert bu[j] == b.data.feat_index[b.data.row_ptr[1]]
    bounds = block_window_bounds(ba, 7)
    closed = ba.span_closed_before()
    assert bounds[0] == 0 and bounds[-1] == ba.num_block and all(closed[p] for p in bounds) and bounds == sorted(bounds)
    parts = [shard_block_windows(ba, rk, 3, 7) for rk in range(3)]
    # every block lands in exactly one (rank, window); per rank the file order is kept
    total_rows = sum(w.num_row for p in parts for w in p)
    total_blocks = sum(w.num_block for p in parts for w in p)
    assert total_rows == ba.num_row and total_blocks == ba.num_block
    for rk, p in enumerate(parts):
        for w in p:
            assert np.all(w.block_user() % 3 == rk)
            tags = list(w.extend_tag)
            open_ = False
            for t in tags:   # spans are complete inside one (rank, window)
                if t == 1:
                    assert not open_
                    open_ = True
                elif t == 3:
                    assert open_
                elif t == 2:
                    assert open_
                    open_ = False
                else:
                    assert not open_
            assert not open_
    # select keeps content: concatenating one rank's windows == selecting that rank's blocks directly
    rk0 = ba.select(bu % 3 == 0)
    cat_rows = np.concatenate([w.row_label for w in parts[0]])
    np.testing.assert_array_equal(cat_rows, rk0.row_label)
    np.testing.assert_array_equal(np.concatenate([w.feat_index for w in parts[0]]), rk0.feat_index)


def test_pair_sharding_and_deferred_tails():
    from svdfeature_amd.multi_gpu import Pairs, shard_pair_windows
    u, p, q = cases.planted_pairs(50000, PNU, PNI, seed=2)
    assert np.all(p != q)
    parts = [shard_pair_windows(u, p, q, rk, 4, 5) for rk in range(4)]
    assert sum(len(w.user) for pr in parts for w in pr) == len(u)
    assert all(isinstance(w, Pairs) and np.all(w.user % 4 == rk) for rk, pr in enumerate(parts) for w in pr)
    before = parts[1]
    after = defer_tails(before, PNU, PNI, 0.05)
    key = lambda ws: np.sort(np.concatenate([(x.user.astype(np.int64) * PNI + x.pos) * PNI + x.neg for x in ws]))
    np.testing.assert_array_equal(key(before), key(after))
    assert all(isinstance(w, Pairs) for w in after)
    assert sum(abs(len(a.user) - len(b.user)) for a, b in zip(before, after)) > 0


def test_pairwise_accuracy_contract_of_the_exchange():
    """Rank pairs on 4 simulated ranks against the sequential reference order: held-out pair accuracy within 3e-3 and the
    mean score margin within 2 % after equal passes (the pairwise analogue of the RMSE contract; 400 K pairs, 5 K x 500,
    k=16, 3 passes at 10x the demo learning rate so that the model actually learns in 3 passes -- accuracy 0.89 --,
    32 windows = about 50 item updates per item per window, bench.py's rule for pair data)."""
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
    seq = margins(simulate(conf, Pairs(u, p, q), None, None, 1, 1, 3, active=3), 1)
    par = margins(simulate(conf, Pairs(u, p, q), None, None, 4, 32, 3, active=3), 4)
    assert cases.pair_accuracy(seq) > 0.85                     # the model learned the planted preference
    assert abs(cases.pair_accuracy(par) - cases.pair_accuracy(seq)) <= 3e-3
    assert abs(float(par.mean()) - float(seq.mean())) <= 0.02 * abs(float(seq.mean()))


# ---- piece-wise exchange: the all-reduce of one item range overlaps with training on another -------------------------------
def _worker_parts(rank, world, port, windows, passes, parts, outdir, minibatch=False):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from svdfeature_amd.multi_gpu import shard_windows_parts
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    u, i, r = cases.planted_triples(40000, NU, NI, seed=9)
    a = OracleShard(make_oracle(CONF), torch, parts=parts, minibatch=minibatch)

    class AsyncShard(OracleShard):   # gloo: async collectives on CPU tensors
        pass
    a.all_reduce_async = lambda d_, t_: d_.all_reduce(t_, async_op=True)
    wins = a.make_windows(shard_windows_parts(u, i, r, rank, world, windows, NI, parts))
    st = ShardedTrainer(a, wins, world, dist, parts=parts)
    for _ in range(passes):
        st.train_pass()
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), W_item=a.t.view("W_item"), i_bias=a.t.view("i_bias"),
             W_user=a.t.view("W_user"), u_bias=a.t.view("u_bias"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("parts", [2, 3])
def test_two_gloo_ranks_piecewise_exchange_matches_the_synchronous_simulation(parts, tmp_path):
    """ShardedTrainer(parts=p): the window's exchange is cut by item id range and every piece's (asynchronous) all-reduce is
    finished only after the NEXT piece has been trained -- the values must be those of the synchronous piece-by-piece
    schedule, bit for bit (a piece's rows are not touched between its pack and its unpack)."""
    import torch.multiprocessing as mp
    from multi_rank_utils import simulate_parts
    world, windows, passes = 2, 4, 2
    mp.spawn(_worker_parts, args=(world, _free_port(), windows, passes, parts, str(tmp_path)), nprocs=world, join=True)
    u, i, r = cases.planted_triples(40000, NU, NI, seed=9)
    sim = simulate_parts(CONF, u, i, r, world, windows, passes, parts, NI)
    for rk in range(world):
        z = np.load(str(tmp_path / ("rank%d.npz" % rk)))
        for name in ("W_item", "i_bias", "W_user", "u_bias"):
            np.testing.assert_array_equal(z[name].view(np.uint32), sim[rk].t.view(name).view(np.uint32))
    z0, z1 = np.load(str(tmp_path / "rank0.npz")), np.load(str(tmp_path / "rank1.npz"))
    np.testing.assert_array_equal(z0["W_item"], z1["W_item"])


def test_two_gloo_ranks_piecewise_window_minibatch_matches_the_synchronous_simulation(tmp_path):
    """the same pipeline over the window-minibatch step (HipShard(minibatch=True, parts=2) on the GPU ranks): a piece's collective is
    finished after the next piece's user walks were issued, and the next window's walks over a piece only start after that piece's sum
    has been added -- values of the synchronous piece-by-piece schedule, bit for bit"""
    import torch.multiprocessing as mp
    from multi_rank_utils import simulate_parts
    world, windows, passes, parts = 2, 4, 2, 2
    mp.spawn(_worker_parts, args=(world, _free_port(), windows, passes, parts, str(tmp_path), True), nprocs=world, join=True)
    u, i, r = cases.planted_triples(40000, NU, NI, seed=9)
    sim = simulate_parts(CONF, u, i, r, world, windows, passes, parts, NI, minibatch=True)
    for rk in range(world):
        z = np.load(str(tmp_path / ("rank%d.npz" % rk)))
        for name in ("W_item", "i_bias", "W_user", "u_bias"):
            np.testing.assert_array_equal(z[name].view(np.uint32), sim[rk].t.view(name).view(np.uint32))


def test_piecewise_exchange_keeps_the_rmse_contract():
    """same windows, exchange in 2 item-range pieces: the order inside a window changes (items of the lower half first), the
    staleness does not -- within 1e-4 of the sequential reference like the whole-window exchange"""
    from multi_rank_utils import simulate_parts
    nu, ni, n = 20000, 2000, 1_000_000
    u, i, r = cases.planted_triples(n + 100_000, nu, ni, seed=5)
    tu, ti, tr = u[n:], i[n:], r[n:]
    u, i, r = u[:n], i[:n], r[:n]
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=16)
    ref = cases.rmse(merged_predict(simulate(conf, u, i, r, 1, 1, 5), 1, tu, ti, tr), tr)
    got = cases.rmse(merged_predict(simulate_parts(conf, u, i, r, 8, 16, 5, 2, ni), 8, tu, ti, tr), tr)
    assert abs(got - ref) <= 1e-4


def test_default_chunks_follow_the_skew_of_the_catalogue():
    """multi_gpu.default_chunks (round 6; profiles/r06_contract_zipf_c2.txt): uniform items keep 8 / 4 chunks per pass, a skewed catalogue trains
    with 12 at every rank count; the rule reads only the item column, so every rank computes the same number"""
    from svdfeature_amd.multi_gpu import default_chunks
    rng = np.random.default_rng(0)
    uni = rng.integers(0, 5000, 400000).astype(np.uint32)
    assert default_chunks(uni, 5000, 2) == 8 and default_chunks(uni, 5000, 8) == 4
    w = 1.0 / np.arange(1, 5001) ** 0.7
    zipf = rng.choice(5000, 400000, p=w / w.sum()).astype(np.uint32)
    assert default_chunks(zipf, 5000, 2) == 12 and default_chunks(zipf, 5000, 8) == 12
    assert default_chunks(np.zeros(0, np.uint32), 10, 8) == 4
