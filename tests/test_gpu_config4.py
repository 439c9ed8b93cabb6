"""BASELINE configs[4] (pairwiseRank, 200 M pairs, k=128, N GPUs) and the user-group data path on N ranks, on one MI355X:
  * rank pairs through svdf_dataset_from_pairs: bit-exact vs the oracle (10 M pairs at the full model shape), including the
    sigmoid rank loss (glibc's expf restated on the device);
  * N simulated ranks on one GPU (HipShard windows of Pairs / BlockArrays, explicit sum instead of the collective) against
    the oracle-backed simulation of tests/multi_rank_utils.py, bit for bit;
  * the FULL size -- 200 M pairs, k=128 -- through size-independent properties (the CPU oracle would need minutes)."""
import numpy as np
import pytest

import cases
import svdfeature_amd as sa
from oracle import oracle
from svdfeature_amd.multi_gpu import Pairs

pytestmark = pytest.mark.gpu

VIEWS = ("W_user", "W_item", "u_bias", "i_bias", "W_ufeedback", "ufeedback_bias", "g_bias")


def _ready(mk, fmt, act, conf, seed=10):
    t = mk(fmt, act)
    t.seed(seed)
    for k, v in conf:
        t.set_param(k, v)
    t.init_model()
    t.init_trainer()
    return t


def hip(f, a):
    return sa.Trainer(f, a)


def port(f, a):
    return oracle.OracleTrainer("port", f, a)


def _same(t, o, names=VIEWS):
    for name in names:
        a, b = t.view(name), o.view(name)
        if a is None or b is None or a.size == 0:
            continue
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), name


@pytest.mark.parametrize("k,nobias", [(16, 1), (128, 1), (100, 0), (300, 1)])
def test_dataset_from_pairs_equals_the_csr_instances_and_the_oracle(k, nobias):
    """Three-column pairs == the merged rank instances (user:1, {lo: +-1, hi: -+1}, label 1) fed as CSR rows, == the oracle,
    byte for byte, two passes; k=300 takes the general kernel, the others the few-row fused kernel."""
    nu, ni, n = 900, 250, 30000
    u, p, q = cases.planted_pairs(n, nu, ni, seed=k)
    conf = cases.conf_with(cases.PAIR_CONF, num_user=nu, num_item=ni, num_factor=k, no_user_bias=nobias, learning_rate=0.05, ui_init_sigma=0.1)
    a, b, o = _ready(hip, 0, 3, conf), _ready(hip, 0, 3, conf), _ready(port, 0, 3, conf)
    da = a.dataset_from_pairs(u, p, q)
    csr = sa.pairs_as_csr(u, p, q)
    db = b.dataset_from_csr(csr)
    assert da.num_row == n and da.kind == (2 if k <= 256 else 1)
    assert da.algorithmic_bytes == n * (24 * k + 8 * (3 - nobias) + 16 + 24)
    for _ in range(2):
        a.train_dataset(da)
        b.train_dataset(db)
        o.update_batch(csr)
    _same(a, o)
    _same(b, o)
    tp = sa.pairs_as_csr(u[:2000], p[:2000], q[:2000])
    assert np.array_equal(a.predict_batch(tp).view(np.uint32), o.predict_batch(tp).view(np.uint32))
    with pytest.raises(sa.SvdfError, match="must differ"):
        a.dataset_from_pairs(u[:3], p[:3], p[:3])
    with pytest.raises(sa.SvdfError, match="item feature index exceed bound"):
        a.dataset_from_pairs(u[:3], p[:3], np.array([ni, 0, 1], np.uint32))
    e = a.dataset_from_pairs(u[:0], p[:0], q[:0])
    assert e.num_row == 0
    a.train_dataset(e)


def test_ten_million_pairs_match_the_oracle():
    """configs[4] model shape (1M x 100K, k=128, active_type=3, no user bias) on a 10 M-pair prefix of bench.py's stream, one
    pass: every parameter byte-identical to the sequential C oracle."""
    import bench
    nu, ni, n = 1_000_000, 100_000, 10_000_000
    u, p, q = bench.synth_pairs(n, nu, ni)
    conf = cases.conf_with(cases.PAIR_CONF, num_user=nu, num_item=ni, num_factor=128)
    t, o = _ready(hip, 0, 3, conf), _ready(port, 0, 3, conf)
    ds = t.dataset_from_pairs(u, p, q)
    t.train_dataset(ds)
    o.update_batch(sa.pairs_as_csr(u, p, q))
    _same(t, o, ("W_user", "W_item", "i_bias", "u_bias"))


def test_full_size_properties_200m_pairs():
    """BASELINE configs[4] at FULL size (200 M pairs, k=128) on one GPU, size-independent properties:
      * learning_rate = 0, wd = 0: a full pass leaves every parameter byte-identical,
      * composition: a pass over the first 120 M then over the last 80 M == one pass over all 200 M,
      * determinism: the same pass on two trainers gives identical bytes,
      * the schedule is a permutation: 200 M instances counted, no batch larger than num_item / 2 pairs (an item occurs once
        per batch, two items per pair), algorithmic bytes = 200 M x 3128."""
    import bench
    nu, ni, n = 1_000_000, 100_000, 200_000_000
    u, p, q = bench.synth_pairs(n, nu, ni)

    def make(**kw):
        return _ready(hip, 0, 3, cases.conf_with(cases.PAIR_CONF, num_user=nu, num_item=ni, num_factor=128, **kw))
    z = make(learning_rate=0, wd_user=0, wd_item=0)
    before = [z.view(v).copy() for v in ("W_user", "W_item", "i_bias")]
    dz = z.dataset_from_pairs(u, p, q)
    assert dz.num_row == n and dz.max_batch <= ni // 2 and dz.algorithmic_bytes == n * 3128
    z.train_dataset(dz)
    assert z.counter(0) == n
    for a, name in zip(before, ("W_user", "W_item", "i_bias")):
        assert np.array_equal(a.view(np.uint32), z.view(name).view(np.uint32)), "lr=0 pass changed " + name
    dz.close()
    z.close()
    a, b = make(), make()
    da = a.dataset_from_pairs(u, p, q)
    a.train_dataset(da)
    wa, ia = a.view("W_item").copy(), a.view("W_user").copy()
    ba = a.view("i_bias").copy()
    a.train_dataset(da)   # second pass on the same trainer for the determinism leg below
    w2 = a.view("W_item").copy()
    da.close()
    a.close()
    cut = 120_000_000
    for lo, hi in ((0, cut), (cut, n)):
        db = b.dataset_from_pairs(u[lo:hi], p[lo:hi], q[lo:hi])
        b.train_dataset(db)
        b.synchronize()
        db.close()
    assert np.array_equal(wa.view(np.uint32), b.view("W_item").view(np.uint32)), "composition broke W_item"
    assert np.array_equal(ia.view(np.uint32), b.view("W_user").view(np.uint32)), "composition broke W_user"
    assert np.array_equal(ba.view(np.uint32), b.view("i_bias").view(np.uint32)), "composition broke i_bias"
    b.close()
    c = make()
    dc = c.dataset_from_pairs(u, p, q)
    c.train_dataset(dc)
    c.train_dataset(dc)
    assert np.array_equal(w2.view(np.uint32), c.view("W_item").view(np.uint32)), "two passes are not deterministic"


@pytest.mark.parametrize("kind,world,windows", [("pairs", 2, 3), ("pairs", 4, 5), ("svdpp", 2, 3), ("svdpp", 3, 4)])
def test_simulated_ranks_on_one_gpu_pairs_and_user_groups(kind, world, windows):
    """The MI355X side of the N-rank path for configs[4] data: one trainer per rank on one GPU, HipShard windows built from
    Pairs / BlockArrays shards, the collective replaced by an explicit sum -- against the oracle-backed simulation, bit for
    bit (replicated ranges: W_item, i_bias and, for user-group trainers, W_ufeedback + its bias)."""
    import torch
    from multi_rank_utils import simulate
    from svdfeature_amd import BlockArrays
    from svdfeature_amd.multi_gpu import HipShard, shard_block_windows, shard_pair_windows
    passes = 2
    dev = torch.device("cuda", 0)
    if kind == "pairs":
        nu, ni = 2000, 300
        conf = cases.conf_with(cases.PAIR_CONF, num_user=nu, num_item=ni, num_factor=32, learning_rate=0.05, ui_init_sigma=0.1)
        u, p, q = cases.planted_pairs(30000, nu, ni, seed=6)
        fmt, act, data = 0, 3, Pairs(u, p, q)
        shard = lambda rk: shard_pair_windows(u, p, q, rk, world, windows)
    else:
        nu, ni = 600, 200
        conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=24, num_ufeedback=ni, wd_ufeedback=0.004,
                               ufeedback_init_sigma=0.01)
        data = BlockArrays.from_blocks(cases.user_blocks(500, nu, ni, ni, seed=12, max_rows=9, max_fb=6, split_every=5))
        fmt, act = 1, 0
        shard = lambda rk: shard_block_windows(data, rk, world, windows)
    ranks = []
    for rk in range(world):
        a = HipShard(_ready(hip, fmt, act, conf), torch, dev)
        ranks.append((a, a.make_windows(shard(rk))))
    for _ in range(passes):
        for w in range(windows):
            ds = []
            for a, wins in ranks:
                if w == 0:
                    a.delta_begin()
                a.train(wins[w])
                d = a.delta_get()
                a.stream.synchronize()
                ds.append(d.clone())
            total = ds[0]
            for d in ds[1:]:
                total = total + d
            torch.cuda.synchronize()
            for a, _ in ranks:
                a.delta_set(total)
    sim = simulate(conf, data, None, None, world, windows, passes, fmt=fmt, active=act)
    for rk in range(world):
        _same(ranks[rk][0].t, sim[rk].t)


def test_gather_of_user_rows_completes_the_model():
    """Each rank owns the user rows of user % world == rank; HipShard.gather_user_side sums the owners' rows so that a model
    saved by any rank is complete.  Emulated with the explicit masks on one GPU (the collective itself is a plain SUM)."""
    nu, ni, world = 400, 120, 3
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=16)
    u, i, r = cases.planted_triples(20000, nu, ni, seed=4)
    ts = []
    for rk in range(world):
        t = _ready(hip, 0, 0, conf)
        m = (u % world) == rk
        t.update_batch(sa.CSRData.from_triples(u[m], i[m], r[m]))
        ts.append(t)
    full_w = sum(np.where(((np.arange(nu) % world) == rk)[:, None], ts[rk].view("W_user"), 0) for rk in range(world)).astype(np.float32)
    full_b = sum(np.where((np.arange(nu) % world) == rk, ts[rk].view("u_bias"), 0) for rk in range(world)).astype(np.float32)
    ts[0].set_view("W_user", full_w)
    ts[0].set_view("u_bias", full_b)
    for rk in range(world):
        own = (np.arange(nu) % world) == rk
        np.testing.assert_array_equal(ts[0].view("W_user")[own], ts[rk].view("W_user")[own])
        np.testing.assert_array_equal(ts[0].view("u_bias")[own], ts[rk].view("u_bias")[own])
    with pytest.raises(sa.SvdfError):
        ts[0].set_view("W_user", full_w[:-1])


@pytest.mark.parametrize("world,windows,parts,k", [(2, 4, 2, 16), (3, 3, 3, 64)])
def test_simulated_ranks_piecewise_exchange(world, windows, parts, k):
    """svdf_item_delta_select: the exchange cut into item-range pieces (what bench.py --gpus N overlaps with training), N
    trainers playing the ranks on one GPU with the collective replaced by a sum, in the pipelined order of
    ShardedTrainer._train_pass_parts -- against the synchronous oracle simulation, bit for bit."""
    import torch
    from multi_rank_utils import simulate_parts
    from svdfeature_amd.multi_gpu import HipShard, shard_windows_parts
    nu, ni, n, passes = 3000, 400, 40000, 2
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=k)
    u, i, r = cases.planted_triples(n, nu, ni, seed=9)
    dev = torch.device("cuda", 0)
    ranks = []
    for rk in range(world):
        a = HipShard(_ready(hip, 0, 0, conf), torch, dev, parts=parts)
        ranks.append((a, a.make_windows(shard_windows_parts(u, i, r, rk, world, windows, ni, parts))))
    for _ in range(passes):
        pending = None
        for w in range(windows):
            if w == 0:
                for a, _w in ranks:
                    a.delta_begin()
            for part in range(parts):
                ds = []
                for a, wins in ranks:
                    a.train(wins[w][part])
                    d = a.delta_get(part)
                    a.stream.synchronize()
                    ds.append(d.clone())
                total = ds[0]
                for d in ds[1:]:
                    total = total + d
                torch.cuda.synchronize()
                if pending is not None:      # the previous piece is applied only now, after this piece has trained
                    for a, _w in ranks:
                        a.delta_set(pending[0], pending[1])
                pending = (total, part)
        for a, _w in ranks:
            a.delta_set(pending[0], pending[1])
    sim = simulate_parts(conf, u, i, r, world, windows, passes, parts, ni)
    for rk in range(world):
        _same(ranks[rk][0].t, sim[rk].t, ("W_item", "i_bias", "W_user", "u_bias"))


def test_piecewise_exchange_through_rccl_async_collectives_world_one():
    """The real pipeline of bench.py --gpus N -- ShardedTrainer(parts=2) over HipShard with torch.distributed backend "nccl"
    (RCCL) and async_op collectives ordered against the trainer's stream -- on the one GPU of this box (world 1: the sum is the
    identity, the delta round trip snapshot + (current - snapshot) is not): equal to the synchronous simulation bit for bit."""
    import os
    import torch
    import torch.distributed as dist
    from multi_rank_utils import simulate_parts
    from svdfeature_amd.multi_gpu import HipShard, ShardedTrainer, shard_windows_parts
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = "29617"
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        nu, ni, n, windows, parts, passes = 2000, 300, 30000, 3, 2, 2
        conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=64)
        u, i, r = cases.planted_triples(n, nu, ni, seed=2)
        a = HipShard(_ready(hip, 0, 0, conf), torch, torch.device("cuda", 0), parts=parts)
        wins = a.make_windows(shard_windows_parts(u, i, r, 0, 1, windows, ni, parts))
        st = ShardedTrainer(a, wins, 1, dist, force_exchange=True, parts=parts)
        for _ in range(passes):
            st.train_pass()
        sim = simulate_parts(conf, u, i, r, 1, windows, passes, parts, ni)
        _same(a.t, sim[0].t, ("W_item", "i_bias", "W_user", "u_bias"))
    finally:
        dist.destroy_process_group()
