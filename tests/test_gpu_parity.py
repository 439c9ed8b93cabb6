"""GPU parity tests (run with -m gpu on an MI355X): the HIP engine, driven through the C ABI, against
the golden vectors generated from the compiled reference and against the C oracle on the same
seeded inputs.

Bar: BIT-EXACT everywhere -- model files byte-identical to the reference's, predictions bit-identical.  That includes
the sigmoid links (active_type 1, 2, 3, 7): the device code restates glibc's expf (svdf_device.h: glibc_expf, checked
against the host libm on all 2^32 inputs by tools/check_expf.c), so there is no tolerance branch in this file.
"""
import hashlib
import os

import numpy as np
import pytest

import cases
import scenarios
import svdfeature_amd as sa
from oracle import oracle

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(cases.GOLDEN, "scenarios.npz"))


def hip(f, a):
    return sa.Trainer(f, a)


def port(f, a):
    return oracle.OracleTrainer("port", f, a)


def _params(model_bytes):
    return np.frombuffer(model_bytes[4 + 1056:], dtype=np.float32)


@pytest.mark.parametrize("name", list(scenarios.SCENARIOS))
def test_scenarios_match_reference_golden(name):
    res = scenarios.run_scenario(name, hip)
    dg = scenarios.digest(res)
    assert dg["model0_md5"] == str(GOLD[name + "/model0_md5"])
    assert dg["model_len"] == int(GOLD[name + "/model_len"])
    np.testing.assert_array_equal(dg["model_sample"].view(np.uint32), GOLD[name + "/model_sample"].view(np.uint32))
    assert dg["model_md5"] == str(GOLD[name + "/model_md5"]), "model file is not byte-identical to the reference's"
    assert dg["pred_md5"] == str(GOLD[name + "/pred_md5"])
    assert dg["rmse"] == float(GOLD[name + "/rmse"])


@pytest.mark.parametrize("name", ["sparse_side_tables", "svdpp_random", "basicmf_ml100k_k16", "sparse_reg_project",
                                  "sparse_lazy_l2", "svdpp_random_lazy"])
@pytest.mark.parametrize("chunk,window", [(7, 1 << 22), (64, 50), (1, 3)])
def test_staging_boundaries_do_not_change_the_result(name, chunk, window):
    """Feeding rows in small update() chunks and flushing every `window` staged instances must give the
    bytes of one big batch (sequential semantics are kept across flushes)."""
    if name == "basicmf_ml100k_k16" and chunk == 1:
        pytest.skip("90k single-row calls: covered by the smaller scenarios")

    def mk(f, a):
        t = sa.Trainer(f, a)
        t.set_knob("stage_window", window)
        return t
    res = scenarios.run_scenario(name, mk, chunk=chunk)
    assert hashlib.md5(res["model"]).hexdigest() == str(GOLD[name + "/model_md5"])
    assert hashlib.md5(res["pred"].tobytes()).hexdigest() == str(GOLD[name + "/pred_md5"])


@pytest.mark.parametrize("k", [1, 2, 3, 4, 5, 8, 10, 16, 17, 31, 32, 33, 64, 100, 128, 200, 256,
                               257, 300, 511, 512, 513, 700, 768, 770, 1023, 1024])
def test_every_factor_width_basic_and_general(k):
    """All lane-group shapes (1..64 lanes per row, ragged tails) on both kernels, and the wide rows beyond 256 factors
    (whole wave per row, 2..4 float4 slots per lane, general kernel only), bit-exact vs the oracle."""
    nu, ni, ng = 60, 45, 5
    u, i, r = cases.planted_triples(3000, nu, ni, seed=k)
    basic = sa.CSRData.from_triples(u, i, r)
    general = cases.sparse_feature_rows(400, nu, ni, ng, seed=100 + k)
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_global=ng, num_factor=k, wd_global=0.003,
                           wd_user_bias=0.001, learning_rate=0.01)
    outs = []
    for mk in (hip, port):
        t = mk(0, 0)
        t.seed(7)
        for kk, v in conf:
            t.set_param(kk, v)
        t.init_model()
        t.init_trainer()
        t.update_batch(basic)
        t.update_batch(general)
        t.update_batch(basic)
        t.finish_round()
        outs.append([t.view(v).copy() for v in ("W_user", "W_item", "u_bias", "i_bias", "g_bias")] + [t.predict_batch(general), t.predict_batch(basic)])
    for a, b in zip(*outs):
        np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32))


def test_single_instance_calls_interleaved_with_predict():
    """update(Elem) / predict(Elem) one instance at a time, like svd_feature.cpp's loop does."""
    d = cases.sparse_feature_rows(150, 20, 15, 5, 9)
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=20, num_item=15, num_global=5, num_factor=7, wd_global=0.01)
    trs = []
    for mk in (hip, port):
        t = mk(0, 0)
        t.seed(3)
        for k, v in conf:
            t.set_param(k, v)
        t.init_model()
        t.init_trainer()
        trs.append(t)
    for r in range(d.num_row):
        row = d.row(r)
        if r % 3 == 0:
            pa, pb = trs[0].predict_csr(*row), trs[1].predict_csr(*row)
            assert np.float32(pa).view(np.uint32) == np.float32(pb).view(np.uint32)
        for t in trs:
            t.update_csr(*row)
    for name in ("u_bias", "W_user", "i_bias", "W_item", "g_bias"):
        np.testing.assert_array_equal(trs[0].view(name).view(np.uint32), trs[1].view(name).view(np.uint32))
    assert trs[0].counter(0) == d.num_row


@pytest.mark.parametrize("i8", [0, 1])
@pytest.mark.parametrize("gpw", [1, 2, 4, 8])
def test_resident_dataset_basicmf_two_million_ratings(gpw, i8):
    """svdf_dataset_from_triples + svdf_train_dataset (the bench path) on 2M ratings, 50k x 5k, k=64,
    two passes: byte-identical parameters vs the sequential oracle, for every groups_per_wave and both row layouts of the
    specialised kernel (16 lanes per row; 8 lanes with two chunks each, k_basicmf_i8)."""
    nu, ni, n = 50000, 5000, 2_000_000 if gpw == 4 else 300_000 + 13 * gpw
    u, i, r = cases.planted_triples(n, nu, ni, seed=42)
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=64)
    t = hip(0, 0)
    t.set_knob("groups_per_wave", gpw)
    t.set_knob("basic_i8", i8)
    o = port(0, 0)
    for x in (t, o):
        x.seed(10)
        for k, v in conf:
            x.set_param(k, v)
        x.init_model()
        x.init_trainer()
    ds = t.dataset_from_triples(u, i, r)
    assert ds.num_row == n and ds.kind == (10 if (n >= (1 << 20) and i8) else 0) and ds.algorithmic_bytes == n * 1072
    d = sa.CSRData.from_triples(u, i, r)
    for _ in range(2):
        t.train_dataset(ds)
        o.update_batch(d)
    for name in ("W_user", "W_item", "u_bias", "i_bias"):
        np.testing.assert_array_equal(t.view(name).view(np.uint32), o.view(name).view(np.uint32))
    np.testing.assert_array_equal(t.predict_dataset(ds).view(np.uint32), o.predict_batch(d).view(np.uint32))
    ds.close()


def test_resident_dataset_general_and_nonunit_basic():
    nu, ni, ng = 300, 200, 12
    gen = cases.sparse_feature_rows(5000, nu, ni, ng, 5)
    u, i, r = cases.planted_triples(5000, nu, ni, seed=6)
    nonunit = sa.CSRData.from_triples(u, i, r)
    nonunit.feat_value[:] = np.random.default_rng(1).choice([1.0, 0.5, -1.0, 0.75], size=nonunit.feat_value.size).astype(np.float32)
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_global=ng, num_factor=24, wd_global=0.002)
    t, o = hip(0, 0), port(0, 0)
    for x in (t, o):
        x.seed(11)
        for k, v in conf:
            x.set_param(k, v)
        x.init_model()
        x.init_trainer()
    dg, dn = t.dataset_from_csr(gen), t.dataset_from_csr(nonunit)
    assert dg.kind == 1 and dn.kind == 0
    for _ in range(2):
        t.train_dataset(dg)
        t.train_dataset(dn)
        o.update_batch(gen)
        o.update_batch(nonunit)
    for name in ("W_user", "W_item", "u_bias", "i_bias", "g_bias"):
        np.testing.assert_array_equal(t.view(name).view(np.uint32), o.view(name).view(np.uint32))
    np.testing.assert_array_equal(t.predict_dataset(dg).view(np.uint32), o.predict_batch(gen).view(np.uint32))
    np.testing.assert_array_equal(t.predict_dataset(dn).view(np.uint32), o.predict_batch(nonunit).view(np.uint32))


def test_edge_cases_empty_and_degenerate_inputs():
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=10, num_item=10, num_global=3, num_factor=6)
    t, o = hip(0, 0), port(0, 0)
    for x in (t, o):
        x.seed(2)
        for k, v in conf:
            x.set_param(k, v)
        x.init_model()
        x.init_trainer()
    empty = sa.CSRData.empty()
    t.update_batch(empty)
    t.finish_round()
    assert t.predict_batch(empty).size == 0
    # rows with no user / no item / nothing at all, and the same id on both ends of one row
    rows = [(3.0, [], [], [(1, 1.0)]), (2.0, [(0, 1.0)], [(1, 1.0)], []), (4.0, [], [], []),
            (5.0, [(2, 0.5), (2, 0.5)], [(3, 1.0), (3, 1.0)], [(4, 2.0), (4, -1.0)])]
    d = sa.CSRData.from_rows(rows)
    for x in (t, o):
        x.update_batch(d)
        x.update_batch(d)
    for name in ("W_user", "W_item", "u_bias", "i_bias", "g_bias"):
        np.testing.assert_array_equal(t.view(name).view(np.uint32), o.view(name).view(np.uint32))
    # out-of-range ids are rejected when staged, with the reference's messages
    with pytest.raises(sa.SvdfError, match="user feature index exceed bound"):
        t.update_batch(sa.CSRData.from_triples([10], [0], [1.0]))
    with pytest.raises(sa.SvdfError, match="item feature index exceed bound"):
        t.update_batch(sa.CSRData.from_triples([0], [10], [1.0]))
    with pytest.raises(sa.SvdfError, match="global feature index exceed setting"):
        t.update_batch(sa.CSRData.from_rows([(1.0, [(3, 1.0)], [(0, 1.0)], [(0, 1.0)])]))


def test_item_delta_roundtrip():
    """begin -> train -> buffer gives (current - snapshot); apply restores snapshot + delta (single rank:
    parameters unchanged by the round trip, bit for bit)."""
    nu, ni = 500, 300
    u, i, r = cases.planted_triples(20000, nu, ni, seed=3)
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=32)
    t = hip(0, 0)
    t.seed(10)
    for k, v in conf:
        t.set_param(k, v)
    t.init_model()
    t.init_trainer()
    w0, b0 = t.view("W_item").copy(), t.view("i_bias").copy()
    t.item_delta_begin()
    t.update_batch(sa.CSRData.from_triples(u, i, r))
    ptr, n = t.item_delta_buffer()
    assert n == ni * 32 + ni
    w1, b1 = t.view("W_item").copy(), t.view("i_bias").copy()
    t.item_delta_apply()
    t.synchronize()
    w2, b2 = t.view("W_item"), t.view("i_bias")
    np.testing.assert_allclose(w2, w0 + (w1 - w0), rtol=0, atol=0)
    np.testing.assert_allclose(b2, b0 + (b1 - b0), rtol=0, atol=0)


@pytest.mark.parametrize("world,windows,k", [(2, 4, 16), (3, 5, 10), (4, 3, 64)])
def test_two_simulated_ranks_on_one_gpu_match_the_oracle_simulation(world, windows, k):
    """The MI355X side of the multi-GPU exchange (HipShard: item_delta begin/export/import/apply on torch
    device tensors) driven by ShardedTrainer with a fake all-reduce that sums the two ranks' tensors;
    must equal the oracle-backed single-process simulation bit for bit."""
    import torch
    from multi_rank_utils import simulate
    from svdfeature_amd.multi_gpu import HipShard, ShardedTrainer, shard_windows
    nu, ni, n = 3000, 400, 40000
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=k)
    u, i, r = cases.planted_triples(n, nu, ni, seed=9)
    passes = 2
    dev = torch.device("cuda", 0)
    shards = []
    for rk in range(world):
        t = hip(0, 0)
        t.seed(10)
        for k, v in conf:
            t.set_param(k, v)
        t.init_model()
        t.init_trainer()
        a = HipShard(t, torch, dev)
        shards.append((a, a.make_windows(shard_windows(u, i, r, rk, world, windows))))
    # lock-step emulation of ShardedTrainer.train_pass over both ranks with an explicit sum
    for _ in range(passes):
        for w in range(windows):
            ds = []
            for a, wins in shards:
                if w == 0:                 # like ShardedTrainer: delta_set moves the snapshot along inside a pass
                    a.delta_begin()
                a.train(wins[w])
                d = a.delta_get()          # enqueued on the adaptor's own stream
                a.stream.synchronize()
                ds.append(d.clone())
            total = ds[0]
            for d in ds[1:]:
                total = total + d
            torch.cuda.synchronize()
            for a, _ in shards:
                a.delta_set(total)
    sim = simulate(conf, u, i, r, world, windows, passes)
    for rk in range(world):
        for name in ("W_item", "i_bias", "W_user", "u_bias"):
            np.testing.assert_array_equal(shards[rk][0].t.view(name).view(np.uint32), sim[rk].t.view(name).view(np.uint32))


def test_sharded_trainer_world_one_with_nccl_process_group():
    """torch.distributed backend "nccl" (RCCL) initialises on this box and all-reduces a delta tensor in
    place; with one rank the sum is the identity, so forcing the exchange must reproduce
    snapshot + (current - snapshot)."""
    import torch
    import torch.distributed as dist
    from svdfeature_amd.multi_gpu import HipShard
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29611")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        nu, ni = 2000, 300
        conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=64)
        u, i, r = cases.planted_triples(30000, nu, ni, seed=2)
        t = hip(0, 0)
        t.seed(10)
        for k, v in conf:
            t.set_param(k, v)
        t.init_model()
        t.init_trainer()
        a = HipShard(t, torch, torch.device("cuda", 0))
        ds = a.make_windows([(u, i, r)])[0]
        w0 = t.view("W_item").copy()
        a.delta_begin()
        a.train(ds)
        w1 = t.view("W_item").copy()
        d = a.delta_get()
        a.all_reduce(dist, d)
        a.delta_set(d)
        np.testing.assert_array_equal(t.view("W_item"), w0 + (w1 - w0))
        # fp16 wire format: same protocol, deltas rounded to half precision
        w2 = t.view("W_item").copy()
        a.delta_begin()
        a.train(ds)
        w3 = t.view("W_item").copy()
        a.set_wire_half(True)
        d = a.delta_get()
        assert d.dtype == torch.float16
        a.all_reduce(dist, d)
        a.delta_set(d)
        np.testing.assert_array_equal(t.view("W_item"), w2 + (w3 - w2).astype(np.float16).astype(np.float32))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("shape", ["pairwise", "globals_two_by_two", "mixed_absent"])
@pytest.mark.parametrize("k", [10, 64, 128])
def test_few_row_fused_kernel_matches_oracle_and_general_kernel(shape, k):
    """k_fused (<=2 user ids, <=2 item ids, distinct ids, any number of globals) vs the oracle, bit for bit,
    through the staged path and the resident-dataset path; use_fused=0 routes the same data through
    k_general and must give the same bytes."""
    nu, ni, ng = 400, 150, 40
    rng = np.random.default_rng(k)
    if shape == "pairwise":   # BPR pairs: one user, two items with +1/-1, label 1, no user bias
        n = 6000
        rows = []
        for _ in range(n):
            a, b = rng.choice(ni, 2, replace=False)
            lo, hi = min(a, b), max(a, b)
            rows.append((1.0, [], [(int(rng.integers(0, nu)), 1.0)], [(int(lo), 1.0 if lo == a else -1.0), (int(hi), 1.0 if hi == a else -1.0)]))
        d = sa.CSRData.from_rows(rows)
        active, extra = 0, dict(no_user_bias=1)
    else:
        d = cases.sparse_feature_rows(5000, nu, ni, ng, seed=k + 1, max_g=4 if shape != "mixed_absent" else 1, max_u=2, max_i=2, allow_dup=False)
        # sparse_feature_rows may repeat an id by chance: drop those rows so the data is fused-eligible
        keep = []
        for r in range(d.num_row):
            _, g, u_, i_, idx, _v = d.row(r)
            if len(set(idx[g:g + u_])) == u_ and len(set(idx[g + u_:])) == i_:   # global ids MAY repeat
                keep.append(r)
        d = sa.CSRData.concat([d.slice_rows(r, r + 1) for r in keep])
        active, extra = 0, {}
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_global=ng, num_factor=k, wd_global=0.004,
                           wd_user_bias=0.002, wd_item_bias=0.001, learning_rate=0.01, **extra)

    def make(mk):
        t = mk(0, active)
        t.seed(12)
        for kk, v in conf:
            t.set_param(kk, v)
        t.init_model()
        t.init_trainer()
        return t
    o = make(port)
    t_staged, t_ds, t_gen = make(hip), make(hip), make(hip)
    t_gen.set_knob("use_fused", 0)
    ds = t_ds.dataset_from_csr(d)
    assert ds.kind == 2
    for _ in range(2):
        o.update_batch(d)
        t_staged.update_batch(d)
        t_staged.finish_round()
        t_ds.train_dataset(ds)
        t_gen.update_batch(d)
        t_gen.finish_round()
    assert t_staged.counter(6) > 0 and t_staged.counter(5) == 0
    assert t_gen.counter(6) == 0 and t_gen.counter(5) > 0
    for name in ("W_user", "W_item", "u_bias", "i_bias", "g_bias"):
        ref = o.view(name).view(np.uint32)
        for t in (t_staged, t_ds, t_gen):
            np.testing.assert_array_equal(t.view(name).view(np.uint32), ref)
    np.testing.assert_array_equal(t_ds.predict_dataset(ds).view(np.uint32), o.predict_batch(d).view(np.uint32))


@pytest.mark.parametrize("i16", [0, 1])
@pytest.mark.parametrize("shape", ["pairs_sigmoid", "one_by_one", "two_by_two_absent"])
def test_few_row_kernel_k128_both_row_layouts(shape, i16):
    """k = 128 through the specialised few-row kernel in both layouts (32 lanes per row; 16 lanes with two chunks each,
    k_fewrow_i16): rank pairs with the sigmoid rank loss, plain (user, item) rows, rows with one or two user / item ids
    (absent slots), resident data sets whose last wave is partly filled -- parameters identical to the oracle's."""
    nu, ni = 500, 180
    rng = np.random.default_rng(7)
    if shape == "pairs_sigmoid":
        pu, pp, pq = cases.planted_pairs(9001, nu, ni, seed=3)
        d = sa.pairs_as_csr(pu, pp, pq)
        active, extra = 3, dict(no_user_bias=1)
    elif shape == "one_by_one":
        u, i, r = cases.planted_triples(9003, nu, ni, seed=4)
        d = sa.CSRData.from_triples(u, i, r)
        active, extra = 0, dict(wd_user=0.00001)   # decay factor that snaps to one
    else:
        rows = []
        for _ in range(7002):
            us = rng.choice(nu, int(rng.integers(1, 3)), replace=False)
            its = rng.choice(ni, int(rng.integers(1, 3)), replace=False)
            rows.append((float(rng.integers(1, 6)), [], [(int(x), float(np.float32(rng.uniform(0.2, 1.0)))) for x in sorted(us)],
                         [(int(x), float(np.float32(rng.uniform(0.2, 1.0)))) for x in sorted(its)]))
        d = sa.CSRData.from_rows(rows)
        active, extra = 0, {}
    conf = cases.conf_with(cases.PAIR_CONF if shape == "pairs_sigmoid" else cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=128,
                           wd_user_bias=0.002, wd_item_bias=0.001, learning_rate=0.01, ui_init_sigma=0.05, **extra)
    o, t = port(0, active), hip(0, active)
    for x in (o, t):
        x.seed(12)
        for kk, v in conf:
            x.set_param(kk, v)
        x.init_model()
        x.init_trainer()
    t.set_knob("fewrow_i16", i16)
    ds = t.dataset_from_csr(d)
    for _ in range(2):
        o.update_batch(d)
        t.train_dataset(ds)
    for name in ("W_user", "W_item", "u_bias", "i_bias"):
        np.testing.assert_array_equal(t.view(name).view(np.uint32), o.view(name).view(np.uint32))


@pytest.mark.parametrize("helpers", [4, 8, 16])
@pytest.mark.parametrize("k", [64, 128, 192, 256])
def test_svdpp_helper_waves_equal_one_wave_per_user(k, helpers):
    """svdpp_helpers = 4: four waves per user, the feedback rows gathered into LDS by all of them, accumulated by wave 0 in list
    order, the scatter applied by every wave to its share -- the same parameters as one wave per user and as the oracle; users with
    long lists (beyond the LDS budget), empty lists, split users (START / MIDDLE / END blocks) included."""
    nu, ni = 500, 700
    blocks = cases.user_blocks(260, nu, ni, ni, seed=31, max_rows=25, max_fb=40, split_every=5)
    blocks += cases.user_blocks(6, nu, ni, ni, seed=32, max_rows=8, max_fb=300)   # lists longer than the LDS rows at k = 192 / 256
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=k, num_ufeedback=ni, wd_ufeedback=0.004,
                           ufeedback_init_sigma=0.01)
    o = _ready(port, 1, conf)
    ts = [_ready(hip, 1, conf), _ready(hip, 1, conf)]
    ts[0].set_knob("svdpp_helpers", 1)
    ts[1].set_knob("svdpp_helpers", helpers)
    dss = [t.dataset_from_blocks(blocks) for t in ts]
    assert dss[0].num_simple_units > 0
    for _ in range(2):
        for b in blocks:
            o.update_block(b)
        for t, ds in zip(ts, dss):
            t.train_dataset(ds)
    for name in ("W_user", "W_item", "u_bias", "i_bias", "W_ufeedback", "ufeedback_bias"):
        ref = o.view(name).view(np.uint32)
        for t in ts:
            np.testing.assert_array_equal(t.view(name).view(np.uint32), ref)


@pytest.mark.parametrize("k", [3, 10, 16, 33, 64, 100, 128, 130, 192, 203, 256])
@pytest.mark.parametrize("nobias", [0, 1])
def test_svdpp_simple_unit_fast_path_and_block_dataset(k, nobias):
    """User-group data as a resident dataset (svdf_dataset_from_blocks): wave-per-user fast path
    (UNIT_SIMPLE: one user id per unit, distinct feedback ids) and the generic path
    (use_simple_units=0) both byte-identical to the oracle; split users (START/MIDDLE/END), users
    with a repeated item (fast path, the repeated rows re-read their item at use) and users with a
    repeated feedback id (not simple: generic path) included."""
    nu, ni = 600, 500
    blocks = cases.user_blocks(400, nu, ni, ni, seed=k + nobias, max_rows=40, max_fb=30, split_every=6)
    for b in blocks[::9]:     # the same item twice, a few rows apart and far apart
        if b.data.num_row >= 2 and b.extend_tag == 0:
            b.data.feat_index[3] = b.data.feat_index[1]
            if b.data.num_row >= 30:
                b.data.feat_index[2 * 29 + 1] = b.data.feat_index[1]
    for b in blocks[4::11]:   # a feedback id listed twice
        if b.num_ufeedback >= 2 and b.extend_tag == 0:
            b.index_ufeedback[1] = b.index_ufeedback[0]
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=k, num_ufeedback=ni, wd_ufeedback=0.004,
                           wd_ufeedback_bias=0.002, scale_lr_ufeedback=0.7, ufeedback_init_sigma=0.01, learning_rate=0.01,
                           no_user_bias=nobias, wd_user_bias=0.001)

    def make(mk):
        t = mk(1, 0)
        t.seed(21)
        for kk, v in conf:
            t.set_param(kk, v)
        t.init_model()
        t.init_trainer()
        return t
    o, t_ds, t_gen, t_staged = make(port), make(hip), make(hip), make(hip)
    t_gen.set_knob("use_simple_units", 0)
    ds, ds_gen = t_ds.dataset_from_blocks(blocks), t_gen.dataset_from_blocks(blocks)
    assert ds.kind == 3 and 0 < ds.num_simple_units < ds.num_units and ds_gen.num_simple_units == 0
    for _ in range(2):
        for b in blocks:
            o.update_block(b)
            t_staged.update_block(b)
        t_staged.finish_round()
        t_ds.train_dataset(ds)
        t_gen.train_dataset(ds_gen)
    for name in ("W_user", "W_item", "u_bias", "i_bias", "W_ufeedback", "ufeedback_bias"):
        ref = o.view(name).view(np.uint32)
        for t in (t_ds, t_gen, t_staged):
            np.testing.assert_array_equal(t.view(name).view(np.uint32), ref)
    want = np.concatenate([o.predict_block(b) for b in blocks if b.extend_tag == 0])
    got = t_ds.predict_dataset(ds)
    rows = np.concatenate([np.full(b.data.num_row, b.extend_tag == 0) for b in blocks])
    np.testing.assert_array_equal(got[rows].view(np.uint32), want.view(np.uint32))


def test_ten_million_ratings_match_the_oracle():
    """BASELINE configs[1] shape (1M x 100K, k=64) on a 10M-rating prefix, one pass: every parameter
    byte-identical to the sequential C oracle (about 3 s of CPU work)."""
    import bench
    nu, ni, n = 1_000_000, 100_000, 10_000_000
    u, i, r = bench.synth_triples(n, nu, ni)
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=64)
    t, o = hip(0, 0), port(0, 0)
    for x in (t, o):
        x.seed(10)
        for k, v in conf:
            x.set_param(k, v)
        x.init_model()
        x.init_trainer()
    ds = t.dataset_from_triples(u, i, r)
    t.train_dataset(ds)
    o.update_batch(sa.CSRData.from_triples(u, i, r))
    for name in ("W_user", "W_item", "u_bias", "i_bias"):
        assert np.array_equal(t.view(name).view(np.uint32), o.view(name).view(np.uint32)), name


def test_full_size_properties_100m_ratings():
    """BASELINE configs[1] at FULL size (100M ratings), where the CPU oracle would take minutes: size-
    independent properties instead --
      * learning_rate = 0, wd = 0: a full pass leaves every parameter byte-identical (every instance is
        read, scored and written back; nothing may be corrupted or dropped),
      * composition: pass over the first 60M then over the last 40M == one pass over all 100M (the schedule
        of a prefix composes with the schedule of the rest),
      * determinism: the same pass on two trainers gives identical bytes,
      * the schedule is a permutation: counters report exactly 100M instances in <= 100K-instance batches."""
    import bench
    nu, ni, n = 1_000_000, 100_000, 100_000_000
    u, i, r = bench.synth_triples(n, nu, ni)

    def make(**kw):
        t = hip(0, 0)
        t.seed(10)
        for k, v in cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=64, **kw):
            t.set_param(k, v)
        t.init_model()
        t.init_trainer()
        return t
    z = make(learning_rate=0, wd_user=0, wd_item=0)
    before = [z.view(v).copy() for v in ("W_user", "W_item", "u_bias", "i_bias")]
    dz = z.dataset_from_triples(u, i, r)
    assert dz.num_row == n and dz.max_batch <= ni and dz.algorithmic_bytes == n * 1072
    z.train_dataset(dz)
    assert z.counter(0) == n
    for a, name in zip(before, ("W_user", "W_item", "u_bias", "i_bias")):
        assert np.array_equal(a.view(np.uint32), z.view(name).view(np.uint32)), "lr=0 pass changed " + name
    dz.close()
    z.close()
    a, b = make(), make()
    da = a.dataset_from_triples(u, i, r)
    cut = 60_000_000
    db1, db2 = b.dataset_from_triples(u[:cut], i[:cut], r[:cut]), b.dataset_from_triples(u[cut:], i[cut:], r[cut:])
    a.train_dataset(da)
    b.train_dataset(db1)
    b.train_dataset(db2)
    for name in ("W_item", "i_bias", "u_bias", "W_user"):
        assert np.array_equal(a.view(name).view(np.uint32), b.view(name).view(np.uint32)), "composition broke " + name
    c = make()
    dc = c.dataset_from_triples(u, i, r)
    c.train_dataset(dc)
    assert np.array_equal(a.view("W_item").view(np.uint32), c.view("W_item").view(np.uint32))
    assert np.array_equal(a.view("W_user").view(np.uint32), c.view("W_user").view(np.uint32))


def _ready(mk, fmt, conf, seed=21):
    t = mk(fmt, 0)
    t.seed(seed)
    for kk, v in conf:
        t.set_param(kk, v)
    t.init_model()
    t.init_trainer()
    return t


def test_buffer_file_datasets_csr(tmp_path):
    """svdf_dataset_from_buffer_file on CSR buffers: ML-100K in the reference's 1000-row batches
    (basicMF kernel), sparse multi-feature rows in ragged 7-row batches (general kernel), and the buffer bytes
    the reference's own tool wrote (tests/golden/fixtures/ua.base.buffer): same parameters, bit for bit, as
    the oracle fed instance by instance."""
    tr, _ = cases.ml100k()
    p1 = str(tmp_path / "ua.base.buffer")
    sa.data.write_csr_buffer(p1, tr, 1000)
    conf = cases.conf_with(cases.BASICMF_CONF, num_factor=64)
    o, t = _ready(port, 0, conf), _ready(hip, 0, conf)
    ds = t.dataset_from_buffer_file(p1)
    assert ds.num_row == tr.num_row and ds.kind == 0
    for _ in range(2):
        o.update_batch(tr)
        t.train_dataset(ds)
    for name in ("W_user", "W_item", "u_bias", "i_bias"):
        np.testing.assert_array_equal(t.view(name).view(np.uint32), o.view(name).view(np.uint32))

    nu, ni, ng = 300, 200, 50
    rows = cases.sparse_feature_rows(5000, nu, ni, ng, seed=3)
    p2 = str(tmp_path / "sparse.buffer")
    sa.data.write_csr_buffer(p2, rows, 7)
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_global=ng, num_factor=10, wd_global=0.001)
    o, t = _ready(port, 0, conf), _ready(hip, 0, conf)
    ds = t.dataset_from_buffer_file(p2)
    assert ds.num_row == rows.num_row and ds.kind in (1, 2)
    o.update_batch(rows)
    t.train_dataset(ds)
    for name in ("W_user", "W_item", "u_bias", "i_bias", "g_bias"):
        np.testing.assert_array_equal(t.view(name).view(np.uint32), o.view(name).view(np.uint32))
    np.testing.assert_array_equal(t.predict_dataset(ds).view(np.uint32), o.predict_batch(rows).view(np.uint32))

    fx = os.path.join(cases.GOLDEN, "fixtures", "ua.base.buffer")
    want = sa.data.read_csr_buffer(fx)
    conf = cases.conf_with(cases.BASICMF_CONF, num_factor=8)
    o, t = _ready(port, 0, conf), _ready(hip, 0, conf)
    ds = t.dataset_from_buffer_file(fx)
    assert ds.num_row == want.num_row > 0
    o.update_batch(want)
    t.train_dataset(ds)
    np.testing.assert_array_equal(t.view("W_item").view(np.uint32), o.view("W_item").view(np.uint32))


def test_buffer_file_datasets_user_group(tmp_path):
    """User-group buffer (make_ugroup_buffer format, incl. START/MIDDLE/END split users whose blocks carry the
    bit-31 tag word): native reader + SVD++ unit schedule == oracle update(block) in file order."""
    nu, ni = 400, 300
    blocks = cases.user_blocks(250, nu, ni, ni, seed=8, max_rows=30, max_fb=20, split_every=5)
    path = str(tmp_path / "ug.buffer")
    sa.data.write_ugroup_buffer(path, blocks)
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=32, num_ufeedback=ni, wd_ufeedback=0.004,
                           ufeedback_init_sigma=0.01, learning_rate=0.01)
    o, t = _ready(port, 1, conf), _ready(hip, 1, conf)
    ds = t.dataset_from_buffer_file(path, user_group=True)
    assert ds.kind == 3 and ds.num_row == sum(b.data.num_row for b in blocks)
    for _ in range(2):
        for b in blocks:
            o.update_block(b)
        t.train_dataset(ds)
    for name in ("W_user", "W_item", "u_bias", "i_bias", "W_ufeedback"):
        np.testing.assert_array_equal(t.view(name).view(np.uint32), o.view(name).view(np.uint32))


def test_buffer_file_errors(tmp_path):
    conf = cases.conf_with(cases.BASICMF_CONF, num_factor=8)
    t = _ready(hip, 0, conf)
    with pytest.raises(sa.SvdfError, match="can not open"):
        t.dataset_from_buffer_file(str(tmp_path / "missing.buffer"))
    tr, _ = cases.ml100k()
    good = str(tmp_path / "good.buffer")
    sa.data.write_csr_buffer(good, tr.slice_rows(0, 2500), 1000)
    raw = open(good, "rb").read()
    cut = str(tmp_path / "cut.buffer")
    open(cut, "wb").write(raw[: len(raw) - 10])
    with pytest.raises(sa.SvdfError, match="truncated"):
        t.dataset_from_buffer_file(cut)
    empty = str(tmp_path / "empty.buffer")
    np.array([0, 1000, 0], np.int32).tofile(empty)
    ds = t.dataset_from_buffer_file(empty)
    assert ds.num_row == 0
    t.train_dataset(ds)
    # an index beyond num_user is the reference's bound error, raised when the buffer is scheduled
    bad = bytearray(raw)
    first_index = 12 + 8 + 4 * (3 * 1000 + 1) + 4 * 1000
    bad[first_index:first_index + 4] = np.array([10 ** 6], np.uint32).tobytes()
    badp = str(tmp_path / "bad.buffer")
    open(badp, "wb").write(bytes(bad))
    with pytest.raises(sa.SvdfError, match="user feature index exceed bound"):
        t.dataset_from_buffer_file(badp)


def test_lazy_decay_on_resident_datasets():
    """reg_method/reg_global 4/5 (lazy decay, restated with the reference's unsigned counter arithmetic) through the
    resident-dataset path: the sample counter keeps running across passes, basicMF-shaped triples fall back to
    the general kernel, user-group units leave the register-resident fast path; bit-exact against the oracle."""
    nu, ni = 500, 300
    u, i, r = cases.planted_triples(20000, nu, ni, seed=31)
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=20, reg_method=4)
    o, t = _ready(port, 0, conf), _ready(hip, 0, conf)
    ds = t.dataset_from_triples(u, i, r)
    assert ds.kind == 1
    d = sa.CSRData.from_triples(u, i, r)
    for _ in range(3):
        o.update_batch(d)
        t.train_dataset(ds)
    for name in ("W_user", "W_item", "u_bias", "i_bias"):
        np.testing.assert_array_equal(t.view(name).view(np.uint32), o.view(name).view(np.uint32))
    assert np.count_nonzero(t.view("u_bias")) > 0   # (factor rows are wiped by the wrapped exponent; biases keep training)

    blocks = cases.user_blocks(200, nu, ni, ni, seed=32, max_rows=20, max_fb=15, split_every=7)
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=32, num_ufeedback=ni, wd_ufeedback=0.004,
                           ufeedback_init_sigma=0.01, reg_method=5, wd_user=0.0)
    o, t = _ready(port, 1, conf), _ready(hip, 1, conf)
    ds = t.dataset_from_blocks(blocks)
    assert ds.kind == 3 and ds.num_simple_units == 0
    for _ in range(2):
        for b in blocks:
            o.update_block(b)
        t.train_dataset(ds)
    for name in ("W_user", "W_item", "u_bias", "i_bias", "W_ufeedback"):
        np.testing.assert_array_equal(t.view(name).view(np.uint32), o.view(name).view(np.uint32))

    # a dataset scheduled for the fast kernels cannot be trained after switching to a lazy mode
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=20)
    t = _ready(hip, 0, conf)
    ds = t.dataset_from_triples(u, i, r)
    assert ds.kind == 0
    t.set_param("reg_method", "4")
    with pytest.raises(sa.SvdfError, match="lazy decay"):
        t.train_dataset(ds)


@pytest.mark.parametrize("k,ng,fmt", [(64, 0, 0), (10, 7, 0), (33, 0, 1)])
def test_fused_delta_pack_unpack_equals_the_separate_kernels(k, ng, fmt):
    """svdf_item_delta_pack / _unpack (one launch over all replicated ranges, fp32 or fp16 on the wire, snapshot
    moved along) against the per-range sub / add kernels: same packed delta, same parameters after apply, and a
    second window packed from the refreshed snapshot equals begin + pack.  k=10 / 33: pitch != k; ng > 0: the
    global-bias range; format 1: the feedback rows in front of the user rows."""
    import torch
    nu, ni = 400, 250
    kw = dict(num_user=nu, num_item=ni, num_factor=k, num_global=ng)
    if fmt == 1:
        kw.update(num_ufeedback=ni, ufeedback_init_sigma=0.01)
    conf = cases.conf_with(cases.BASICMF_CONF, **kw)
    t1, t2 = _ready(hip, fmt, conf), _ready(hip, fmt, conf)
    if fmt == 1:
        blocks = cases.user_blocks(150, nu, ni, ni, seed=k, max_rows=10, max_fb=8)
        steps = [blocks[:70], blocks[70:]]
        train = lambda t, part: ([t.update_block(b) for b in part], t.finish_round())
    else:
        rows = cases.sparse_feature_rows(6000, nu, ni, max(ng, 1), seed=k) if ng else None
        u, i, r = cases.planted_triples(6000, nu, ni, seed=k)
        data = rows if ng else sa.CSRData.from_triples(u, i, r)
        steps = [data.slice_rows(0, 3000), data.slice_rows(3000, 6000)]
        train = lambda t, part: (t.update_batch(part), t.finish_round())
    n = t1.item_delta_count()
    pitch = (k + 3) // 4 * 4
    assert n == (ni * pitch + ni + ng) + ((ni * pitch + ni) if fmt == 1 else 0)
    dev = torch.device("cuda", 0)
    for half in (False, True):
        legacy = torch.empty(n, dtype=torch.float32, device=dev)
        fused = torch.empty(n, dtype=torch.float16 if half else torch.float32, device=dev)
        t1.item_delta_begin()
        t2.item_delta_begin()
        for w, part in enumerate(steps):
            train(t1, part)
            train(t2, part)
            t1.item_delta_into(legacy.data_ptr())
            t2.item_delta_pack(fused.data_ptr(), half)
            t1.synchronize()
            t2.synchronize()
            want = legacy.half() if half else legacy
            assert torch.equal(want, fused)
            summed = want + want               # "two identical ranks"
            t1.item_delta_apply_from(summed.float().contiguous().data_ptr())
            t2.item_delta_unpack(summed.data_ptr(), half, refresh_snapshot=True)
            t1.synchronize()
            t2.synchronize()
            names = ["W_item", "i_bias"] + (["g_bias"] if ng else []) + (["W_ufeedback", "ufeedback_bias"] if fmt == 1 else [])
            for name in names + ["W_user"]:
                np.testing.assert_array_equal(t1.view(name).view(np.uint32), t2.view(name).view(np.uint32))
            t1.item_delta_begin()              # legacy path copies; the fused path already moved its snapshot


def test_graph_replay_of_a_pass_is_the_same_launch_sequence():
    """use_graph=1 captures a resident dataset's pass into a hipGraph and replays it; a learning-rate change
    (decay_learning_rate, per round) must re-capture.  Same bytes as plain launches and as the oracle."""
    nu, ni = 3000, 800
    u, i, r = cases.planted_triples(200000, nu, ni, seed=12)
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=64, decay_learning_rate=1, decay_rate=0.9)
    o, t_plain, t_graph = _ready(port, 0, conf), _ready(hip, 0, conf), _ready(hip, 0, conf)
    t_graph.set_knob("use_graph", 1)
    d = sa.CSRData.from_triples(u, i, r)
    ds_p, ds_g = t_plain.dataset_from_triples(u, i, r), t_graph.dataset_from_triples(u, i, r)
    for rnd in range(3):
        for t in (o, t_plain, t_graph):
            t.set_round(rnd)
        for _ in range(2):                  # second pass of a round replays the captured graph
            o.update_batch(d)
            t_plain.train_dataset(ds_p)
            t_graph.train_dataset(ds_g)
    for name in ("W_user", "W_item", "u_bias", "i_bias"):
        ref = o.view(name).view(np.uint32)
        np.testing.assert_array_equal(t_plain.view(name).view(np.uint32), ref)
        np.testing.assert_array_equal(t_graph.view(name).view(np.uint32), ref)


@pytest.mark.parametrize("k,method", [(320, 0), (600, 2), (1000, 1)])
def test_wide_rows_user_groups_and_regularisers(k, method):
    """num_factor > 256 on the user-group (SVD++) path and with the L1 / projection regularisers, side tables and
    lazy decay excluded from nothing: wide rows run the same per-instance code as narrow ones."""
    nu, ni = 120, 90
    blocks = cases.user_blocks(60, nu, ni, ni, seed=k, max_rows=8, max_fb=6, split_every=5)
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=k, num_ufeedback=ni, wd_ufeedback=0.004,
                           ufeedback_init_sigma=0.01, learning_rate=0.01, reg_method=method, wd_user=0.02 if method else 0.004)
    o, t = _ready(port, 1, conf), _ready(hip, 1, conf)
    ds = t.dataset_from_blocks(blocks)
    assert ds.num_simple_units == 0
    for _ in range(2):
        for b in blocks:
            o.update_block(b)
        t.train_dataset(ds)
    for name in ("W_user", "W_item", "u_bias", "i_bias", "W_ufeedback", "ufeedback_bias"):
        np.testing.assert_array_equal(t.view(name).view(np.uint32), o.view(name).view(np.uint32))
    want = np.concatenate([o.predict_block(b) for b in blocks if b.extend_tag == 0])
    rows = np.concatenate([np.full(b.data.num_row, b.extend_tag == 0) for b in blocks])
    np.testing.assert_array_equal(t.predict_dataset(ds)[rows].view(np.uint32), want.view(np.uint32))
    # basicMF-shaped triples at this width are scheduled onto the general kernel
    u, i, r = cases.planted_triples(2000, nu, ni, seed=k)
    conf0 = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=k)
    o, t = _ready(port, 0, conf0), _ready(hip, 0, conf0)
    ds = t.dataset_from_triples(u, i, r)
    assert ds.kind == 1
    o.update_batch(sa.CSRData.from_triples(u, i, r))
    t.train_dataset(ds)
    np.testing.assert_array_equal(t.view("W_item").view(np.uint32), o.view("W_item").view(np.uint32))
    with pytest.raises(sa.SvdfError, match="num_factor > 1024"):
        _ready(hip, 0, cases.conf_with(cases.BASICMF_CONF, num_factor=1025))


@pytest.mark.parametrize("variant", ["l1", "project", "mixed3_nonneg", "ranges", "logistic"])
@pytest.mark.parametrize("k", [24, 128])
def test_svdpp_wave_path_general_configuration(variant, k):
    """The one-wave-per-user kernel outside its specialised configuration (k_svdpp_wave<NR, FAST=false>): L1 / projection /
    mixed regularisers, nonnegativity clamp, per-range decay, sigmoid link -- against the oracle, bit for bit."""
    nu, ni = 300, 260
    extra, kw, act = [], {}, 0
    if variant == "l1":
        kw = dict(reg_method=1, wd_user=0.02, wd_item=0.03)
    elif variant == "project":
        kw = dict(reg_method=2, wd_user=0.0008, wd_item=0.0009, ui_init_sigma=0.02)
    elif variant == "mixed3_nonneg":
        kw = dict(reg_method=3, wd_user=0.02, user_nonnegative=1)
    elif variant == "ranges":
        extra = [("up:wd", "0.01"), ("up:bound", "150"), ("up:wd", "0.001"), ("up:bound", str(nu)),
                 ("ip:wd", "0.02"), ("ip:bound", "100"), ("ip:wd", "0.003"), ("ip:bound", str(ni))]
    elif variant == "logistic":
        act, kw = 2, dict(base_score=0.4)
    blocks = cases.user_blocks(200, nu, ni, ni, seed=k, max_rows=25, max_fb=20, split_every=7,
                               **({"binary_label": True} if variant == "logistic" else {}))
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=k, num_ufeedback=ni, wd_ufeedback=0.004,
                           wd_ufeedback_bias=0.002, scale_lr_ufeedback=0.7, ufeedback_init_sigma=0.01, learning_rate=0.01, **kw) + extra

    def make(mk):
        t = mk(1, act)
        t.seed(5)
        for kk, v in conf:
            t.set_param(kk, v)
        t.init_model()
        t.init_trainer()
        return t
    o, t = make(port), make(hip)
    ds = t.dataset_from_blocks(blocks)
    assert ds.num_simple_units > 0
    for _ in range(2):
        for b in blocks:
            o.update_block(b)
        t.train_dataset(ds)
    for name in ("W_user", "W_item", "u_bias", "i_bias", "W_ufeedback", "ufeedback_bias"):
        np.testing.assert_array_equal(t.view(name).view(np.uint32), o.view(name).view(np.uint32))


@pytest.mark.parametrize("i8", [0, 1])
@pytest.mark.parametrize("wd", [(0.00005, 0.00002), (0.0, 0.0), (0.004, 0.0)])
def test_specialised_kernels_with_decay_factors_that_round_to_one(wd, i8):
    """The specialised basicMF and SVD++ kernels hoist the L2 decay factors and apply "skip the multiply when |s-1| <= 1e-6"
    as a multiply by exactly 1.0f: with lr*wd below 1e-6 (or zero) the factor snaps to one -- results must still be the
    oracle's bit for bit."""
    wd_user, wd_item = wd
    nu, ni = 2000, 700
    u, i, r = cases.planted_triples(100000, nu, ni, seed=17)
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=64, wd_user=wd_user, wd_item=wd_item,
                           wd_user_bias=0.0, wd_item_bias=0.00001)
    o, t = _ready(port, 0, conf), _ready(hip, 0, conf)
    t.set_knob("basic_i8", i8)
    ds = t.dataset_from_triples(u, i, r)
    assert ds.kind == 0
    d = sa.CSRData.from_triples(u, i, r)
    for _ in range(2):
        o.update_batch(d)
        t.train_dataset(ds)
    for name in ("W_user", "W_item", "u_bias", "i_bias"):
        np.testing.assert_array_equal(t.view(name).view(np.uint32), o.view(name).view(np.uint32))
    if i8:
        return
    blocks = cases.user_blocks(150, nu, ni, ni, seed=18, max_rows=30, max_fb=20)
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=128, num_ufeedback=ni, wd_ufeedback=0.00001,
                           ufeedback_init_sigma=0.01, wd_user=wd_user, wd_item=wd_item)
    o, t = _ready(port, 1, conf), _ready(hip, 1, conf)
    ds = t.dataset_from_blocks(blocks)
    assert ds.num_simple_units > 0
    for _ in range(2):
        for b in blocks:
            o.update_block(b)
        t.train_dataset(ds)
    for name in ("W_user", "W_item", "u_bias", "i_bias", "W_ufeedback"):
        np.testing.assert_array_equal(t.view(name).view(np.uint32), o.view(name).view(np.uint32))


def test_relaxed_shared_ids_lose_no_update_and_keep_the_accuracy():
    """Opt-in relaxed mode (extension keys amd:relax_*; NOT the reference's sequential semantics): shared ids are left out of
    the conflict schedule and updated with float atomics.  (1) No update may get lost: 60 000 instances of ONE launch all
    add to the same global bias from 256 CUs / 8 XCDs -- with a tiny learning rate the expected sum is
    lr * sum(label - pred0) to first order.  (2) With a realistic learning rate and shared globals + a shared second user
    feature, held-out RMSE stays within 2e-3 of the exact sequential oracle, in a fraction of the launches."""
    nu, ni = 60000, 60000
    n = 60000
    rng = np.random.default_rng(3)
    rows = [(float(rng.integers(1, 6)), [(0, 1.0)], [(j, 1.0)], [(j, 1.0)]) for j in range(n)]   # distinct user / item per instance
    d = sa.CSRData.from_rows(rows)
    lr = 1e-7
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_global=1, num_factor=16, learning_rate=lr,
                           wd_global=0.0, wd_user=0.0, wd_item=0.0) + [("amd:relax_global", "1")]
    t = _ready(hip, 0, conf)
    pred0 = t.predict_batch(d)
    ds = t.dataset_from_csr(d)
    assert ds.kind == 2 and ds.num_batches == 1          # the shared global no longer serialises the instances
    t.train_dataset(ds)
    got = float(t.view("g_bias")[0])
    want = float(np.sum(lr * (d.row_label.astype(np.float64) - pred0.astype(np.float64))))
    assert abs(got - want) <= 2e-3 * abs(want), (got, want)

    # (2) accuracy against the exact oracle on data whose shared ids would serialise exact execution
    nu, ni, ng, nbucket = 3000, 800, 40, 8
    u, i, r = cases.planted_triples(120000, nu, ni, seed=23)
    g = rng.integers(0, ng, (len(r), 2))
    bucket = nu + (u % nbucket)                           # second user feature: one of 8 shared ids after the real users
    rows = [(float(r[j]), [(int(x), 1.0) for x in sorted(set(g[j]))], [(int(u[j]), 1.0), (int(bucket[j]), 1.0)], [(int(i[j]), 1.0)])
            for j in range(len(r))]
    train, test = sa.CSRData.from_rows(rows[:100000]), sa.CSRData.from_rows(rows[100000:])
    base = cases.conf_with(cases.BASICMF_CONF, num_user=nu + nbucket, num_item=ni, num_global=ng, num_factor=16, wd_global=0.001)
    o = _ready(port, 0, base)
    t = _ready(hip, 0, base + [("amd:relax_global", "1"), ("amd:relax_user_from", str(nu))])
    ds = t.dataset_from_csr(train)
    exact_batches = _ready(hip, 0, base).dataset_from_csr(train).num_batches
    assert ds.num_batches * 20 < exact_batches
    for _ in range(3):
        o.update_batch(train)
        t.train_dataset(ds)
    rm_o = cases.rmse(o.predict_batch(test), test.row_label)
    rm_t = cases.rmse(t.predict_batch(test), test.row_label)
    assert abs(rm_o - rm_t) <= 2e-3, (rm_o, rm_t)
    # exact execution of shapes the fused kernel cannot take is refused in relaxed mode instead of silently serialised
    many = sa.CSRData.from_rows([(1.0, [], [(0, 1.0), (1, 1.0), (2, 1.0)], [(0, 1.0)])])
    with pytest.raises(sa.SvdfError, match="relaxed shared ids need few-row"):
        t.dataset_from_csr(many)


def test_relaxed_global_layout_is_invisible_outside_the_device(tmp_path):
    """Relaxed-global mode keeps one global bias per 128-byte line in HBM; model files, views and predictions must be
    unaffected by the layout, also when the mode is switched after the model is on the device."""
    nu, ni, ng = 300, 200, 37
    rows = cases.sparse_feature_rows(3000, nu, ni, ng, seed=9, max_u=1, max_i=1)
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_global=ng, num_factor=16, wd_global=0.002)
    exact = _ready(hip, 0, conf)
    exact.update_batch(rows)
    exact.finish_round()
    g_before = exact.view("g_bias").copy()
    pred_before = exact.predict_batch(rows)
    path = str(tmp_path / "m.model")
    exact.save_model(path)
    exact.set_param("amd:relax_global", "1")          # re-lays the device copy out (stride 32), values unchanged
    np.testing.assert_array_equal(exact.view("g_bias"), g_before)
    np.testing.assert_array_equal(exact.predict_batch(rows).view(np.uint32), pred_before.view(np.uint32))
    path2 = str(tmp_path / "m2.model")
    exact.save_model(path2)
    assert open(path, "rb").read() == open(path2, "rb").read()
    relaxed = sa.Trainer(0, 0)
    for kk, v in conf + [("amd:relax_global", "1")]:
        relaxed.set_param(kk, v)
    relaxed.load_model(path)
    relaxed.init_trainer()
    np.testing.assert_array_equal(relaxed.view("g_bias"), g_before)
    np.testing.assert_array_equal(relaxed.predict_batch(rows).view(np.uint32), pred_before.view(np.uint32))
    relaxed.set_param("amd:relax_global", "0")        # and back
    np.testing.assert_array_equal(relaxed.view("g_bias"), g_before)


def test_relaxed_svdpp_rows_shared_between_users():
    """User-group (SVD++) data with amd:relax_item_from = 0 and amd:relax_feedback = 1: item and feedback rows are no
    scheduling resource any more (all users of a pass fit a handful of launches) and are updated with atomic adds; each
    user's own rows stay sequential.  Accuracy against the exact oracle: held-out RMSE within 3e-3 after 3 passes."""
    nu, ni = 4000, 600
    blocks = cases.user_blocks(3000, nu, ni, ni, seed=41, max_rows=30, max_fb=20)
    test = cases.user_blocks(600, nu, ni, ni, seed=42, max_rows=10, max_fb=20)
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=32, num_ufeedback=ni, wd_ufeedback=0.004,
                           ufeedback_init_sigma=0.01, learning_rate=0.005)
    o = _ready(port, 1, conf)
    t = _ready(hip, 1, conf + [("amd:relax_item_from", "0"), ("amd:relax_feedback", "1")])
    ds = t.dataset_from_blocks(blocks)
    exact_batches = _ready(hip, 1, conf).dataset_from_blocks(blocks).num_batches
    assert ds.num_batches * 50 < exact_batches and ds.num_simple_units == ds.num_units
    for _ in range(3):
        for b in blocks:
            o.update_block(b)
        t.train_dataset(ds)
    want = np.concatenate([o.predict_block(b) for b in test])
    got = np.concatenate([t.predict_block(b) for b in test])
    lab = np.concatenate([b.data.row_label for b in test])
    assert abs(cases.rmse(want, lab) - cases.rmse(got, lab)) <= 3e-3, (cases.rmse(want, lab), cases.rmse(got, lab))
    with pytest.raises(sa.SvdfError, match="relaxed mode is amd:relax_item_from = 0"):
        bad = _ready(hip, 1, conf + [("amd:relax_item_from", "100")])
        bad.dataset_from_blocks(blocks)


@pytest.mark.parametrize("shape", ["inline_distinct", "more_than_four", "repeated_ids"])
@pytest.mark.parametrize("reg_global", [0, 1])
def test_fused_kernel_global_bias_paths(shape, reg_global):
    """The three ways k_fused handles an instance's global biases -- inline slots of the schedule record (every instance
    has <= 4 distinct ids), registers decided per instance (<= 4 distinct ids in a data set where others have more), and
    the walk through memory in the reference's order (an id listed twice: updated twice, then decayed twice,
    apex_svd_base.h:384-387, 288-292) -- against the oracle, bit for bit, with decay-free ids and per-range decay."""
    nu, ni, ng, k = 300, 120, 50, 32
    rng = np.random.default_rng(7 + reg_global)
    rows = []
    for r in range(4000):
        if shape == "inline_distinct":
            g = rng.choice(ng, size=int(rng.integers(0, 5)), replace=False)
        elif shape == "more_than_four":
            g = rng.choice(ng, size=int(rng.integers(0, 8)), replace=False)
        else:
            g = rng.integers(0, ng, size=int(rng.integers(1, 5)))
            if r % 3 == 0 and len(g) >= 2:
                g[1] = g[0]
        gl = [(int(x), float(np.float32(rng.uniform(-1, 1)))) for x in g]
        rows.append((float(rng.integers(1, 6)), gl, [(int(rng.integers(0, nu)), 1.0)], [(int(rng.integers(0, ni)), float(np.float32(rng.uniform(0.5, 1.5))))]))
    d = sa.CSRData.from_rows(rows)
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_global=ng, num_factor=k, wd_global=0.02, reg_global=reg_global,
                           num_regfree_global=5, learning_rate=0.01) + [("gp:wd", "0.05"), ("gp:bound", "20"), ("gp:wd", "0.001"), ("gp:bound", "50")]

    def make(mk):
        t = mk(0, 0)
        t.seed(3)
        for kk, v in conf:
            t.set_param(kk, v)
        t.init_model()
        t.init_trainer()
        return t
    o, t_ds, t_staged = make(port), make(hip), make(hip)
    ds = t_ds.dataset_from_csr(d)
    assert ds.kind == 2
    for _ in range(2):
        o.update_batch(d)
        t_ds.train_dataset(ds)
        t_staged.update_batch(d)
        t_staged.finish_round()
    for name in ("W_user", "W_item", "u_bias", "i_bias", "g_bias"):
        ref = o.view(name).view(np.uint32)
        np.testing.assert_array_equal(t_ds.view(name).view(np.uint32), ref)
        np.testing.assert_array_equal(t_staged.view(name).view(np.uint32), ref)
    np.testing.assert_array_equal(t_ds.predict_dataset(ds).view(np.uint32), o.predict_batch(d).view(np.uint32))


def test_dataset_may_outlive_its_trainer():
    """svdf_destroy before svdf_dataset_destroy (garbage collection order in a host language) must not touch freed state."""
    base, _ = cases.ml100k()
    for _ in range(20):
        t = hip(0, 0)
        for k, v in cases.conf_with(cases.BASICMF_CONF, num_factor=8):
            t.set_param(k, v)
        t.init_model()
        t.init_trainer()
        ds = [t.dataset_from_csr(base.slice_rows(0, 3000)), t.dataset_from_csr(cases.sparse_feature_rows(300, 943, 1682, 0, 1))]
        t.train_dataset(ds[0])
        t.close()
        for d in ds:
            d.close()


def test_device_expf_is_the_host_libm_expf_bit_for_bit():
    """glibc's expf restated on the device (svdf_device.h: glibc_expf) against the host libm the reference links: every
    4093rd bit pattern of the whole float space plus the full neighbourhoods of the special points (0, +-88, the overflow
    and underflow thresholds, denormal results, NaN / inf) -- about 1.1 M + 6 x 2^16 inputs, compared as bits.  (All 2^32
    inputs were compared once on the CPU restatement: tools/check_expf.c.)"""
    n = (1 << 32) // 4093
    dev = sa.device_expf(first=17, step=4093, n=n)
    ref = oracle.libm_expf(first=17, step=4093, n=n)
    same = (dev.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(dev) & np.isnan(ref))
    assert same.all(), "first differing input: bits 0x%08x" % (17 + 4093 * int(np.argmin(same)))
    for centre in (0.0, 88.0, -88.0, 88.7228, -103.972, -103.28, -87.3365, 32.5646, -63.0994):
        c = int(np.float32(centre).view(np.uint32))
        first = (c - (1 << 15)) & 0xFFFFFFFF
        dev = sa.device_expf(first=first, step=1, n=1 << 16)
        ref = oracle.libm_expf(first=first, step=1, n=1 << 16)
        assert ((dev.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(dev) & np.isnan(ref))).all(), centre
    special = np.array([np.inf, -np.inf, np.nan, 0.0, -0.0, 1e-45, -1e-45, 3.4e38, -3.4e38], np.float32)
    dev, ref = sa.device_expf(special), oracle.libm_expf(special)
    assert ((dev.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(dev) & np.isnan(ref))).all()
    rng = np.random.default_rng(5)
    x = (rng.standard_normal(1 << 20) * 12).astype(np.float32)   # the range sigmoid arguments live in
    assert np.array_equal(sa.device_expf(x).view(np.uint32), oracle.libm_expf(x).view(np.uint32))


def test_block_dataset_built_twice_keeps_its_fast_path_units():
    """The scheduler's per-unit distinctness stamps carry a per-call epoch: scheduling the same blocks a second time on
    the same trainer (unit indices restart at 0) must not mistake the first call's marks for repeats inside a unit --
    the number of fast-path units stays the same, also in relaxed mode where a demoted unit would be refused."""
    nu, ni = 500, 300
    blocks = cases.user_blocks(300, nu, ni, ni, seed=77, max_rows=12, max_fb=10)
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=32, num_ufeedback=ni, wd_ufeedback=0.004,
                           ufeedback_init_sigma=0.01)
    for extra in ([], [("amd:relax_item_from", "0"), ("amd:relax_feedback", "1")]):
        t = _ready(hip, 1, conf + extra)
        a = t.dataset_from_blocks(blocks)
        b = t.dataset_from_blocks(blocks)
        c = t.dataset_from_blocks(blocks)
        assert a.num_simple_units > 0
        assert a.num_simple_units == b.num_simple_units == c.num_simple_units
        assert a.num_batches == b.num_batches == c.num_batches
        if not extra:   # and the second copy trains to the oracle's bytes
            o = _ready(port, 1, conf)
            for blk in blocks:
                o.update_block(blk)
            t.train_dataset(b)
            for name in ("W_user", "W_item", "W_ufeedback", "u_bias", "i_bias", "ufeedback_bias"):
                np.testing.assert_array_equal(t.view(name).view(np.uint32), o.view(name).view(np.uint32))


def test_dataset_is_refused_under_another_scheduling_configuration():
    """A resident dataset bakes in the conflict schedule of the configuration it was built under; training it after
    the relaxed-id keys changed is refused instead of racing plain read-modify-writes on non-conflict-free batches."""
    nu, ni, ng = 200, 150, 6
    rows = cases.sparse_feature_rows(500, nu, ni, ng, seed=9, max_u=1, max_i=1, allow_dup=False)
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_global=ng, num_factor=16)
    t = _ready(hip, 0, conf)
    ds = t.dataset_from_csr(rows)
    t.train_dataset(ds)
    t.set_knob("use_fused", 0)
    with pytest.raises(sa.SvdfError, match="scheduled under another configuration"):
        t.train_dataset(ds)
    t.set_knob("use_fused", 1)
    t.train_dataset(ds)
    t.set_param("amd:relax_global", "1")
    with pytest.raises(sa.SvdfError, match="scheduled under another configuration"):
        t.train_dataset(ds)


def test_failed_prediction_rows_never_reach_the_training_stage():
    """predict() validates its rows and stages them in a buffer of its own: a prediction batch with a bad id raises and
    leaves nothing behind that a later update or flush could train on."""
    nu, ni = 120, 80
    u, i, r = cases.planted_triples(2000, nu, ni, seed=3)
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=16)
    t, o = _ready(hip, 0, conf), _ready(port, 0, conf)
    d = sa.CSRData.from_triples(u, i, r)
    t.update_batch(d)
    o.update_batch(d)
    bad = sa.CSRData.from_triples(np.array([1, 2, nu + 5], np.uint32), np.array([1, 2, 3], np.uint32), np.array([5, 5, 5], np.float32))
    with pytest.raises(sa.SvdfError, match="user feature index exceed bound"):
        t.predict_batch(bad)
    t.update_batch(d)
    o.update_batch(d)
    t.finish_round()
    for name in ("W_user", "W_item", "u_bias", "i_bias"):
        np.testing.assert_array_equal(t.view(name).view(np.uint32), o.view(name).view(np.uint32))
