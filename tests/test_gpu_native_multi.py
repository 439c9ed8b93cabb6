"""N GPUs behind ONE C-ABI handle, no Python in the exchange (svdf_multi.cpp; config key amd:gpus = N).  On the one-GPU test
box the ranks share the device ("virtual ranks": same sharding, same windows, same peer-pointer exchange kernels and events as N devices):
  * bit for bit the oracle-backed simulation of the algorithm (tests/multi_rank_utils.py) with fp32 deltas,
  * predictions routed to the owner of the user, model files with the owners' user rows gathered,
  * the accuracy contract |dRMSE| <= 1e-4 with the default fp16 wire format,
  * the reference's own CLI linked against the engine trains with amd:gpus = 2 from its config file."""
import os
import subprocess

import numpy as np
import pytest

import cases
import svdfeature_amd as sa
from multi_rank_utils import merged_predict, simulate
from svdfeature_amd import data as D

pytestmark = pytest.mark.gpu


def _ready(conf, extra=()):
    t = sa.Trainer(0, 0)
    t.seed(10)
    for k, v in list(conf) + list(extra):
        t.set_param(k, str(v))
    t.init_model()
    t.init_trainer()
    return t


@pytest.mark.parametrize("world,windows,k,step", [(2, 4, 16, "levels"), (3, 5, 10, "levels"), (4, 2, 64, "levels"),
                                                  (2, 4, 16, "minibatch"), (3, 5, 10, "minibatch"), (4, 2, 64, "minibatch"), (8, 3, 64, "minibatch")])
def test_virtual_ranks_match_the_oracle_simulation(world, windows, k, step, tmp_path):
    """staged update() calls on an amd:gpus handle, both window steps (amd:step): the window-minibatch step (default; user side
    exact, item side applied at the window's end) and exact conflict-free levels per rank -- against the simulation of each"""
    nu, ni, n = 3000, 400, 40000 if windows != 3 else 39000
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=k)
    u, i, r = cases.planted_triples(n, nu, ni, seed=9)
    passes = 2
    t = _ready(conf, [("amd:gpus", world), ("amd:delta_half", 0), ("amd:window", n // windows), ("amd:step", step)])
    assert n % windows == 0
    for _ in range(passes):
        t.update_batch(sa.CSRData.from_triples(u, i, r))   # one call: cut into `windows` exchange windows by the engine
        t.finish_round()
    assert t.counter(8) == passes * windows and t.counter(10) == 0   # exchanges happened; ranks share the device here
    assert t.counter(11) == (passes * windows if step == "minibatch" else 0) and t.counter(12) == 0   # p2p exchange kernels
    sim = simulate(conf, u, i, r, world, windows, passes, minibatch=(step == "minibatch"))
    # replicated side: identical on every rank, equal to the simulation's
    for name in ("W_item", "i_bias"):
        np.testing.assert_array_equal(t.view(name).view(np.uint32), sim[0].t.view(name).view(np.uint32))
    # user side: the owners' rows
    wu, bu = t.view("W_user"), t.view("u_bias")
    for rk in range(world):
        own = (np.arange(nu) % world) == rk
        np.testing.assert_array_equal(wu[own].view(np.uint32), sim[rk].t.view("W_user")[own].view(np.uint32))
        np.testing.assert_array_equal(bu[own].view(np.uint32), sim[rk].t.view("u_bias")[own].view(np.uint32))
    # predictions go to the owner of the user
    tu, ti, tr = cases.planted_triples(3000, nu, ni, seed=10)
    got = t.predict_batch(sa.CSRData.from_triples(tu, ti, tr))
    np.testing.assert_array_equal(got.view(np.uint32), merged_predict(sim, world, tu, ti, tr).view(np.uint32))
    # a saved model is complete: loading it into a single-GPU trainer reproduces the predictions
    p = str(tmp_path / "m.model")
    t.save_model(p)
    s = sa.Trainer(0, 0)
    s.load_model(p)
    s.init_trainer()
    np.testing.assert_array_equal(s.predict_batch(sa.CSRData.from_triples(tu, ti, tr)).view(np.uint32), got.view(np.uint32))
    # streaming single instances cuts the same windows as the one big call
    # the same data as ONE resident data set of the handle (sharded and windowed once, every piece in its rank's HBM): same result
    t2 = _ready(conf, [("amd:gpus", world), ("amd:delta_half", 0), ("amd:window", n // windows), ("amd:step", step)])
    ds = t2.dataset_from_triples(u, i, r)
    assert ds.kind == 6 and ds.num_row == n
    for _ in range(passes):
        t2.train_dataset(ds)
        t2.finish_round()
    assert t2.counter(8) == passes * windows
    for name in ("W_item", "i_bias", "W_user", "u_bias"):
        np.testing.assert_array_equal(t2.view(name).view(np.uint32), t.view(name).view(np.uint32))
    with pytest.raises(sa.SvdfError, match="training sets"):
        t2.predict_dataset(ds)
    ds.close()


@pytest.mark.parametrize("step", ["levels", "minibatch"])
def test_resident_rows_with_globals_and_user_group_blocks_on_virtual_ranks(step, tmp_path):
    """(i) rows with global features as a resident data set of the handle: amd:step = levels (exact conflict-free levels per rank, g_bias
    travels with the item side) == the same rows staged through update(); amd:step = minibatch (default: the window-minibatch step for user
    units, svdf_k_wunit.hip) == the oracle-backed simulation of that step; (ii) a user-group (SVD++) pass as a resident data set on 2 / 4
    virtual ranks == the multi_gpu.py simulation of the same step (blocks follow their user, windows cut where no START..END span is open),
    bit for bit; the same pass from a user-group buffer file."""
    mb = step == "minibatch"
    from multi_rank_utils import simulate as sim_blocks
    nu, ni, ng, n, world, windows = 300, 120, 6, 6000, 3, 3
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_global=ng, num_factor=20, wd_global=0.002)
    d = _rows_with_one_rank_per_row(n, nu, ni, ng, world, seed=4)
    if mb:   # the window-minibatch step for user units wants exactly one user entry per row (other rows keep the level scheme)
        from test_gpu_wunit import _rows_with_globals
        d = _rows_with_globals(n, nu, ni, ng, 3, seed=4, fixed=False)
    a = _ready(conf, [("amd:gpus", world), ("amd:delta_half", 0), ("amd:window", n // windows), ("amd:step", step)])
    b = _ready(conf, [("amd:gpus", world), ("amd:delta_half", 0), ("amd:window", n // windows), ("amd:step", step)])
    a.update_batch(d)
    a.finish_round()
    ds = b.dataset_from_csr(d)
    assert ds.kind == 6
    b.train_dataset(ds)
    assert b.counter(8) == windows and b.counter(11) == (windows if mb else 0)
    if not mb:
        for name in ("W_item", "i_bias", "g_bias", "W_user", "u_bias"):
            np.testing.assert_array_equal(a.view(name).view(np.uint32), b.view(name).view(np.uint32))
    else:   # (staged rows with global entries keep the level scheme; the resident data set takes the window-minibatch step)
        sim = sim_blocks(conf, d, None, None, world, windows, 1, minibatch=True)
        for name in ("W_item", "i_bias", "g_bias"):
            np.testing.assert_array_equal(b.view(name).view(np.uint32), sim[0].t.view(name).view(np.uint32))
        wu = b.view("W_user")
        for rk in range(world):
            own = (np.arange(nu) % world) == rk
            np.testing.assert_array_equal(wu[own].view(np.uint32), sim[rk].t.view("W_user")[own].view(np.uint32))
    # (ii) SVD++ blocks
    for world, windows in ((2, 3), (4, 2)):
        nu, ni = 240, 90
        blocks = cases.user_blocks(150, nu, ni, ni, seed=6 + world, max_rows=7, max_fb=5, split_every=4)
        ba = sa.BlockArrays.from_blocks(blocks)
        assert -(-ba.num_row // -(-ba.num_row // windows)) == windows
        pconf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=16, num_ufeedback=ni, wd_ufeedback=0.004, ufeedback_init_sigma=0.01)
        t = sa.Trainer(1, 0)
        t.seed(10)
        for k, v in pconf + [("amd:gpus", world), ("amd:delta_half", 0), ("amd:window", -(-ba.num_row // windows)), ("amd:step", step)]:
            t.set_param(k, str(v))
        t.init_model()
        t.init_trainer()
        ds = t.dataset_from_blocks(ba)
        assert ds.kind == 6
        for _ in range(2):
            t.train_dataset(ds)
        assert t.counter(8) == 2 * windows and t.counter(11) == (2 * windows if mb else 0)
        sim = sim_blocks(pconf, ba, None, None, world, windows, 2, fmt=1, minibatch=mb)
        for name in ("W_item", "i_bias", "W_ufeedback", "ufeedback_bias"):
            np.testing.assert_array_equal(t.view(name).view(np.uint32), sim[0].t.view(name).view(np.uint32))
        wu = t.view("W_user")
        for rk in range(world):
            own = (np.arange(nu) % world) == rk
            np.testing.assert_array_equal(wu[own].view(np.uint32), sim[rk].t.view("W_user")[own].view(np.uint32))
        # predictions of a block come from the owner of its user
        for b in blocks[:25]:
            own = int(b.data.feat_index[0]) % world
            np.testing.assert_array_equal(t.predict_block(b).view(np.uint32), sim[own].t.predict_block(b).view(np.uint32))
        path = str(tmp_path / ("ug%d.buffer" % world))
        D.write_ugroup_buffer(path, blocks)
        t3 = sa.Trainer(1, 0)
        t3.seed(10)
        for k, v in pconf + [("amd:gpus", world), ("amd:delta_half", 0), ("amd:window", -(-ba.num_row // windows)), ("amd:step", step)]:
            t3.set_param(k, str(v))
        t3.init_model()
        t3.init_trainer()
        ds3 = t3.dataset_from_buffer_file(path, user_group=True)
        for _ in range(2):
            t3.train_dataset(ds3)
        np.testing.assert_array_equal(t3.view("W_ufeedback").view(np.uint32), t.view("W_ufeedback").view(np.uint32))
        with pytest.raises(sa.SvdfError, match="resident data sets"):
            t3.update_block(blocks[0])


def test_refusals_of_an_amd_gpus_handle():
    """nothing is silently wrong: rows whose user ids belong to different ranks, a feature_user side table, RCCL with ranks that
    share a device, more than 16 ranks"""
    nu, ni = 60, 20
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=8)
    t = _ready(conf, [("amd:gpus", 2)])
    bad = sa.CSRData.from_rows([(3.0, [], [(2, 1.0), (5, 1.0)], [(1, 1.0)])])
    with pytest.raises(sa.SvdfError, match="belong to different ranks"):
        t.update_batch(bad)
        t.finish_round()
    t = _ready(conf, [("amd:gpus", 2), ("amd:exchange", "rccl")])
    u, i, r = cases.planted_triples(200, nu, ni, seed=1)
    with pytest.raises(sa.SvdfError, match="needs one device per rank"):
        t.update_batch(sa.CSRData.from_triples(u, i, r))
        t.finish_round()
    t = _ready(conf, [("amd:gpus", 2)])
    with pytest.raises(sa.SvdfError, match="exchanges by itself"):   # the per-rank exchange API belongs to the one-process-per-GPU scheme
        t.item_delta_begin()
    with pytest.raises(sa.SvdfError, match="exchanges by itself"):
        t.item_block_count()
    with pytest.raises(sa.SvdfError, match="at most 16 ranks"):
        _ready(conf, [("amd:gpus", 17)])
    with pytest.raises(sa.SvdfError, match="amd:exchange must be p2p or rccl"):
        _ready(conf, [("amd:exchange", "gloo")])


def test_native_multi_gpu_rmse_contract_fp16_wire():
    """default wire format (fp16) and the default window rule: within 1e-4 of the sequential single-GPU result after equal passes"""
    nu, ni, n = 20000, 2000, 1_000_000
    u, i, r = cases.planted_triples(n + 100_000, nu, ni, seed=5)
    tu, ti, tr = u[n:], i[n:], r[n:]
    u, i, r = u[:n], i[:n], r[:n]
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=16)
    d, dt = sa.CSRData.from_triples(u, i, r), sa.CSRData.from_triples(tu, ti, tr)
    seq = _ready(conf)
    multi = _ready(conf, [("amd:gpus", 8)])   # staged rows: windows cut on line at 12 updates per item (mean over the window's entries) = ~24 K instances
    for _ in range(5):
        for t in (seq, multi):
            t.update_batch(d)
            t.finish_round()
    assert multi.counter(8) == multi.counter(11) and 5 * 21 <= multi.counter(8) <= 5 * 60   # window-minibatch step, ~42 windows per pass
    a, b = cases.rmse(seq.predict_batch(dt), tr), cases.rmse(multi.predict_batch(dt), tr)
    assert abs(a - b) <= 1e-4, (a, b)


REFDIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref")
REF_CLI, AMD_CLI = os.path.join(REFDIR, "svd_feature"), os.path.join(REFDIR, "svd_feature_amd")


@pytest.mark.skipif(not (os.path.exists(REF_CLI) and os.path.exists(AMD_CLI)), reason="oracle/_ref CLIs are built in the build container only")
@pytest.mark.parametrize("rounds", [5, 40])
def test_reference_cli_trains_on_n_ranks_from_its_config_file(rounds, tmp_path):
    """svd_feature (the reference's trainer CLI, its config parser, buffer iterator and loader thread) linked against the
    engine, with `amd:gpus = N` in the config file: no Python, no torch, and NO hand-set window (round 4 needed `amd:window = 10000`:
    ML-100K's catalogue is skewed -- an instance meets 148 updates of its own item per pass, the top item 495 -- and the staged path
    trained a whole round as one window).  The staged path now cuts its windows from the data on line (svdf_multi.cpp: multi_flush).
    Held-out RMSE within 1e-4 of the unmodified reference binary after equal rounds, 2 / 4 / 8 virtual ranks, both steps; the model
    file is complete (user rows of every rank)."""
    base, test = cases.ml100k()
    conf = cases.conf_with(cases.BASICMF_CONF, num_factor=16)

    def run(name, cli, extra):
        d = tmp_path / name
        d.mkdir()
        D.write_csr_buffer(str(d / "train.buffer"), base)
        with open(str(d / "run.conf"), "w") as f:
            for k, v in conf + extra + [("buffer_feature", '"train.buffer"'), ("model_out_folder", '"./"')]:
                f.write("%s = %s\n" % (k, v))
        p = subprocess.run([cli, "run.conf", "num_round=%d" % rounds, "silent=1"], cwd=str(d), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
        assert p.returncode == 0, p.stdout.decode()
        path = str(d / ("%04d.model" % rounds))
        t = sa.Trainer(0, 0)
        t.load_model(path)
        t.init_trainer()
        return cases.rmse(t.predict_batch(test), test.row_label), path
    ref, ref_path = run("ref", REF_CLI, [])
    for step in ("minibatch", "levels"):
        for gpus in (2, 4, 8):
            rm, path = run("amd%d%s" % (gpus, step), AMD_CLI, [("amd:gpus", str(gpus)), ("amd:step", step)])
            assert abs(rm - ref) <= 1e-4, (step, gpus, rm, ref)
            assert open(ref_path, "rb").read() != open(path, "rb").read()   # window-synchronous, not sequential


def _rows_with_one_rank_per_row(n, nu, ni, ng, world, seed):
    """ragged rows (0..2 global, 0..2 user, 1..2 item entries) whose user ids all belong to ONE rank (id % world), as the handle requires"""
    rng = np.random.default_rng(seed)
    rows = []
    for _ in range(n):
        rk = int(rng.integers(0, world))
        nus = int(rng.integers(0, 3))
        us = sorted(set(int(x) * world + rk for x in rng.integers(0, nu // world, nus)))
        its = sorted(set(int(x) for x in rng.integers(0, ni, int(rng.integers(1, 3)))))
        gs = sorted(set(int(x) for x in rng.integers(0, ng, int(rng.integers(0, 3)))))
        rows.append((float(rng.integers(1, 6)), [(g, float(rng.uniform(0.2, 1))) for g in gs], [(x, float(rng.uniform(0.5, 1.5))) for x in us],
                     [(x, float(rng.uniform(0.5, 1.5))) for x in its]))
    return sa.CSRData.from_rows(rows)


def test_virtual_ranks_with_global_features_and_ragged_rows():
    """instances with global features and several user / item entries on an amd:gpus handle: a row follows its user ids (all of one
    rank), rows without a user entry go to rank 0, g_bias travels with the item side -- against the same algorithm run by hand with
    one single-GPU trainer per rank (explicit sharding, explicit delta sum)."""
    nu, ni, ng, n, world, windows = 300, 120, 6, 6000, 3, 3
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_global=ng, num_factor=20, wd_global=0.002)
    d = _rows_with_one_rank_per_row(n, nu, ni, ng, world, seed=4)
    t = _ready(conf, [("amd:gpus", world), ("amd:delta_half", 0), ("amd:window", n // windows)])
    t.update_batch(d)
    t.finish_round()
    # by hand
    ranks = [_ready(conf) for _ in range(world)]
    first_user = np.array([d.feat_index[d.row_ptr[3 * r + 1]] if d.row_ptr[3 * r + 2] > d.row_ptr[3 * r + 1] else 0 for r in range(n)])
    owner = np.where(np.array([d.row_ptr[3 * r + 2] > d.row_ptr[3 * r + 1] for r in range(n)]), first_user % world, 0)
    names = ("W_item", "i_bias", "g_bias")
    for w in range(windows):
        lo, hi = w * (n // windows), (w + 1) * (n // windows)
        snaps = [{v: x.view(v).copy() for v in names} for x in ranks]
        for rk, x in enumerate(ranks):
            rows = [r for r in range(lo, hi) if owner[r] == rk]
            if rows:
                x.update_batch(sa.CSRData.concat([d.slice_rows(r, r + 1) for r in rows]))
            x.finish_round()
        for v in names:
            total = None
            for rk, x in enumerate(ranks):
                dv = x.view(v) - snaps[rk][v]
                total = dv if total is None else total + dv
            for rk, x in enumerate(ranks):
                x.set_view(v, snaps[rk][v] + total)
    for v in names:
        np.testing.assert_array_equal(t.view(v).view(np.uint32), ranks[0].view(v).view(np.uint32))
    wu = t.view("W_user")
    for rk in range(world):
        own = (np.arange(nu) % world) == rk
        np.testing.assert_array_equal(wu[own].view(np.uint32), ranks[rk].view("W_user")[own].view(np.uint32))


@pytest.mark.parametrize("world,windows,k", [(2, 4, 16), (3, 5, 128), (8, 2, 64)])
def test_rank_pairs_as_a_resident_data_set_of_the_handle(world, windows, k):
    """BASELINE configs[4] on an amd:gpus handle: svdf_dataset_from_pairs shards the pairs by user and cuts them into exchange windows;
    every (rank, window) piece trains with the window-minibatch step (two signed item entries per pair) -- bit for bit the oracle-backed
    simulation, predictions routed to the owners"""
    from svdfeature_amd.multi_gpu import Pairs
    nu, ni, n, passes = 1200, 300, 40000, 2
    u, p, q = cases.planted_pairs(n, nu, ni, seed=k)
    conf = cases.conf_with(cases.PAIR_CONF, num_user=nu, num_item=ni, num_factor=k, learning_rate=0.05, ui_init_sigma=0.1)
    t = sa.Trainer(0, 3)
    t.seed(10)
    for kk, v in list(conf) + [("amd:gpus", world), ("amd:delta_half", 0), ("amd:window", n // windows)]:
        t.set_param(kk, str(v))
    t.init_model()
    t.init_trainer()
    ds = t.dataset_from_pairs(u, p, q)
    assert ds.kind == 6 and ds.num_row == n
    for _ in range(passes):
        t.train_dataset(ds)
        t.finish_round()
    assert t.counter(8) == passes * windows and t.counter(11) == passes * windows
    sim = simulate(conf, Pairs(u, p, q), None, None, world, windows, passes, active=3, minibatch=True)
    for name in ("W_item", "i_bias"):
        np.testing.assert_array_equal(t.view(name).view(np.uint32), sim[0].t.view(name).view(np.uint32))
    wu = t.view("W_user")
    for rk in range(world):
        own = (np.arange(nu) % world) == rk
        np.testing.assert_array_equal(wu[own].view(np.uint32), sim[rk].t.view("W_user")[own].view(np.uint32))
    with pytest.raises(sa.SvdfError, match="must differ"):
        t.dataset_from_pairs(u[:4], p[:4], p[:4])
    ds.close()
    # the exact level scheme has no pair entry point on the handle: the refusal says where to go
    t2 = sa.Trainer(0, 3)
    t2.seed(10)
    for kk, v in list(conf) + [("amd:gpus", 2), ("amd:step", "levels")]:
        t2.set_param(kk, str(v))
    t2.init_model()
    t2.init_trainer()
    with pytest.raises(sa.SvdfError, match="window-minibatch step"):
        t2.dataset_from_pairs(u, p, q)


def test_rank_pass_from_a_candidate_file_is_sharded_on_the_handle(tmp_path):
    """input_type = 2 on an amd:gpus handle: svdf_dataset_from_rank_buffer_file draws the round's pairs (device or host sampler, same libc
    stream) and shards the blocks over the ranks -- it must not build one engine's all-in-HBM pass and train rank 0 alone"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from perf_rank_input import write_candidates
    src = str(tmp_path / "cand.buffer")
    users, rows, items, k = 400, 9, 150, 16
    write_candidates(src, users, rows, items, seed=5)
    conf = [("num_user", users), ("num_item", items), ("num_global", 0), ("num_factor", k), ("num_ufeedback", 0), ("learning_rate", "0.01"),
            ("wd_user", "0.004"), ("wd_item", "0.004"), ("no_user_bias", 1), ("ui_init_sigma", "0.05")]
    out = {}
    for device_rank in (1, 0):
        t = sa.Trainer(1, 3)
        t.seed(10)
        for kk, v in conf + [("amd:gpus", 2), ("amd:delta_half", 0), ("amd:window", 300)]:
            t.set_param(kk, str(v))
        t.init_model()
        t.init_trainer()
        t.set_knob("device_rank", device_rank)
        for r in range(2):
            t.set_round(r)
            ds = t.dataset_from_rank_buffer_file(src)
            assert ds.kind == 6 and ds.num_row > 0
            t.train_dataset(ds)
            t.finish_round()
            ds.close()
        assert t.counter(8) > 0
        out[device_rank] = {n: t.view(n).copy() for n in ("W_user", "W_item", "i_bias")}
        t.close()
    for n in out[0]:
        np.testing.assert_array_equal(out[0][n].view(np.uint32), out[1][n].view(np.uint32))
    # both ranks' users moved (rank 1 owns the odd ids)
    s = sa.Trainer(1, 3)
    s.seed(10)
    for kk, v in conf:
        s.set_param(kk, str(v))
    s.init_model()
    w0 = s.view("W_user").copy()
    moved = np.any(out[1]["W_user"] != w0, axis=1)
    assert moved[0::2].sum() > 50 and moved[1::2].sum() > 50


@pytest.mark.parametrize("step", ["minibatch", "levels"])
def test_eval_dataset_on_the_handle(step, tmp_path):
    """svdf_eval_dataset on an amd:gpus handle: a resident data set (sharded, windowed) is scored piece by piece on the rank that holds
    it -- window data sets included -- and the squared error equals the one of the saved model on ONE GPU over the same rows"""
    nu, ni, n, world = 2000, 300, 30000, 3
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=32)
    u, i, r = cases.planted_triples(n, nu, ni, seed=3)
    tu, ti, tr = cases.planted_triples(9000, nu, ni, seed=4)
    t = _ready(conf, [("amd:gpus", world), ("amd:delta_half", 0), ("amd:window", 5000), ("amd:step", step)])
    ds = t.dataset_from_triples(u, i, r)
    test = t.dataset_from_triples(tu, ti, tr)
    for _ in range(2):
        t.train_dataset(ds)
        t.finish_round()
    ss, cnt = t.eval_dataset(test)
    assert cnt == len(tr)
    p = str(tmp_path / "m.model")
    t.save_model(p)
    s = sa.Trainer(0, 0)
    s.load_model(p)
    s.init_trainer()
    ss1, cnt1 = s.eval_dataset(s.dataset_from_triples(tu, ti, tr))
    assert cnt1 == cnt and abs(ss - ss1) <= 1e-9 * ss1
    pred = t.predict_batch(sa.CSRData.from_triples(tu, ti, tr))
    assert abs(ss - float(np.sum((pred.astype(np.float64) - tr) ** 2))) <= 1e-6 * ss
    with pytest.raises(sa.SvdfError, match="no file order"):
        t.predict_dataset(test)


def test_malformed_rows_are_reported_before_the_handle_reads_an_owner_from_them():
    """predict(block) and resident block data sets on an amd:gpus handle pick the owner rank from the rows: a decreasing row_ptr must
    raise the usual message instead of being read out of bounds"""
    nu, ni = 60, 20
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=8, num_ufeedback=ni)
    t = sa.Trainer(1, 0)
    t.seed(10)
    for k, v in conf + [("amd:gpus", 2)]:
        t.set_param(k, str(v))
    t.init_model()
    t.init_trainer()
    blk = cases.user_blocks(1, nu, ni, ni, seed=2, max_rows=3, max_fb=2)[0]
    bad = blk.data.row_ptr.copy()
    bad[1] = bad[2] + 5
    blk.data.row_ptr = bad
    with pytest.raises(sa.SvdfError, match="non-decreasing"):
        t.predict_block(blk)


def test_block_shapes_outside_the_window_step_keep_exact_levels_on_the_handle():
    """amd:gpus data sets from blocks default to the user-unit window step, whose builders refuse a feedback id listed twice
    in a block (the reference accepts it: prepare_ufeedback just adds the row twice, apex_svd_base.h:523-538).  Such data must keep the exact
    level scheme per rank -- as multi_dataset_from_csr's pre-scan already did for rows -- instead of failing inside the rank pool."""
    nu, ni, world = 120, 40, 2
    blocks = cases.user_blocks(60, nu, ni, ni, seed=3, max_rows=5, max_fb=4)
    ba = sa.BlockArrays.from_blocks(blocks)
    b = next(j for j in range(ba.num_block) if ba.fb_ptr[j + 1] - ba.fb_ptr[j] >= 2)
    ba.fb_index = ba.fb_index.copy()
    ba.fb_index[ba.fb_ptr[b] + 1] = ba.fb_index[ba.fb_ptr[b]]   # the same feedback id twice in one block
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=16, num_ufeedback=ni, wd_ufeedback=0.004, ufeedback_init_sigma=0.01)
    models = {}
    for step in (None, "levels"):
        t = sa.Trainer(1, 0)
        t.seed(10)
        for k, v in conf + [("amd:gpus", world), ("amd:delta_half", 0), ("amd:window", 100)] + ([("amd:step", step)] if step else []):
            t.set_param(k, str(v))
        t.init_model()
        t.init_trainer()
        ds = t.dataset_from_blocks(ba)
        t.train_dataset(ds)
        assert t.counter(11) == 0   # no window of the pass took the minibatch step
        models[step] = {n: t.view(n) for n in ("W_user", "W_item", "W_ufeedback", "i_bias")}
    for n, a in models[None].items():
        np.testing.assert_array_equal(a.view(np.uint32), models["levels"][n].view(np.uint32))
