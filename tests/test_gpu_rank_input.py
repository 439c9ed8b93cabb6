"""Rank-pair input (input_type = 2) end to end on the GPU: svdf_dataset_from_rank_buffer_file draws the pairs of one
pass on the host (libc rand(), the reference's order), schedules them and trains them as resident user units.
Compared with the model files the reference's trainer CLI wrote (tests/golden/rank_input.npz), with the per-block
path fed from the sampled buffer, and -- where oracle/_ref is present -- with the reference's own generator feeding the
engine through the CLI binding."""
import os
import subprocess

import numpy as np
import pytest

import cases
import svdfeature_amd as sa
from svdfeature_amd import data as D

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(cases.GOLDEN, "rank_input.npz"))
REFDIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref")
AMD_CLI = os.path.join(REFDIR, "svd_feature_amd")
HEAD = 4 + 1056   # SVDTypeParam + SVDModelParam (apex_svd_model.h:373-477)


def _train_resident(tmp_path, src, conf, rounds, seed=10, kinds=None):
    t = sa.Trainer(1, 3)
    t.seed(seed)
    for k, v in conf:
        t.set_param(k, v)
    t.init_model()
    t.init_trainer()
    models = []
    p = str(tmp_path / "m.model")
    t.save_model(p)
    models.append(open(p, "rb").read())
    rows = 0
    for r in range(rounds):
        t.set_round(r)
        ds = t.dataset_from_rank_buffer_file(src)   # a new draw every pass, like itr->before_first()
        rows += ds.info(0)
        if kinds is not None:
            kinds.append(ds.kind)
        t.train_dataset(ds)
        t.finish_round()
        t.save_model(p)
        models.append(open(p, "rb").read())
        ds.close()
    t.close()
    return models, rows


def test_resident_rank_input_matches_the_reference_cli_models(tmp_path):
    src = str(tmp_path / "train.buffer")
    D.write_ugroup_buffer(src, cases.rank_blocks(150, 60, 50, 8, 900))
    models, rows = _train_resident(tmp_path, src, cases.RANK_E2E_CONF, cases.RANK_E2E_ROUNDS)
    assert rows > 500
    for r, m in enumerate(models):
        ref = GOLD["e2e/model_r%d" % r].tobytes()
        assert len(m) == len(ref) and m[:HEAD] == ref[:HEAD]
        a, b = np.frombuffer(m[HEAD:], np.float32), np.frombuffer(ref[HEAD:], np.float32)
        np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32))   # same libc stream, glibc's expf restated on the device


@pytest.mark.parametrize("max_fb", [3, 0], ids=["feedback", "nofeedback"])
@pytest.mark.parametrize("keys", [{}, {"rank_sample_method": "1", "rank_sample_gap": "0.5"}, {"rank_sample_pointwise": "1"}],
                         ids=["posneg", "cmp", "pointwise"])
def test_resident_rank_input_equals_the_per_block_path_on_the_sampled_file(keys, max_fb, tmp_path):
    """Same seed, same draws: training the resident dataset is byte-identical to sampling the pass into a buffer file
    and pushing its blocks through update(SVDPlusBlock) one by one.  Without any feedback id (the shape of
    demo/pairwiseRank) the resident path schedules the pairs one by one instead of walking whole users."""
    graded = "rank_sample_method" in keys
    src = str(tmp_path / "train.buffer")
    D.write_ugroup_buffer(src, cases.rank_blocks(120, 60, 50, 8, 31, graded, max_fb=max_fb))
    conf = cases.RANK_E2E_CONF + list(keys.items())
    kinds = []
    resident, _ = _train_resident(tmp_path, src, conf, 2, kinds=kinds)
    assert all(k == 3 for k in kinds) if max_fb else all(k in (1, 2) for k in kinds)   # user units vs single instances
    t = sa.Trainer(1, 3)
    t.seed(10)
    for k, v in conf:
        t.set_param(k, v)
    t.init_model()
    t.init_trainer()
    p = str(tmp_path / "b.model")
    for r in range(2):
        t.set_round(r)
        out = str(tmp_path / ("pass%d.buffer" % r))
        t.rank_sample_buffer_file(src, out)
        for b in D.read_ugroup_buffer(out):
            t.update_block(b)
        t.finish_round()
        t.save_model(p)
        assert open(p, "rb").read() == resident[r + 1], "round %d differs" % r
    t.close()


@pytest.mark.skipif(not os.path.exists(AMD_CLI), reason="oracle/_ref CLIs are built in the build container only")
def test_resident_rank_input_equals_the_reference_generator_feeding_the_engine(tmp_path):
    """The reference's CLI + its own PairwiseRankGenerator, linked against this engine (per-instance virtual calls),
    writes byte-identical models to the resident path: same pairs, same arithmetic."""
    d = tmp_path / "cli"
    d.mkdir()
    src = str(d / "train.buffer")
    D.write_ugroup_buffer(src, cases.rank_blocks(150, 60, 50, 8, 900))
    with open(str(d / "run.conf"), "w") as f:
        for k, v in cases.RANK_E2E_CONF:
            f.write("%s = %s\n" % (k, v))
        f.write('buffer_feature = "train.buffer"\nmodel_out_folder = "./"\n')
    p = subprocess.run([AMD_CLI, "run.conf", "num_round=%d" % cases.RANK_E2E_ROUNDS, "silent=1"], cwd=str(d),
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert p.returncode == 0, p.stdout.decode()
    models, _ = _train_resident(tmp_path, src, cases.RANK_E2E_CONF, cases.RANK_E2E_ROUNDS)
    for r, m in enumerate(models):
        assert m == open(str(d / ("%04d.model" % r)), "rb").read(), "round %d differs" % r


@pytest.mark.parametrize("factor", [32, 100])
def test_feedback_free_blocks_row_by_row_equal_whole_users(factor, tmp_path):
    """Blocks without feedback ids: scheduling the rows one by one (default) and walking every user as a sequential unit
    (knob rows_without_feedback = 0, the literal update(SVDPlusBlock) order) leave byte-identical models, at a size where
    users conflict heavily on items (2 000 users x up to 16 candidates over 300 items)."""
    src = str(tmp_path / "train.buffer")
    D.write_ugroup_buffer(src, cases.rank_blocks(2000, 500, 300, 8, 5, max_rows=16, max_fb=0))
    conf = cases.conf_with(cases.RANK_E2E_CONF, num_user=500, num_item=300, num_ufeedback=300, num_factor=factor)
    models = []
    for rows_knob in (1, 0):
        t = sa.Trainer(1, 3)
        t.set_knob("rows_without_feedback", rows_knob)
        t.seed(4)
        for k, v in conf:
            t.set_param(k, v)
        t.init_model()
        t.init_trainer()
        kinds = []
        for r in range(2):
            t.set_round(r)
            ds = t.dataset_from_rank_buffer_file(src)
            kinds.append(ds.kind)
            t.train_dataset(ds)
            t.finish_round()
            ds.close()
        p = str(tmp_path / ("k%d.model" % rows_knob))
        t.save_model(p)
        models.append(open(p, "rb").read())
        assert all(k == 3 for k in kinds) if rows_knob == 0 else all(k in (1, 2) for k in kinds)
        t.close()
    assert models[0] == models[1]


def test_rank_input_at_scale_count_and_determinism(tmp_path):
    """Size-independent properties at a size the host oracle would not finish quickly (50 K users x 64 candidates,
    100 K items, k=128, ~1.6 M pairs per pass): the number of pairs of a pass equals the number of negatives of the users
    that have both kinds (apex_svd_data.cpp:949-961, no rank_sample_num), every pass draws a different set, and two runs from
    the same seed write the same bytes."""
    import perf_rank_input
    users, rows, items = 50000, 64, 100000
    src = str(tmp_path / "cand.buffer")
    perf_rank_input.write_candidates(src, users, rows, items, 7)
    raw = np.fromfile(src, np.uint8)[16:]
    rec = 4 * (3 + (3 * rows + 1) + rows + 2 * rows + 2 * rows)
    labels = raw.reshape(users, rec)[:, 4 * (3 + 3 * rows + 1):4 * (3 + 3 * rows + 1 + rows)].copy().view(np.float32)
    npos, nneg = (labels >= 0.8).sum(1), (labels <= 0.0).sum(1)
    expected = int(nneg[(npos > 0) & (nneg > 0)].sum())
    conf = [("num_user", users), ("num_item", items), ("num_global", 0), ("num_factor", 128), ("num_ufeedback", 0), ("learning_rate", 0.005),
            ("wd_user", 0.004), ("wd_item", 0.004), ("active_type", 3), ("no_user_bias", 1)]
    outs = []
    for run in range(2):
        t = sa.Trainer(1, 3)
        t.seed(10)
        for k, v in conf:
            t.set_param(k, str(v))
        t.init_model()
        t.init_trainer()
        batches = []
        for r in range(2):
            t.set_round(r)
            ds = t.dataset_from_rank_buffer_file(src)
            assert ds.num_row == expected and ds.kind == 2
            batches.append(ds.num_batches)
            t.train_dataset(ds)
            t.finish_round()
            ds.close()
        assert batches[0] != batches[1]   # a new draw per pass
        outs.append((t.view("W_item").copy(), t.view("W_user").copy(), t.view("i_bias").copy(), tuple(batches)))
        t.close()
    assert outs[0][3] == outs[1][3]
    for a, b in zip(outs[0][:3], outs[1][:3]):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert np.isfinite(outs[0][0]).all() and np.abs(outs[0][0]).max() < 10


@pytest.mark.parametrize("max_fb", [0, 2], ids=["nofeedback", "feedback"])
def test_rank_input_without_any_pair_is_a_no_op(max_fb, tmp_path):
    """No user has a negative: the generator emits empty blocks (apex_svd_data.cpp:949), a pass changes nothing."""
    blocks = cases.rank_blocks(40, 60, 50, 8, 3, max_fb=max_fb)
    for b in blocks:
        b.data.row_label[:] = 1.0
    src = str(tmp_path / "train.buffer")
    D.write_ugroup_buffer(src, blocks)
    t = sa.Trainer(1, 3)
    t.seed(10)
    for k, v in cases.RANK_E2E_CONF:
        t.set_param(k, v)
    t.init_model()
    t.init_trainer()
    before = [t.view(v).copy() for v in ("W_user", "W_item", "i_bias", "g_bias", "W_ufeedback")]
    ds = t.dataset_from_rank_buffer_file(src)
    assert ds.num_row == 0
    t.train_dataset(ds)
    t.finish_round()
    for a, v in zip(before, ("W_user", "W_item", "i_bias", "g_bias", "W_ufeedback")):
        np.testing.assert_array_equal(a.view(np.uint32), t.view(v).view(np.uint32))
    ds.close()
    t.close()


def test_prefetched_rank_passes_train_the_same_bytes(tmp_path):
    """Drawing pass r+1 on the background thread while the device trains pass r changes nothing but the wall time."""
    src = str(tmp_path / "train.buffer")
    D.write_ugroup_buffer(src, cases.rank_blocks(300, 60, 50, 8, 12, max_rows=12, max_fb=0))
    plain, _ = _train_resident(tmp_path, src, cases.RANK_E2E_CONF, 3)
    t = sa.Trainer(1, 3)
    t.seed(10)
    for k, v in cases.RANK_E2E_CONF:
        t.set_param(k, v)
    t.init_model()
    t.init_trainer()
    p = str(tmp_path / "pf.model")
    ds = t.dataset_from_rank_buffer_file(src)
    for r in range(3):
        t.set_round(r)
        t.train_dataset(ds)                      # asynchronous
        if r + 1 < 3:
            t.rank_prefetch_buffer_file(src)     # host draws the next pass meanwhile
        t.finish_round()
        t.save_model(p)
        assert open(p, "rb").read() == plain[r + 1], "round %d differs" % r
        ds.close()
        if r + 1 < 3:
            ds = t.dataset_from_rank_buffer_file(src)
    t.close()
