"""ISVDRanker: the C oracle's restatement of SVDFeatureRanker (apex_svd_base.h:597-813) against the compiled
reference (create_svd_ranker of oracle/_ref/libsvdf_ref.so) and against tests/golden/ranker.npz written from it; the
evaluator's accumulator (svd_feature_infer.cpp:38-56)."""
import os

import numpy as np
import pytest

import cases
from oracle import oracle

GOLD_PATH = os.path.join(cases.GOLDEN, "ranker.npz")

RANK_CASES = {   # name: (format_type, top_k, side tables, num_factor)
    "positions_k12": (0, 0, False, 12),
    "top5_k12": (0, 5, False, 12),
    "positions_side_tables_k10": (0, 0, True, 10),
    "positions_user_group_k16": (1, 0, False, 16),
    "top3_user_group_k7": (1, 3, False, 7),
}


def trained_model(tmp, fmt, k, side):
    """a small model trained by the C oracle, saved to a file every ranker loads"""
    nu, ni, ng = 50, 40, 6
    extra = []
    if side:
        fu, fi = os.path.join(tmp, "fu.txt"), os.path.join(tmp, "fi.txt")
        cases.write_side_table(fu, nu - 5, nu, 3)
        cases.write_side_table(fi, ni, ni, 4)
        extra = [("feature_user", fu), ("feature_item", fi)]
    kw = dict(num_user=nu, num_item=ni, num_global=ng, num_factor=k, wd_global=0.002, learning_rate=0.02, ui_init_sigma=0.05)
    if fmt == 1:
        kw.update(num_ufeedback=ni, wd_ufeedback=0.004, ufeedback_init_sigma=0.05)
    conf = cases.conf_with(cases.BASICMF_CONF, **kw) + extra
    t = oracle.OracleTrainer("port", fmt, 0)
    t.seed(10)
    for kk, v in conf:
        t.set_param(kk, v)
    t.init_model()
    t.init_trainer()
    if fmt == 0:
        t.update_batch(cases.sparse_feature_rows(800, nu, ni, ng, 7))
    else:
        for b in cases.user_blocks(60, nu, ni, ni, 8):
            t.update_block(b)
    path = os.path.join(tmp, "rank.model")
    t.save_model(path)
    return path, extra, (nu, ni, ng)


def run_ranker(make, name, tmp):
    fmt, top_k, side, k = RANK_CASES[name]
    path, extra, (nu, ni, ng) = trained_model(tmp, fmt, k, side)
    items, sections = cases.ranker_stream(35, 12, nu, ni, ng, seed=len(name))
    r = make(fmt)
    for kk, v in extra + [("top_k", str(top_k))]:
        r.set_param(kk, v)
    r.load_model(path)
    r.init_ranker(items.num_row + 3)
    out = [r.process_rows(items)]
    assert out[0].size == 0
    from svdfeature_amd.data import PlusBlock
    for s, sec in enumerate(sections):
        if fmt == 1:   # user-grouped input: the section's rows in one block with the user's feedback list
            fb = np.sort(np.random.default_rng(s).choice(ni, size=3, replace=False)).astype(np.uint32)
            out.append(r.process_block(PlusBlock(fb, np.full(3, 0.5, np.float32), sec, 0)))
        else:
            out.append(r.process_rows(sec))
    r.close()
    return np.concatenate(out).astype(np.int32)


@pytest.mark.parametrize("name", list(RANK_CASES))
def test_oracle_ranker_matches_golden(name, tmp_path):
    gold = np.load(GOLD_PATH)
    got = run_ranker(lambda f: oracle.OracleRanker("port", f, 0), name, str(tmp_path))
    np.testing.assert_array_equal(got, gold[name])
    assert got.size > 0


@pytest.mark.skipif(not oracle.have_reference(), reason="compiled reference (oracle/_ref) not present")
@pytest.mark.parametrize("name", list(RANK_CASES))
def test_oracle_ranker_matches_live_reference(name, tmp_path):
    a = run_ranker(lambda f: oracle.OracleRanker("port", f, 0), name, str(tmp_path))
    b = run_ranker(lambda f: oracle.OracleRanker("reference", f, 0), name, str(tmp_path))
    np.testing.assert_array_equal(a, b)


def user_group_lines_before_any_block(make, tmp):
    """a user-group model whose user sections arrive as plain lines, no block yet: tmp_ufeedback is still the clone of
    W_user[0] made by init_ranker (apex_svd_base.h:680-682)"""
    path, extra, (nu, ni, ng) = trained_model(tmp, 1, 9, False)
    items, sections = cases.ranker_stream(30, 6, nu, ni, ng, seed=3)
    r = make(1)
    r.set_param("top_k", "0")
    r.load_model(path)
    r.init_ranker(items.num_row)
    out = [r.process_rows(items)] + [r.process_rows(s) for s in sections]
    r.close()
    return np.concatenate(out).astype(np.int32)


@pytest.mark.skipif(not oracle.have_reference(), reason="compiled reference (oracle/_ref) not present")
def test_oracle_ranker_initial_feedback_is_user_row_zero(tmp_path):
    a = user_group_lines_before_any_block(lambda f: oracle.OracleRanker("port", f, 0), str(tmp_path))
    b = user_group_lines_before_any_block(lambda f: oracle.OracleRanker("reference", f, 0), str(tmp_path))
    np.testing.assert_array_equal(a, b)
    assert a.size > 0


def test_rmse_accumulator_restatement():
    rng = np.random.default_rng(1)
    p, l = rng.uniform(1, 5, 100000).astype(np.float32), rng.integers(1, 6, 100000).astype(np.float32)
    s = oracle.sum_sq_err(p, l, 1.0)
    ref = float(np.sum(((p - l).astype(np.float32)).astype(np.float64) ** 2))
    assert abs(s - ref) <= 1e-9 * ref
    if oracle.have_reference():
        assert oracle.sum_sq_err(p, l, 0.5, "reference") == oracle.sum_sq_err(p, l, 0.5, "port")
