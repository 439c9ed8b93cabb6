"""GPU parity of the ranker and the on-device evaluator: svdf_ranker_* against tests/golden/ranker.npz (compiled
reference) and the C oracle -- the int results are identical, including sections whose scores tie (finished by the
reference's own sort on the host); svdf_eval_dataset against the reference's long double accumulator."""
import numpy as np
import pytest

import cases
import svdfeature_amd as sa
from oracle import oracle
from test_ranker import GOLD_PATH, RANK_CASES, run_ranker, trained_model, user_group_lines_before_any_block

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", list(RANK_CASES))
def test_ranker_matches_reference_golden(name, tmp_path):
    gold = np.load(GOLD_PATH)
    got = run_ranker(lambda f: sa.Ranker(f, 0), name, str(tmp_path))
    np.testing.assert_array_equal(got, gold[name])


@pytest.mark.parametrize("k", [5, 64, 128, 300])
@pytest.mark.parametrize("top_k", [0, 10])
def test_ranker_large_item_set_matches_the_oracle(k, top_k, tmp_path):
    """2000 candidates x 40 user sections at several factor widths (lane groups and wide rows), top_k and position mode."""
    nu, ni, ng = 300, 2000, 5
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_global=ng, num_factor=k, ui_init_sigma=0.1, wd_global=0.001)
    t = oracle.OracleTrainer("port", 0, 0)
    t.seed(3)
    for kk, v in conf:
        t.set_param(kk, v)
    t.init_model()
    t.init_trainer()
    t.update_batch(cases.sparse_feature_rows(3000, nu, ni, ng, 11))
    path = str(tmp_path / "m.model")
    t.save_model(path)
    items, sections = cases.ranker_stream(2000, 40, nu, ni, ng, seed=k)
    outs = []
    for mk in (lambda: oracle.OracleRanker("port", 0, 0), lambda: sa.Ranker(0, 0)):
        r = mk()
        r.set_param("top_k", str(top_k))
        r.load_model(path)
        r.init_ranker(items.num_row)
        r.process_rows(items)
        outs.append(np.concatenate([r.process_rows(s) for s in sections]))
        if isinstance(r, sa.Ranker):
            assert r.counter(0) == 40 and r.counter(1) == 0
    np.testing.assert_array_equal(outs[0], outs[1])


@pytest.mark.parametrize("top_k", [1, 2047, 2048, 3500, 3990])
def test_ranker_long_top_k_prefixes(top_k, tmp_path):
    """top_k + 1 <= 2048 goes through the radix selection (three digit passes, append, one-workgroup sort), longer prefixes
    through the full device sort; 4000 candidates of which some are banned, top_k up to all ranked candidates but ten."""
    nu, ni = 100, 4000
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=20, ui_init_sigma=0.3)
    t = oracle.OracleTrainer("port", 0, 0)
    t.seed(9)
    for kk, v in conf:
        t.set_param(kk, v)
    t.init_model()
    t.init_trainer()
    path = str(tmp_path / "m.model")
    t.save_model(path)
    items = sa.CSRData.from_rows([(0.0, [], [], [(c, 1.0)]) for c in range(ni)])
    secs = sa.CSRData.from_rows([row for u in range(6) for row in ((2.0, [], [(u * 7, 1.0)], []), (-1.0, [], [(u, 1.0), (u + 50, 1.0)], []),
                                                                     (4.0, [], [], []))])
    outs = []
    for mk in (lambda: oracle.OracleRanker("port", 0, 0), lambda: sa.Ranker(0, 0)):
        r = mk()
        r.set_param("top_k", str(top_k))
        r.load_model(path)
        r.init_ranker(ni)
        r.process_rows(items)
        outs.append(r.process_rows(secs))
        if isinstance(r, sa.Ranker):
            assert r.counter(1) == 0   # distinct random rows: no tie, nothing finished on the host
    assert len(outs[0]) == 6 * top_k
    np.testing.assert_array_equal(outs[0], outs[1])


@pytest.mark.skipif(not oracle.have_reference(), reason="compiled reference (oracle/_ref) not present")
def test_ranker_tied_scores_follow_the_reference_sort(tmp_path):
    """Candidates that are copies of one another score exactly the same: their order is whatever std::sort makes of the
    reference's entry vector.  The engine detects the tie on the device and finishes the section with that very sort."""
    nu, ni = 40, 30
    path, extra, _ = trained_model(str(tmp_path), 0, 12, False)
    rows = [(0.0, [], [], [(int(c % 7), 1.0)]) for c in range(60)]   # 60 candidates, only 7 distinct items
    items = sa.CSRData.from_rows(rows)
    sec = sa.CSRData.from_rows([(2.0, [], [(3, 1.0)], []), (1.0, [], [(0, 1.0), (8, 1.0), (20, 1.0)], []), (4.0, [], [], [])])
    for top_k in (0, 9):
        outs = []
        for mk in (lambda: oracle.OracleRanker("reference", 0, 0), lambda: sa.Ranker(0, 0)):
            r = mk()
            r.set_param("top_k", str(top_k))
            r.load_model(path)
            r.init_ranker(60)
            r.process_rows(items)
            outs.append(r.process_rows(sec))
            if isinstance(r, sa.Ranker):
                assert r.counter(1) == 1   # finished on the host
        np.testing.assert_array_equal(outs[0], outs[1])


@pytest.mark.parametrize("top_k", [0, 7])
@pytest.mark.parametrize("spec", [False, True])
def test_ranker_bulk_rows_pipelined_equals_line_by_line(top_k, spec, tmp_path):
    """svdf_ranker_process_rows keeps up to 8 user sections in flight on the device; its results must be those of one
    svdf_ranker_process_csr call per line, in the same order: 100 sections (bans, special samples, duplicate candidates =
    tied scores that finish on the host, candidates arriving between sections) in ONE call against line-by-line calls
    and against the C oracle."""
    nu, ni, ng = 200, 1500, 4
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_global=ng, num_factor=48, ui_init_sigma=0.1, wd_global=0.001)
    t = oracle.OracleTrainer("port", 0, 0)
    t.seed(5)
    for kk, v in conf:
        t.set_param(kk, v)
    t.init_model()
    t.init_trainer()
    t.update_batch(cases.sparse_feature_rows(3000, nu, ni, ng, 12))
    path = str(tmp_path / "m.model")
    t.save_model(path)
    items, sections = cases.ranker_stream(1200, 100, nu, ni, ng, seed=17 + top_k, spec=spec)
    # 40 more candidates arrive after the 50th section; with the compiled reference at hand they duplicate earlier ones (exactly
    # tied scores: the order inside a tie is libstdc++'s std::sort's, which only the reference itself and the engine's host
    # fallback reproduce -- the C port's sort is not bound to it)
    kind = "reference" if oracle.have_reference() else "port"
    late = sa.CSRData.from_rows([(0.0, [], [], [(int(c % 11) if kind == "reference" else 1200 + c, 1.0)]) for c in range(40)])
    stream = sa.CSRData.concat([items] + sections[:50] + [late] + sections[50:])
    outs, hosted = {}, {}
    for name in ("oracle", "lines", "bulk"):
        r = oracle.OracleRanker(kind, 0, 0) if name == "oracle" else sa.Ranker(0, 0)
        r.set_param("top_k", str(top_k))
        r.load_model(path)
        r.init_ranker(items.num_row + late.num_row)
        if name == "bulk":
            outs[name] = r.process_rows(stream)
        else:
            outs[name] = np.concatenate([r.process(*stream.row(i)) for i in range(stream.num_row)])
        if name != "oracle":
            assert r.counter(0) == 100
            hosted[name] = r.counter(1)
            # without special samples the bulk call scores a TILE of sections per pass over the candidate matrix (positions and top_k alike)
            assert (r.counter(3) >= 3) == (name == "bulk" and not spec), r.counter(3)
    np.testing.assert_array_equal(outs["lines"], outs["oracle"])
    np.testing.assert_array_equal(outs["bulk"], outs["oracle"])
    assert hosted["bulk"] == hosted["lines"]   # the same sections needed the reference's sort either way


def test_ranker_tiles_equal_one_pass_per_section(tmp_path):
    """k_rank_score_tile (up to 32 user sections per pass over the candidate matrix) against one pass per section (amd:rank_tile = 0) and the
    C oracle: sections with 0 ... 6 positives and bans each, k = 5, 64, 300 (ragged tail / full rows / wide rows), odd tile remainders
    (every kernel width: 4 / 8 / 16 / 32 sections); positions mode and top_k mode -- short prefixes through the wave minima
    (k_rank_tile_select: fewer minima than the prefix at 300 candidates, 9 000 candidates), long ones (top_k = 40) through the radix
    selection of all sections of a tile in one set of launches"""
    for k, nsec, top_k, cand in ((5, 37, 0, 700), (64, 300, 0, 700), (300, 21, 0, 700), (64, 300, 10, 700), (5, 37, 1, 700), (128, 45, 40, 700),
                                 (64, 43, 10, 300), (32, 75, 31, 9000), (128, 70, 0, 9000), (16, 99, 3, 2049)):
        nu, ni, ng = 150, max(900, cand), 3
        conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_global=ng, num_factor=k, ui_init_sigma=0.1)
        t = oracle.OracleTrainer("port", 0, 0)
        t.seed(k)
        for kk, v in conf:
            t.set_param(kk, v)
        t.init_model()
        t.init_trainer()
        path = str(tmp_path / ("m%d.model" % k))
        t.save_model(path)
        items, sections = cases.ranker_stream(cand, nsec, nu, ni, ng, seed=k, spec=False)
        stream = sa.CSRData.concat([items] + sections)
        outs = {}
        for name, tile in (("oracle", None), ("tiles", 1), ("single", 0)):
            r = oracle.OracleRanker("port", 0, 0) if name == "oracle" else sa.Ranker(0, 0)
            if tile is not None:
                r.set_param("amd:rank_tile", str(tile))
            r.set_param("top_k", str(top_k))
            r.load_model(path)
            r.init_ranker(items.num_row)
            outs[name] = r.process_rows(stream) if name != "oracle" else np.concatenate([r.process(*stream.row(i)) for i in range(stream.num_row)])
            if name == "tiles":
                assert r.counter(3) >= nsec // 32
            if name == "single":
                assert r.counter(3) == 0
        np.testing.assert_array_equal(outs["tiles"], outs["oracle"])
        np.testing.assert_array_equal(outs["single"], outs["oracle"])


def test_ranker_initial_feedback_is_user_row_zero(tmp_path):
    a = user_group_lines_before_any_block(lambda f: oracle.OracleRanker("port", f, 0), str(tmp_path))
    b = user_group_lines_before_any_block(lambda f: sa.Ranker(f, 0), str(tmp_path))
    np.testing.assert_array_equal(a, b)


def test_ranker_errors():
    r = sa.Ranker(0, 0)
    with pytest.raises(sa.SvdfError, match="init_ranker has not been called"):
        r.process(0.0, 0, 0, 1, np.array([1], np.uint32), np.array([1], np.float32))


@pytest.mark.parametrize("kind", ["triples", "blocks", "csr"])
def test_eval_dataset_matches_the_reference_accumulator(kind):
    """svdf_eval_dataset: squared errors summed on the device; equal to the reference's sequential long double sum over the
    same predictions to 1e-12 relative, for every dataset kind; RMSE = sqrt(sum / count)."""
    nu, ni, ng = 500, 300, 6
    if kind == "triples":
        conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=32)
        t = sa.Trainer(0, 0)
    elif kind == "csr":
        conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_global=ng, num_factor=20, wd_global=0.001)
        t = sa.Trainer(0, 0)
    else:
        conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=16, num_ufeedback=ni, wd_ufeedback=0.004, ufeedback_init_sigma=0.01)
        t = sa.Trainer(1, 0)
    t.seed(10)
    for k, v in conf:
        t.set_param(k, v)
    t.init_model()
    t.init_trainer()
    if kind == "triples":
        u, i, r = cases.planted_triples(200_000, nu, ni, seed=1)
        ds = t.dataset_from_triples(u, i, r)
        labels = r
    elif kind == "csr":
        d = cases.sparse_feature_rows(5000, nu, ni, ng, 5)
        ds = t.dataset_from_csr(d)
        labels = d.row_label
    else:
        blocks = cases.user_blocks(400, nu, ni, ni, seed=6, max_rows=12, max_fb=8)
        ds = t.dataset_from_blocks(blocks)
        labels = np.concatenate([b.data.row_label for b in blocks])
    t.train_dataset(ds)
    pred = t.predict_dataset(ds)
    for scale in (1.0, 0.2):
        ss, cnt = t.eval_dataset(ds, scale)
        ref = oracle.sum_sq_err(pred, labels, scale)
        assert cnt == len(labels)
        assert abs(ss - ref) <= 1e-12 * ref, (ss, ref)


def test_ranker_edge_cases_match_the_oracle(tmp_path):
    """all candidates banned but one, a positive that is also the only ranked candidate, special samples overriding one another,
    candidates added between user sections, a section without positives, empty feature lines"""
    path, extra, (nu, ni, ng) = trained_model(str(tmp_path), 0, 12, False)
    R = sa.CSRData.from_rows
    items1 = R([(0.0, [], [], [(j % ni, 1.0)]) for j in range(6)])
    items2 = R([(0.0, [(1, 0.5)], [], [(7, 1.0), (9, 0.25)]), (0.0, [], [], [])])   # the second one has no feature at all
    secs = [
        R([(2.0, [], [(3, 1.0)], []), (-1.0, [], [(0, 1), (1, 1), (2, 1), (3, 1), (4, 1)], []), (1.0, [], [(5, 1.0)], []), (4.0, [], [], [])]),
        R([(2.0, [], [], []), (1.0, [], [(2, 1.0), (4, 1.0)], []), (3.0, [(0, 1.0)], [(2, 1.0)], [(1, 0.5)]), (3.0, [], [(2, 1.0)], []), (4.0, [], [], [])]),
        R([(2.0, [], [(5, 1.0), (6, 0.5)], []), (4.0, [], [], [])]),
    ]
    outs = []
    for mk in (lambda: oracle.OracleRanker("port", 0, 0), lambda: sa.Ranker(0, 0)):
        r = mk()
        r.load_model(path)
        r.init_ranker(8)
        res = [r.process_rows(items1), r.process_rows(secs[0]), r.process_rows(secs[1]), r.process_rows(items2), r.process_rows(secs[2]),
               r.process_rows(secs[1])]
        outs.append(np.concatenate(res))
    np.testing.assert_array_equal(outs[0], outs[1])
    assert outs[0].size == 1 + 2 + 0 + 2
    g = sa.Ranker(0, 0)
    g.set_param("top_k", "5")
    g.load_model(path)
    g.init_ranker(3)
    g.process_rows(R([(0.0, [], [], [(1, 1.0)]), (0.0, [], [], [(2, 1.0)])]))
    with pytest.raises(sa.SvdfError, match="k can not exceed candidate size"):
        g.process_rows(R([(2.0, [], [(1, 1.0)], []), (4.0, [], [], [])]))
    with pytest.raises(sa.SvdfError, match="item instance exceed specified item set size"):
        g.process_rows(R([(0.0, [], [], [(1, 1.0)]), (0.0, [], [], [(2, 1.0)])]))
    with pytest.raises(sa.SvdfError, match="sample item index exceed bound"):
        g.process_rows(R([(2.0, [], [(1, 1.0)], []), (1.0, [], [(7, 1.0)], [])]))
