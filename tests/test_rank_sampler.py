"""Rank-pair input (the reference's input_type = 2) on the host: the product's restatement of
PairwiseRankGenerator (svdf_pairgen.cpp) must draw, from libc rand() after the same srand, exactly the pairs the
reference's own generator draws.  Expected outputs: tests/golden/rank_input.npz, produced by the compiled reference
(tests/golden/make_rank_golden.py); where oracle/_ref is present the reference's generator is also run live."""
import os
import subprocess

import numpy as np
import pytest

import cases
import svdfeature_amd as sa
from svdfeature_amd import data as D

GOLD = np.load(os.path.join(cases.GOLDEN, "rank_input.npz"))
REF_DUMP = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "ref_pairgen_dump")


def _sample(tmp_path, blocks, keys, seed, rounds):
    src = str(tmp_path / "in.buffer")
    D.write_ugroup_buffer(src, blocks)
    t = sa.Trainer(1, 3, device=-2)
    t.seed(seed)
    for k, v in keys.items():
        t.set_param(k, v)
    got, rows = [], 0
    for r in range(rounds):
        out = str(tmp_path / ("pass%d.buffer" % r))
        rows += t.rank_sample_buffer_file(src, out)
        got += D.read_ugroup_buffer(out)
    t.close()
    return src, got, rows


@pytest.mark.parametrize("n", range(len(cases.RANK_SAMPLER_CASES)), ids=[c[0] for c in cases.RANK_SAMPLER_CASES])
def test_sampler_draws_the_reference_pairs(n, tmp_path):
    name, graded, keys = cases.RANK_SAMPLER_CASES[n]
    blocks = cases.rank_blocks(200, 60, 50, 8, 500 + n, graded)
    _, got, rows = _sample(tmp_path, blocks, keys, cases.RANK_SAMPLER_SEED, cases.RANK_SAMPLER_ROUNDS)
    assert len(got) == cases.RANK_SAMPLER_ROUNDS * len(blocks)
    assert rows == int(GOLD["sampler/%s/num_row" % name]) == sum(b.data.num_row for b in got)
    if n == 0:   # the vectors themselves, so a mismatch shows where
        np.testing.assert_array_equal(np.concatenate([b.data.row_label for b in got]).view(np.uint32), GOLD["sampler/%s/label" % name].view(np.uint32))
        np.testing.assert_array_equal(np.concatenate([np.diff(b.data.row_ptr) for b in got]), GOLD["sampler/%s/row_len" % name])
        np.testing.assert_array_equal(np.concatenate([b.data.feat_index for b in got]), GOLD["sampler/%s/index" % name])
        np.testing.assert_array_equal(np.concatenate([b.data.feat_value for b in got]).view(np.uint32), GOLD["sampler/%s/value" % name].view(np.uint32))
    assert cases.blocks_digest(got) == str(GOLD["sampler/%s/md5" % name])
    # the feedback part and the tag of every block pass through untouched (apex_svd_data.cpp:999-1019)
    for r in range(cases.RANK_SAMPLER_ROUNDS):
        for a, b in zip(blocks, got[r * len(blocks):(r + 1) * len(blocks)]):
            assert a.extend_tag == b.extend_tag
            np.testing.assert_array_equal(a.index_ufeedback, b.index_ufeedback)


def test_pairs_are_signed_merges_of_one_positive_and_one_negative(tmp_path):
    """Structure of a generated row (method 0): label 1, the user entries of the positive without its ~0 values, and an
    item section that is the positive's entries minus the negative's, merged by index."""
    blocks = cases.rank_blocks(120, 60, 50, 8, 77)
    _, got, _ = _sample(tmp_path, blocks, {}, 3, 1)
    seen = 0
    for src, out in zip(blocks, got):
        cand = [src.data.row(i) for i in range(src.data.num_row)]
        pos = [c for c in cand if c[0] >= 0.8]
        neg = [c for c in cand if c[0] <= 0.0]
        if not pos or not neg:
            assert out.data.num_row == 0
            continue
        assert out.data.num_row == len(neg)   # rank_sample_num unset: one pair per negative (apex_svd_data.cpp:954-961)

        def signed(c, sign):
            _, ng, nu, ni, idx, val = c
            return {int(i): sign * float(v) for i, v in zip(idx[ng + nu:], val[ng + nu:])}
        wanted = []
        for p in pos:
            for q in neg:
                m = signed(p, 1.0)
                for i, v in signed(q, -1.0).items():
                    m[i] = np.float32(m[i]) + np.float32(v) if i in m else v
                wanted.append(sorted((i, float(np.float32(v))) for i, v in m.items()))
        for r in range(out.data.num_row):
            label, ng, nu, ni, idx, val = out.data.row(r)
            assert label == 1.0
            assert np.all(np.abs(val[ng:ng + nu]) > 1e-6)
            assert sorted(zip(idx[ng + nu:].tolist(), val[ng + nu:].tolist())) in wanted
            assert np.all(np.diff(idx[ng + nu:].astype(np.int64)) > 0)
            seen += 1
    assert seen > 100


def test_sampler_errors_carry_the_reference_texts(tmp_path):
    blocks = cases.rank_blocks(5, 60, 50, 8, 1)
    src = str(tmp_path / "in.buffer")
    D.write_ugroup_buffer(src, blocks)
    t = sa.Trainer(1, 3, device=-2)
    t.set_param("rank_sample_method", "2")
    with pytest.raises(sa.SvdfError, match="unkown rank sample method"):   # apex_svd_data.cpp:1008, spelling included
        t.rank_sample_buffer_file(src, str(tmp_path / "o"))
    t = sa.Trainer(1, 3, device=-2)
    t.set_param("rank_sample_gap", "0")
    with pytest.raises(sa.SvdfError, match="must set rank_sample_gap"):    # apex_svd_data.cpp:987
        t.rank_sample_buffer_file(src, str(tmp_path / "o"))
    t = sa.Trainer(0, 3, device=-2)   # rank pairs only exist for the user-group format (svd_feature.cpp:129-133)
    with pytest.raises(sa.SvdfError, match="user-group format"):
        t.dataset_from_rank_buffer_file(src)
    t = sa.Trainer(1, 3, device=-2)
    with pytest.raises(sa.SvdfError, match="can not open"):
        t.rank_sample_buffer_file(str(tmp_path / "missing"), str(tmp_path / "o"))


@pytest.mark.skipif(not os.path.exists(REF_DUMP), reason="oracle/_ref/ref_pairgen_dump is built in the build container only")
@pytest.mark.parametrize("seed", [1, 2024])
def test_sampler_against_the_reference_generator_run_live(seed, tmp_path):
    keys = {"rank_sample_method": str(seed % 2), "rank_sample_num": "7"}
    blocks = cases.rank_blocks(300, 40, 30, 5, seed, graded=True, max_rows=20)
    src, got, _ = _sample(tmp_path, blocks, keys, seed, 3)
    ref_out = str(tmp_path / "ref.buffer")
    subprocess.check_call([REF_DUMP, src, ref_out, str(seed), "3"] + ["%s=%s" % kv for kv in keys.items()],
                          cwd=str(tmp_path), stdout=subprocess.DEVNULL)
    assert cases.blocks_digest(D.read_ugroup_buffer(ref_out)) == cases.blocks_digest(got)


def test_prefetched_passes_are_the_same_draws(tmp_path):
    """svdf_rank_prefetch_buffer_file draws the next pass on a background thread; taken by the next call for the same
    file it is the pass that call would have drawn (golden digest of two consecutive passes)."""
    name, graded, keys = cases.RANK_SAMPLER_CASES[3]
    blocks = cases.rank_blocks(200, 60, 50, 8, 503, graded)
    src = str(tmp_path / "in.buffer")
    D.write_ugroup_buffer(src, blocks)
    t = sa.Trainer(1, 3, device=-2)
    t.seed(cases.RANK_SAMPLER_SEED)
    for k, v in keys.items():
        t.set_param(k, v)
    got = []
    t.rank_prefetch_buffer_file(src)
    for r in range(cases.RANK_SAMPLER_ROUNDS):
        out = str(tmp_path / ("pass%d.buffer" % r))
        t.rank_sample_buffer_file(src, out)          # takes the prefetched pass
        if r + 1 < cases.RANK_SAMPLER_ROUNDS:
            t.rank_prefetch_buffer_file(src)
        got += D.read_ugroup_buffer(out)
    assert cases.blocks_digest(got) == str(GOLD["sampler/%s/md5" % name])
    # a second prefetch before the first is used, and a pick-up for another file, are errors
    t.rank_prefetch_buffer_file(src)
    with pytest.raises(sa.SvdfError, match="still waiting"):
        t.rank_prefetch_buffer_file(src)
    with pytest.raises(sa.SvdfError, match="another file"):
        t.rank_sample_buffer_file(str(tmp_path / "pass0.buffer"), str(tmp_path / "x"))
    t.rank_prefetch_buffer_file(str(tmp_path / "missing"))   # errors of the background pass surface at the pick-up
    with pytest.raises(sa.SvdfError, match="can not open"):
        t.rank_sample_buffer_file(str(tmp_path / "missing"), str(tmp_path / "x"))
    t.close()


@pytest.mark.skipif(not os.path.exists(REF_DUMP), reason="oracle/_ref/ref_pairgen_dump is built in the build container only")
def test_sampler_random_settings_against_the_reference_generator(tmp_path):
    """40 random sampler settings x random candidate sets, two passes each, against the reference's generator run live."""
    rng = np.random.default_rng(99)
    for case in range(40):
        keys = {"rank_sample_method": str(int(rng.integers(0, 2)))}
        if rng.integers(0, 2):
            keys["rank_sample_num"] = str(int(rng.integers(1, 12)))
        if rng.integers(0, 3) == 0:
            keys["rank_sample_max"] = str(int(rng.integers(1, 6)))
        if rng.integers(0, 3) == 0:
            keys["rank_sample_pointwise"] = "1"
        if rng.integers(0, 2):
            keys["rank_sample_gap"] = str(float(rng.choice([0.0001, 0.5, 1.0, 2.5])))
        graded = bool(rng.integers(0, 2))
        if graded and rng.integers(0, 2):
            keys["pos_sample_lowerb"] = str(float(rng.choice([2, 3.5, 4])))
            keys["neg_sample_upperb"] = str(float(rng.choice([1, 2, 3])))
        seed = int(rng.integers(1, 1 << 20))
        d = tmp_path / ("c%d" % case)
        d.mkdir()
        blocks = cases.rank_blocks(int(rng.integers(1, 120)), 40, 30, int(rng.choice([0, 5])), seed, graded=graded,
                                   max_rows=int(rng.integers(1, 30)), max_fb=int(rng.integers(0, 4)))
        src, got, _ = _sample(d, blocks, keys, seed, 2)
        ref_out = str(d / "ref.buffer")
        subprocess.check_call([REF_DUMP, src, ref_out, str(seed), "2"] + ["%s=%s" % kv for kv in keys.items()],
                              cwd=str(d), stdout=subprocess.DEVNULL)
        assert cases.blocks_digest(D.read_ugroup_buffer(ref_out)) == cases.blocks_digest(got), (case, keys)


def test_restated_std_sort_equals_the_library_sort_position_for_position():
    """svdf_stdsort.h restates libstdc++'s std::sort (introsort) because sample_cmp (apex_svd_data.cpp:920-944) picks rows by position
    after an unstable sort of few distinct labels.  Same permutation as the library's own std::sort for every size class: <= 16
    (insertion sort only), a few partitions, thousands of rows, few / many distinct keys, already sorted, reversed, organ pipe and
    median-of-three killer inputs (the latter run into the heap-sort fallback at the depth limit)."""
    rng = np.random.default_rng(5)
    cases_ = []
    for n in list(range(0, 40)) + [63, 64, 65, 100, 257, 1000, 4097, 20000]:
        for distinct in (1, 2, 5, 50, 10 ** 6):
            cases_.append(rng.integers(0, distinct, n).astype(np.float32))
    for n in (17, 33, 200, 3000):
        cases_.append(np.arange(n, dtype=np.float32))
        cases_.append(np.arange(n, dtype=np.float32)[::-1].copy())
        cases_.append(np.concatenate([np.arange(n // 2), np.arange(n - n // 2)[::-1]]).astype(np.float32))

    def killer(n):   # Musser's median-of-3 killer sequence: quadratic for plain quicksort, forces the introsort depth limit
        k = n // 2
        a = np.zeros(n, np.float32)
        for i in range(1, k + 1):
            if i % 2 == 1:
                a[i - 1] = i
                a[i] = k + i
            a[k + i - 1] = 2 * i
        return a
    for n in (64, 512, 4096, 30000):
        cases_.append(killer(n))
    for lab in cases_:
        mine, lib = sa.debug_sort_labels(lab)
        np.testing.assert_array_equal(mine, lib)
        assert np.all(np.diff(lab[mine]) >= 0) and sorted(mine.tolist()) == list(range(len(lab)))


def test_threaded_ranker_sort_equals_std_sort_on_the_entry_struct():
    """Sections whose scores tie are finished by the reference's own ordering step -- std::sort over the Entry vector
    (apex_svd_base.h:617-624, :767) -- with its partitions spread over host threads (svdf_ranker.cpp: host_parallel_sort_scores).  The
    permutation (tied scores included) must be the library's: few distinct values, all equal, sorted / reversed inputs, Musser's
    median-of-3 killer (heap-sort branch), sizes around the sharing grain, 1 ... 16 threads."""
    rng = np.random.default_rng(5)
    cases_ = []
    for n in (0, 1, 2, 15, 16, 17, 100, 2047, 2048, 4097, 30000, 100000):
        cases_.append(rng.normal(size=n).astype(np.float32))
        cases_.append(rng.integers(0, 3, size=n).astype(np.float32))
        cases_.append(np.round(rng.normal(size=n), 1).astype(np.float32))
        cases_.append(np.zeros(n, np.float32))
        cases_.append(np.arange(n, dtype=np.float32))
        cases_.append(np.arange(n, dtype=np.float32)[::-1].copy())

    def killer(n):
        k = n // 2
        a = np.zeros(n, np.float32)
        for i in range(1, k + 1):
            if i % 2 == 1:
                a[i - 1] = i
                a[i] = k + i
            a[k + i - 1] = 2 * i
        return -a   # (descending comparator)
    for n in (4096, 30000, 100000):
        cases_.append(killer(n))
    for j, sc in enumerate(cases_):
        for threads in ((1, 8) if j % 3 else (2, 5, 16)):
            mine, lib = sa.debug_sort_scores(sc, threads)
            np.testing.assert_array_equal(mine, lib)
