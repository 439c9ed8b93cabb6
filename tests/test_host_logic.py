"""CPU-side tests of the product's host logic (no GPU): the C-ABI library loads and exports every
symbol of include/svdfeature_amd.h, rand_init / model-file I/O are byte-identical to the reference
formats, the conflict-free batch scheduler is correct, and the engine refuses to run without a GPU
instead of falling back to anything."""
import os
import re

import numpy as np
import pytest

import cases
import scenarios
import svdfeature_amd as sa
from oracle import oracle

GOLD = np.load(os.path.join(cases.GOLDEN, "scenarios.npz"))


def test_library_loads_and_exports_every_declared_symbol():
    lib = sa.load_library()
    header = open(sa.HEADER_PATH).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    names = sorted(set(re.findall(r"\b(svdf_[a-z_0-9]+)\s*\(", header)))
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), "symbol %s declared in include/svdfeature_amd.h is not exported" % n
    assert b"gfx950" in lib.svdf_version()


def test_no_silent_cpu_fallback():
    """On a box without a GPU the compute handle must fail loudly (and on a GPU box the host-only
    handle must refuse compute)."""
    if sa.device_count() == 0:
        with pytest.raises(sa.SvdfError, match="no HIP device"):
            sa.Trainer(0, 0)
    t = sa.Trainer(0, 0, device=-2)
    for k, v in cases.conf_with(cases.BASICMF_CONF, num_user=5, num_item=5, num_factor=4):
        t.set_param(k, v)
    t.init_model()
    t.init_trainer()
    d = sa.CSRData.from_triples([1], [2], [3.0])
    with pytest.raises(sa.SvdfError, match="needs a GPU"):
        t.update_batch(d)
    with pytest.raises(sa.SvdfError, match="needs a GPU"):
        t.predict_batch(d)


@pytest.mark.parametrize("name", ["basicmf_ml100k_k16", "sparse_k10_tail", "sparse_logistic", "svdpp_random", "implicit_example",
                                  "sparse_nonneg_decaylr", "sparse_alias_keys", "svdpp_common_latent", "common_latent_triples"])
def test_rand_init_and_model_file_match_reference_bytes(name, tmp_path):
    """init_model (libc rand, Marsaglia polar, row-major draw order, base_score transform) followed by
    save_model reproduces the reference's 0000.model byte for byte (golden md5 from the compiled
    reference)."""
    import hashlib
    s = scenarios.SCENARIOS[name](str(tmp_path))
    t = sa.Trainer(s["format_type"], s["active_type"], device=-2)
    t.seed(scenarios.SEED)
    for k, v in s["conf"]:
        if k in ("feature_user", "feature_item"):
            continue
        t.set_param(k, v)
    t.init_model()
    p = str(tmp_path / "0000.model")
    t.save_model(p)
    assert hashlib.md5(open(p, "rb").read()).hexdigest() == str(GOLD[name + "/model0_md5"])
    # load -> save round trip through a second handle
    t2 = sa.Trainer(s["format_type"], s["active_type"], device=-2)
    t2.load_model(p)
    p2 = str(tmp_path / "copy.model")
    t2.save_model(p2)
    assert open(p, "rb").read() == open(p2, "rb").read()
    # views of the host model agree with the oracle's
    o = oracle.OracleTrainer("port", s["format_type"], s["active_type"])
    o.load_model(p)
    for v in ("W_user", "W_item", "u_bias", "i_bias", "g_bias", "W_ufeedback"):
        a, b = t2.view(v), o.view(v)
        if b is None:
            assert a is None
        else:
            np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32))


def test_model_file_shape_mismatch_is_an_error(tmp_path):
    t = sa.Trainer(0, 0, device=-2)
    for k, v in cases.conf_with(cases.BASICMF_CONF, num_user=7, num_item=5, num_factor=4):
        t.set_param(k, v)
    t.init_model()
    p = str(tmp_path / "m.model")
    t.save_model(p)
    raw = bytearray(open(p, "rb").read())
    raw[4 + 1056] ^= 1   # corrupt x_max of u_bias
    open(p, "wb").write(bytes(raw))
    with pytest.raises(sa.SvdfError, match="shape"):
        sa.Trainer(0, 0, device=-2).load_model(p)


def test_parameter_errors_match_reference_messages():
    t = sa.Trainer(0, 0, device=-2)
    with pytest.raises(sa.SvdfError, match="can't give 0 as bound"):
        t.set_param("up:bound", "0")
    t = sa.Trainer(0, 0, device=-2)
    t.set_param("up:wd", "0.1")
    with pytest.raises(sa.SvdfError, match="setting must be exactly"):
        t.set_param("up:wd", "0.2")
    t = sa.Trainer(0, 1, device=-2)   # sigmoid link with base_score outside (0,1)
    for k, v in cases.BASICMF_CONF:
        t.set_param(k, v)
    with pytest.raises(sa.SvdfError, match="sigmoid range constrain"):
        t.init_model()
    t = sa.Trainer(0, 0, device=-2)
    for k, v in cases.conf_with(cases.BASICMF_CONF, reg_method=6):   # 0..5 exist (4, 5 = lazy decay), apex_svd_base.h:239,279
        t.set_param(k, v)
    t.init_model()
    with pytest.raises(sa.SvdfError, match="unknown reg_method"):
        t.init_trainer()
    t = sa.Trainer(0, 0, device=-2)
    for k, v in cases.conf_with(cases.BASICMF_CONF, reg_global=2):   # 0, 1, 4, 5 exist, apex_svd_base.h:192-207
        t.set_param(k, v)
    t.init_model()
    with pytest.raises(sa.SvdfError, match="unknown global decay method"):
        t.init_trainer()


def _resources_of(d, num_user, num_item):
    """resource ids the way the engine numbers them without feedback rows: user rows, item rows, globals"""
    res, ptr = [], [0]
    for r in range(d.num_row):
        label, ng, nu, ni, idx, val = d.row(r)
        res += [num_user + num_item + int(g) for g in idx[:ng]]
        res += [int(u) for u in idx[ng:ng + nu]]
        res += [num_user + int(i) for i in idx[ng + nu:]]
        ptr.append(len(res))
    return np.array(ptr, np.int64), np.array(res, np.uint32)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_scheduler_batches_are_conflict_free_ordered_and_equivalent(seed):
    nu, ni, ng = 30, 20, 6
    d = cases.sparse_feature_rows(500, nu, ni, ng, seed)
    ptr, res = _resources_of(d, nu, ni)
    order, level_ptr = sa.schedule_resources(ptr, res, nu + ni + ng)
    assert sorted(order.tolist()) == list(range(d.num_row))            # a permutation
    assert level_ptr[0] == 0 and level_ptr[-1] == d.num_row
    level_of = np.empty(d.num_row, np.int64)
    for l in range(len(level_ptr) - 1):
        batch = order[level_ptr[l]:level_ptr[l + 1]]
        assert len(batch) > 0
        assert np.all(np.diff(batch) > 0)                               # stable inside a batch
        seen = set()
        for r in batch:
            mine = set(res[ptr[r]:ptr[r + 1]].tolist())
            assert not (seen & mine), "two instances of one batch share a parameter row"
            seen |= mine
            level_of[r] = l
    # instances that share a resource keep their file order across batches
    last = {}
    for r in range(d.num_row):
        for x in set(res[ptr[r]:ptr[r + 1]].tolist()):
            if x in last:
                assert level_of[last[x]] < level_of[r]
            last[x] = r
    # greedy earliest placement: every instance in batch l>0 conflicts with something in batch l-1
    for l in range(1, len(level_ptr) - 1):
        prev = set()
        for r in order[level_ptr[l - 1]:level_ptr[l]]:
            prev |= set(res[ptr[r]:ptr[r + 1]].tolist())
        for r in order[level_ptr[l]:level_ptr[l + 1]]:
            assert prev & set(res[ptr[r]:ptr[r + 1]].tolist())
    # the point of it all: running the (sequential) oracle in batch order gives the SAME BYTES as
    # running it in file order
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_global=ng, num_factor=8, wd_global=0.01)
    outs = []
    for perm in (np.arange(d.num_row), order):
        t = oracle.OracleTrainer("port", 0, 0)
        t.seed(5)
        for k, v in conf:
            t.set_param(k, v)
        t.init_model()
        t.init_trainer()
        for r in perm:
            t.update_csr(*d.row(int(r)))
        outs.append([t.view(v).copy() for v in ("W_user", "W_item", "u_bias", "i_bias", "g_bias")])
    for a, b in zip(*outs):
        np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32))


def test_scheduler_edge_cases():
    order, level_ptr = sa.schedule_resources(np.zeros(1, np.int64), np.zeros(0, np.uint32), 4)
    assert len(order) == 0 and list(level_ptr) == [0]
    # every instance hits the same row: one instance per batch, file order
    n = 50
    order, level_ptr = sa.schedule_resources(np.arange(n + 1, dtype=np.int64), np.zeros(n, np.uint32), 1)
    assert list(order) == list(range(n)) and list(level_ptr) == list(range(n + 1))
    # no instance shares anything: a single batch
    order, level_ptr = sa.schedule_resources(np.arange(n + 1, dtype=np.int64), np.arange(n, dtype=np.uint32), n)
    assert list(level_ptr) == [0, n]


def test_buffer_file_reader_rejects_bad_files_before_touching_a_device(tmp_path):
    """svdf_dataset_from_buffer_file parses on the host first: missing / truncated files are reported with a
    host-only handle too (the parsed stream then needs a GPU: 'no silent fallback')."""
    t = sa.Trainer(0, 0, device=-2)
    for k, v in cases.conf_with(cases.BASICMF_CONF, num_factor=4):
        t.set_param(k, v)
    t.init_model()
    t.init_trainer()
    with pytest.raises(sa.SvdfError, match="can not open"):
        t.dataset_from_buffer_file(str(tmp_path / "nope.buffer"))
    tr, _ = cases.ml100k()
    good = str(tmp_path / "good.buffer")
    sa.data.write_csr_buffer(good, tr.slice_rows(0, 1500), 1000)
    raw = open(good, "rb").read()
    open(str(tmp_path / "cut.buffer"), "wb").write(raw[:-3])
    with pytest.raises(sa.SvdfError, match="truncated"):
        t.dataset_from_buffer_file(str(tmp_path / "cut.buffer"))
    with pytest.raises(sa.SvdfError):   # well-formed file, but this handle has no device to put it on
        t.dataset_from_buffer_file(good)
