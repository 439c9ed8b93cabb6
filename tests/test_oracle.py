"""Pins the C oracle (oracle/svdf_oracle.c) to the reference.

1. against golden vectors generated from the COMPILED reference (tests/golden/scenarios.npz,
   written by tests/golden/make_golden.py): model files byte-identical (md5), predictions
   bit-identical;
2. against the compiled reference itself when oracle/_ref/libsvdf_ref.so is present
   (build container and, prebuilt, the GPU box);
3. against the reference tree's own fixtures for the path (demo/basicMF buffers, eg.pred.txt).
"""
import hashlib
import os

import numpy as np
import pytest

import cases
import scenarios
from oracle import oracle
from svdfeature_amd import data as D

GOLD = np.load(os.path.join(cases.GOLDEN, "scenarios.npz"))
FIX = os.path.join(cases.GOLDEN, "fixtures")


def port(f, a):
    return oracle.OracleTrainer("port", f, a)


def ref(f, a):
    return oracle.OracleTrainer("reference", f, a)


@pytest.mark.parametrize("name", list(scenarios.SCENARIOS))
def test_oracle_matches_golden(name):
    res = scenarios.run_scenario(name, port)
    dg = scenarios.digest(res)
    assert dg["model0_md5"] == str(GOLD[name + "/model0_md5"]), "initial model (rand_init) differs"
    assert dg["model_len"] == int(GOLD[name + "/model_len"])
    np.testing.assert_array_equal(dg["model_sample"].view(np.uint32), GOLD[name + "/model_sample"].view(np.uint32))
    assert dg["model_md5"] == str(GOLD[name + "/model_md5"]), "trained model file is not byte-identical"
    assert dg["pred_md5"] == str(GOLD[name + "/pred_md5"])
    assert abs(dg["rmse"] - float(GOLD[name + "/rmse"])) == 0.0


@pytest.mark.skipif(not oracle.have_reference(), reason="compiled reference (oracle/_ref) not present")
@pytest.mark.parametrize("name", ["basicmf_ml100k_k16", "sparse_side_tables", "sparse_logistic", "svdpp_random"])
@pytest.mark.parametrize("chunk", [None, 7])
def test_oracle_matches_live_reference(name, chunk):
    a = scenarios.run_scenario(name, port, chunk=chunk)
    b = scenarios.run_scenario(name, ref, chunk=chunk)
    assert a["model0"] == b["model0"]
    assert a["model"] == b["model"]
    np.testing.assert_array_equal(a["pred"].view(np.uint32), b["pred"].view(np.uint32))


@pytest.mark.skipif(not oracle.have_reference(), reason="compiled reference (oracle/_ref) not present")
def test_oracle_views_and_single_instance_calls_match_reference():
    """Per-instance update()/predict() entry points and raw views, not only whole batches."""
    d = cases.sparse_feature_rows(120, 20, 15, 5, 9)
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=20, num_item=15, num_global=5, num_factor=7, wd_global=0.01)
    trs = []
    for kind in ("port", "reference"):
        t = oracle.OracleTrainer(kind, 0, 0)
        t.seed(3)
        for k, v in conf:
            t.set_param(k, v)
        t.init_model()
        t.init_trainer()
        trs.append(t)
    for r in range(d.num_row):
        row = d.row(r)
        pa, pb = trs[0].predict_csr(*row), trs[1].predict_csr(*row)
        assert np.float32(pa).view(np.uint32) == np.float32(pb).view(np.uint32)
        for t in trs:
            t.update_csr(*row)
    for name in ("u_bias", "W_user", "i_bias", "W_item", "g_bias"):
        np.testing.assert_array_equal(trs[0].view(name).view(np.uint32), trs[1].view(name).view(np.uint32))


@pytest.mark.skipif(not oracle.have_reference(), reason="compiled reference (oracle/_ref) not present")
def test_partial_rand_init_matches_reference_on_initialised_rows():
    """num_randinit_{u,i}factor (apex_svd_model.h:671-675,688-692): only the first N rows are
    drawn.  The reference leaves the other rows as uninitialised memalign memory
    (apex_tensor_sse.h:26-38) although its header promises 0 (apex_svd_model.h:395-401); the
    oracle (and the HIP engine) zero them, so only the drawn rows are comparable."""
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=40, num_item=30, num_factor=12,
                           num_randinit_ufactor=25, num_randinit_ifactor=11)
    views = []
    for kind in ("port", "reference"):
        t = oracle.OracleTrainer(kind, 0, 0)
        t.seed(10)
        for k, v in conf:
            t.set_param(k, v)
        t.init_model()
        views.append((t.view("W_user"), t.view("W_item")))
    np.testing.assert_array_equal(views[0][0][:25].view(np.uint32), views[1][0][:25].view(np.uint32))
    np.testing.assert_array_equal(views[0][1][:11].view(np.uint32), views[1][1][:11].view(np.uint32))
    assert not views[0][0][25:].any() and not views[0][1][11:].any()


def test_model_roundtrip_and_warm_start(tmp_path):
    """save_model -> load_model -> identical bytes; training resumes identically (task=1 path,
    svd_feature.cpp:175-182)."""
    base, _ = cases.ml100k()
    part = base.slice_rows(0, 5000)
    conf = cases.conf_with(cases.BASICMF_CONF, num_factor=8)

    def fresh():
        t = port(0, 0)
        t.seed(10)
        for k, v in conf:
            t.set_param(k, v)
        return t
    a = fresh()
    a.init_model()
    a.init_trainer()
    a.update_batch(part)
    p = str(tmp_path / "m.model")
    a.save_model(p)
    b = fresh()
    b.load_model(p)
    b.init_trainer()
    p2 = str(tmp_path / "m2.model")
    b.save_model(p2)
    assert open(p, "rb").read() == open(p2, "rb").read()
    a.update_batch(part)
    b.update_batch(part)
    np.testing.assert_array_equal(a.view("W_item").view(np.uint32), b.view("W_item").view(np.uint32))


def test_reference_tree_fixtures():
    """demo/basicMF/ua.base.buffer is reproduced byte for byte from ua.base.example (pins the
    CSR buffer format), and demo/basicMF/eg.pred.txt is met within the loose +-5e-3 the
    platform-dependent rand() allows."""
    for stem in ("ua.base", "ua.test"):
        d = D.read_text_features(os.path.join(FIX, stem + ".example"))
        out = os.path.join("/tmp", "svdf_%s_%d.buffer" % (stem, os.getpid()))
        D.write_csr_buffer(out, d)
        assert open(out, "rb").read() == open(os.path.join(FIX, stem + ".buffer"), "rb").read()
        back = D.read_csr_buffer(out)
        os.unlink(out)
        np.testing.assert_array_equal(back.feat_index, d.feat_index)
        np.testing.assert_array_equal(back.row_ptr, d.row_ptr)
    res = scenarios.run_scenario("basicmf_example", port)
    eg = np.array(open(os.path.join(FIX, "eg.pred.txt")).read().split(), np.float32)
    np.testing.assert_allclose(res["pred"], eg, atol=5e-3)


def test_baseline_config1_rmse():
    """BASELINE config 1: demo/basicMF on ML-100K, k=16, one epoch: test RMSE 1.265035 -> 1.047025."""
    res = scenarios.run_scenario("basicmf_ml100k_k16", port)
    assert abs(res["rmse"] - 1.047025) < 5e-7


REFDIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref")


@pytest.mark.skipif(not os.path.exists(os.path.join(REFDIR, "make_ugroup_buffer")), reason="reference tools (oracle/_ref) not present")
def test_buffer_writers_match_the_reference_tools_byte_for_byte(tmp_path):
    """svdfeature_amd.data writes the binary buffer formats the reference's own tools write
    (tools/make_feature_buffer.cpp, tools/make_ugroup_buffer.cpp -> apex_svd_data.cpp:118-195, 558-595):
    same text input -> identical bytes, including split users (block_max_line) and empty feedback lists."""
    import subprocess
    rng = np.random.default_rng(3)
    # --- random-order CSR buffer with ragged rows
    d = cases.sparse_feature_rows(2500, 50, 40, 7, 21)
    txt = tmp_path / "feat.txt"
    with open(txt, "w") as f:
        for r in range(d.num_row):
            label, ng, nu, ni, idx, val = d.row(r)
            f.write("%g %d %d %d %s\n" % (label, ng, nu, ni, " ".join("%d:%.9g" % (a, b) for a, b in zip(idx, val))))
    subprocess.check_call([os.path.join(REFDIR, "make_feature_buffer"), str(txt), str(tmp_path / "ref.buffer"), "-batch_size", "700"],
                          stdout=subprocess.DEVNULL)
    D.write_csr_buffer(str(tmp_path / "mine.buffer"), D.read_text_features(str(txt)), batch_size=700)
    assert open(tmp_path / "ref.buffer", "rb").read() == open(tmp_path / "mine.buffer", "rb").read()
    back = D.read_csr_buffer(str(tmp_path / "ref.buffer"))
    np.testing.assert_array_equal(back.feat_index, d.feat_index)
    np.testing.assert_array_equal(back.feat_value.view(np.uint32), d.feat_value.view(np.uint32))
    # --- user-group buffer: users with 1..25 rows, block_max_line 10 splits the long ones
    rows, fb = [], []
    for uid in range(60):
        nrow = int(rng.integers(1, 26))
        nfb = 0 if uid % 6 == 0 else int(rng.integers(1, 5))
        fb.append((nrow, sorted(rng.choice(40, nfb, replace=False).tolist())))
        for _ in range(nrow):
            rows.append((int(rng.integers(1, 6)), uid, int(rng.integers(0, 40))))
    gtxt, ftxt = tmp_path / "group.txt", tmp_path / "fb.txt"
    with open(gtxt, "w") as f:
        for lab, uid, iid in rows:
            f.write("%d 0 1 1 %d:1 %d:1\n" % (lab, uid, iid))
    with open(ftxt, "w") as f:
        for nrow, ids in fb:
            f.write("%d %d %s\n" % (nrow, len(ids), " ".join("%d:0.5" % x for x in ids)))
    subprocess.check_call([os.path.join(REFDIR, "make_ugroup_buffer"), str(gtxt), str(tmp_path / "ref.ug"), "-fd", str(ftxt), "-max_block", "10"],
                          stdout=subprocess.DEVNULL)
    blocks = D.make_user_blocks(D.read_text_features(str(gtxt), sort_sections=True), D.read_feedback_file(str(ftxt)), block_max_line=10)
    D.write_ugroup_buffer(str(tmp_path / "mine.ug"), blocks)
    assert open(tmp_path / "ref.ug", "rb").read() == open(tmp_path / "mine.ug", "rb").read()
    back = D.read_ugroup_buffer(str(tmp_path / "ref.ug"))
    assert [b.extend_tag for b in back] == [b.extend_tag for b in blocks]
    assert sum(b.data.num_row for b in back) == len(rows)
