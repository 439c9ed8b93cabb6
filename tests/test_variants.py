"""Variant solvers of the reference's default factory (apex_svd.cpp:32-44) -- extend_type 2 = multi-level
implicit feedback (solvers/multi-imfb/apex_multi_imfb.h), 15 = bilinear (solvers/bilinear/apex_svd_bilinear.h), 1 = SVD++:
pins the C oracle's restatement to tests/golden/variants.npz (written from the compiled default factory,
oracle/_ref/libsvdf_ref_full.so) and, where that library is present, to the library itself."""
import os

import numpy as np
import pytest

import cases
import scenarios
from oracle import oracle

GOLD = np.load(os.path.join(cases.GOLDEN, "variants.npz"))


def port(f, a, e=0):
    return oracle.OracleTrainer("port", f, a, e)


def ref_full(f, a, e=0):
    return oracle.OracleTrainer("reference_full", f, a, e)


def check_against_golden(name, mk):
    res = scenarios.run_scenario(name, mk)
    dg = scenarios.digest(res)
    assert dg["model0_md5"] == str(GOLD[name + "/model0_md5"]), "initial model differs"
    assert dg["model_len"] == int(GOLD[name + "/model_len"])
    np.testing.assert_array_equal(dg["model_sample"].view(np.uint32), GOLD[name + "/model_sample"].view(np.uint32))
    assert dg["model_md5"] == str(GOLD[name + "/model_md5"]), "model file is not byte-identical to the reference's"
    assert dg["pred_md5"] == str(GOLD[name + "/pred_md5"])
    assert dg["rmse"] == float(GOLD[name + "/rmse"])
    return res


@pytest.mark.parametrize("name", list(scenarios.VARIANT_SCENARIOS))
def test_oracle_variants_match_golden(name):
    check_against_golden(name, port)


@pytest.mark.skipif(not oracle.have_reference_full(), reason="compiled default factory (oracle/_ref/libsvdf_ref_full.so) not present")
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_oracle_multi_level_feedback_matches_live_reference(seed):
    """Nested START..END spans three deep, predictions interleaved with training (the level stack persists between calls),
    a disabled level, no user bias: models and predictions identical to the reference's own SVDPPMultiIMFB."""
    nu, ni = 60, 40
    blocks = cases.nested_blocks(80, nu, ni, ni, seed=seed)
    for nob, dis in ((0, None), (1, None), (0, 1), (0, 0)):
        conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=10, num_ufeedback=ni, wd_ufeedback=0.004,
                               ufeedback_init_sigma=0.02, wd_ufeedback_bias=0.001, scale_lr_ufeedback=0.8, learning_rate=0.02, no_user_bias=nob)
        if dis is not None:
            conf = conf + [("ufeedback_disable_level", str(dis))]
        outs = []
        for mk in (ref_full, port):
            t = mk(1, 0, 2)
            t.seed(10)
            for k, v in conf:
                t.set_param(k, v)
            t.init_model()
            t.init_trainer()
            preds = []
            for _ in range(2):
                for b in blocks:
                    t.update_block(b)
                preds += [t.predict_block(b) for b in blocks]
            outs.append((np.concatenate(preds), [t.view(v) for v in ("W_user", "W_item", "W_ufeedback", "ufeedback_bias", "u_bias", "i_bias")]))
        np.testing.assert_array_equal(outs[0][0].view(np.uint32), outs[1][0].view(np.uint32))
        for a, b in zip(outs[0][1], outs[1][1]):
            np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32))


def test_bilinear_model_file_round_trip(tmp_path):
    """extend_type 15: the model file is the SVD++ model + BParam (34 ints) + W_bi (num_item x num_bi_feedback zeros); it
    loads back, trains on and saves the same tail (apex_svd_bilinear.h:60-68,194-205)."""
    res = scenarios.run_scenario("bilinear_is_svdpp_plus_file_tail", port)
    base = scenarios.run_scenario("svdpp_random", port)
    m, b = res["model"], base["model"]
    assert m[4:len(b)] == b[4:] and m[2] == 15
    tail = m[len(b):]
    assert len(tail) == 136 + 8 + 40 * 6 * 4
    assert list(np.frombuffer(tail[:8], np.int32)) == [6, 2] and list(np.frombuffer(tail[136:144], np.int32)) == [6, 40]
    assert not np.frombuffer(tail[144:], np.float32).any()
    p = str(tmp_path / "m.model")
    open(p, "wb").write(m)
    t = port(1, 0, 15)
    t.load_model(p)
    t.init_trainer()
    p2 = str(tmp_path / "m2.model")
    t.save_model(p2)
    assert open(p2, "rb").read() == m
