"""Named end-to-end scenarios of the hot path, runnable on any ISVDTrainer-shaped engine.

``run_scenario(name, make_trainer)`` drives an engine through the reference's call protocol
(svd_feature.cpp:194-283: create -> set_param* -> init_model -> init_trainer -> per round
set_round / update* / finish_round) and returns the model file bytes plus predictions.
The same scenarios are run on: the compiled reference (golden generation), the C oracle
(CPU tests) and the HIP engine (GPU tests).
"""
import hashlib
import os
import tempfile

import numpy as np

import cases
from svdfeature_amd.data import CSRData

SEED = 10  # svd_feature.cpp:293


def _conf(**kw):
    return cases.conf_with(cases.BASICMF_CONF, **kw)


def _scn_basicmf_ml100k_k16(tmp):
    base, test = cases.ml100k()
    return dict(conf=_conf(num_factor=16), format_type=0, active_type=0, rounds=1, train=base, test=test)


def _scn_basicmf_ml100k_k64(tmp):
    base, test = cases.ml100k()
    return dict(conf=_conf(num_factor=64), format_type=0, active_type=0, rounds=2, train=base, test=test)


def _scn_basicmf_example(tmp):  # demo/basicMF/run.sh
    rows = [(5, [], [(1, 1)], [(282, 1)]), (3, [], [(2, 1)], [(270, 1)]), (4, [], [(4, 1)], [(221, 1)]), (1, [], [(5, 1)], [(258, 1)])]
    d = CSRData.from_rows(rows)
    return dict(conf=_conf(), format_type=0, active_type=0, rounds=40, train=d, test=d)


def _scn_neighborhood_example(tmp):  # demo/neighborhoodModel
    rows = [(4, [(1, -1)], [(2, 1)], [(1, 1)]), (2, [(2, -2), (3, 2)], [(2, 1)], [(2, 1)]),
            (1, [(4, 2)], [(2, 1)], [(3, 1)]), (5, [(5, 1)], [(2, 1)], [(4, 1)])]
    d = CSRData.from_rows(rows)
    return dict(conf=_conf(num_global=6, wd_global=0.001), format_type=0, active_type=0, rounds=40, train=d, test=d)


def _scn_binary_example(tmp):  # demo/binaryClassification (base_score stays at its 0.5 default)
    rows = [(0, [], [(1, 1)], [(282, 1)]), (1, [], [(2, 1)], [(270, 1)]), (0, [], [(4, 1)], [(221, 1)]), (1, [], [(5, 1)], [(258, 1)])]
    d = CSRData.from_rows(rows)
    conf = [(k, v) for k, v in _conf() if k != "base_score"]
    return dict(conf=conf, format_type=0, active_type=2, rounds=40, train=d, test=d)


def _scn_implicit_example(tmp):  # demo/implicitFeedback
    from svdfeature_amd.data import make_user_blocks
    rows = [(5, 2, 170), (3, 2, 523), (1, 3, 123), (2, 3, 12), (4, 3, 64), (5, 3, 69), (3, 1, 89), (1, 1, 103), (1, 1, 532)]
    d = CSRData.from_rows([(r, [], [(u, 1)], [(i, 1)]) for r, u, i in rows])
    fb = [(2, np.array([170, 523], np.uint32), np.array([0.5, 0.5], np.float32)),
          (4, np.array([64, 69], np.uint32), np.array([0.5, 0.5], np.float32)),
          (3, np.array([89], np.uint32), np.array([1.0], np.float32))]
    blocks = make_user_blocks(d, fb)
    return dict(conf=_conf(num_ufeedback=1682, wd_ufeedback=0.004), format_type=1, active_type=0, rounds=40,
                train_blocks=blocks, test_blocks=blocks)


def _sparse(tmp, seed, active_type=0, binary=False, side=False, n=600, k=12, extra=(), nu=40, ni=30, **conf_kw):
    ng = 9
    train = cases.sparse_feature_rows(n, nu, ni, ng, seed, binary_label=binary)
    test = cases.sparse_feature_rows(80, nu, ni, ng, seed + 1, binary_label=binary)
    kw = dict(num_user=nu, num_item=ni, num_global=ng, num_factor=k, wd_global=0.002,
              wd_user_bias=0.001, wd_item_bias=0.003, learning_rate=0.01)
    if binary:
        kw["base_score"] = 0.4
    kw.update(conf_kw)
    if side:
        fu, fi = os.path.join(tmp, "feat_user.txt"), os.path.join(tmp, "feat_item.txt")
        cases.write_side_table(fu, nu - 5, nu, seed + 2)
        cases.write_side_table(fi, ni, ni, seed + 3)
        kw.update(feature_user=fu, feature_item=fi)
    return dict(conf=_conf(**kw) + list(extra), format_type=0, active_type=active_type, rounds=3, train=train, test=test)


def _scn_common_latent_triples(tmp):
    """(u, i, r) triples over ONE id space (common_latent_space, apex_svd_model.h:516-536: W_item IS W_user), including
    instances whose user id equals their item id -- the same row is updated twice, through memory, in the reference."""
    n, ids = 900, 35
    rng = np.random.default_rng(123)
    u = rng.integers(0, ids, n).astype(np.uint32)
    i = rng.integers(0, ids, n).astype(np.uint32)
    i[::7] = u[::7]
    r = rng.integers(1, 6, n).astype(np.float32)
    train = CSRData.from_triples(u, i, r)
    test = CSRData.from_triples(u[:90], i[::-1][:90].copy(), r[:90])
    conf = _conf(num_user=ids, num_item=ids, num_factor=8, common_latent_space=1, common_feedback_space=1, learning_rate=0.01)
    return dict(conf=conf, format_type=0, active_type=0, rounds=3, train=train, test=test)


def _scn_svdpp_random(tmp, nu=50, ni=40, **kw):
    blocks = cases.user_blocks(45, nu, ni, ni, 77, split_every=4)
    test = cases.user_blocks(20, nu, ni, ni, 78)
    conf = _conf(num_user=nu, num_item=ni, num_factor=16, num_ufeedback=ni, wd_ufeedback=0.004,
                 wd_ufeedback_bias=0.002, scale_lr_ufeedback=0.7, ufeedback_init_sigma=0.01, learning_rate=0.01, **kw)
    return dict(conf=conf, format_type=1, active_type=0, rounds=3, train_blocks=blocks, test_blocks=test)


def _scn_svdpp_rich_rows(tmp, side):
    """User blocks whose rows carry global features, a second user id and several item ids (cases.rank_blocks, used as
    plain training blocks), with implicit feedback and -- optionally -- both side tables: the generic user-unit path."""
    nu, ni, ng = 60, 50, 8
    blocks = cases.rank_blocks(70, nu, ni, ng, 5, graded=True)
    test = cases.rank_blocks(25, nu, ni, ng, 6, graded=True)
    kw = dict(num_user=nu, num_item=ni, num_global=ng, num_factor=12, num_ufeedback=ni, wd_ufeedback=0.004, wd_ufeedback_bias=0.001,
              wd_global=0.002, ufeedback_init_sigma=0.01, learning_rate=0.01, wd_user_bias=0.001)
    if side:
        fu, fi = os.path.join(tmp, "feat_user.txt"), os.path.join(tmp, "feat_item.txt")
        cases.write_side_table(fu, nu - 7, nu, 31)
        cases.write_side_table(fi, ni, ni, 32)
        kw.update(feature_user=fu, feature_item=fi)
    return dict(conf=_conf(**kw), format_type=1, active_type=0, rounds=3, train_blocks=blocks, test_blocks=test)


SCENARIOS = {
    "basicmf_ml100k_k16": _scn_basicmf_ml100k_k16,
    "basicmf_ml100k_k64": _scn_basicmf_ml100k_k64,
    "basicmf_example": _scn_basicmf_example,
    "neighborhood_example": _scn_neighborhood_example,
    "binary_example": _scn_binary_example,
    "implicit_example": _scn_implicit_example,
    "sparse_linear": lambda t: _sparse(t, 101),
    "sparse_k10_tail": lambda t: _sparse(t, 102, k=10),
    "sparse_side_tables": lambda t: _sparse(t, 103, side=True),
    "sparse_sigmoid_l2": lambda t: _sparse(t, 104, active_type=1, binary=True),
    "sparse_logistic": lambda t: _sparse(t, 105, active_type=2, binary=True),
    "sparse_rank": lambda t: _sparse(t, 106, active_type=3, binary=True, no_user_bias=1),
    "sparse_hinge_smooth": lambda t: _sparse(t, 107, active_type=5, binary=True, base_score=0.5),
    "sparse_hinge_l2": lambda t: _sparse(t, 108, active_type=6, binary=True, base_score=0.5),
    "sparse_qsgrad": lambda t: _sparse(t, 109, active_type=7, binary=True),
    "sparse_reg_l1": lambda t: _sparse(t, 110, reg_method=1, reg_global=1, wd_user=0.02, wd_item=0.03),
    "sparse_reg_project": lambda t: _sparse(t, 111, reg_method=2, wd_user=0.0008, wd_item=0.0009, ui_init_sigma=0.02),
    "sparse_reg_mixed3": lambda t: _sparse(t, 112, reg_method=3, wd_user=0.02),
    "sparse_nonneg_decaylr": lambda t: _sparse(t, 113, user_nonnegative=1, decay_learning_rate=1, decay_rate=0.9),
    # per-range decay: "wd" must precede its "bound" and the last bound must cover every id
    # (apex_svd_base.h:57-74)
    "sparse_regfree_ranges": lambda t: _sparse(t, 114, num_regfree_global=3, extra=[
        ("up:wd", "0.01"), ("up:bound", "20"), ("up:wd", "0.001"), ("up:bound", "40"),
        ("ip:wd", "0.02"), ("ip:bound", "7"), ("uip:wd", "0.003"), ("ip:bound", "30"),
        ("gp:wd", "0.05"), ("gp:bound", "9")]),
    "sparse_wd_tiny_skipmul": lambda t: _sparse(t, 116, wd_user=0.00005, wd_item=0.00002),
    # lazy decay: the reference forms (float)(ref - sample_counter) from UNSIGNED counters (apex_svd_base.h:195,226,266),
    # i.e. ~4.29e9 for every id seen before -- rows are wiped before each reuse unless lambda rounds 1-lambda to 1.
    # Restated as is; these pin that behaviour.
    "sparse_lazy_l2": lambda t: _sparse(t, 117, reg_method=4, reg_global=4),
    "sparse_lazy_l1": lambda t: _sparse(t, 118, reg_method=5, reg_global=5, num_regfree_global=2),
    "sparse_lazy_rows_only_side": lambda t: _sparse(t, 119, side=True, reg_method=4, reg_global=1, wd_user=0.0),
    "sparse_lazy_globals_only": lambda t: _sparse(t, 120, reg_method=1, reg_global=4, wd_user=0.02, wd_item=0.03),
    "svdpp_random": _scn_svdpp_random,
    "svdpp_random_lazy": lambda t: _scn_svdpp_random(t, reg_method=5),
    "svdpp_random_nobias": lambda t: _scn_svdpp_random(t, no_user_bias=1),
    # shared parameter spaces (apex_svd_model.h:511-556): users and items in one matrix, feedback rows = user rows
    "svdpp_rich_rows": lambda t: _scn_svdpp_rich_rows(t, False),
    "svdpp_rich_rows_side_tables": lambda t: _scn_svdpp_rich_rows(t, True),
    # alias keys of the parameter parsers (apex_svd_model.h:350-368, 456-476) and the init-time options they reach
    "sparse_alias_keys": lambda t: _sparse(t, 122, nu=30, ni=30, user_nonnegative=1, decay_learning_rate=1, decay_rate=0.8, extra=[
        ("num_uiset", "30"), ("wd_uiset", "0.003"), ("wd_uiset_bias", "0.002"), ("ui_init_sigma", "0.02"), ("u_init_sigma", "0.015"),
        ("item_nonnegative", "1"), ("min_learning_rate", "0.1")]),   # (num_randinit_* leaves uninitialised rows in the reference: test_oracle.py)
    "sparse_common_latent": lambda t: _sparse(t, 121, nu=30, ni=30, common_latent_space=1, common_feedback_space=1),
    "common_latent_triples": _scn_common_latent_triples,
    "svdpp_common_feedback": lambda t: _scn_svdpp_random(t, common_feedback_space=1),
    "svdpp_common_latent": lambda t: _scn_svdpp_random(t, nu=50, ni=50, common_latent_space=1, common_feedback_space=1),
}


# ---- variant solvers of the reference's default factory (apex_svd.cpp:32-44): make_trainer gets a third
# argument, the extend_type.  Goldens: tests/golden/variants.npz, written from oracle/_ref/libsvdf_ref_full.so.
def _scn_imfb(tmp, seed=31, nested=True, **kw):
    nu, ni = 70, 45
    if nested:
        blocks = cases.nested_blocks(70, nu, ni, ni, seed)
        test = cases.nested_blocks(25, nu, ni, ni, seed + 1)
    else:   # the shapes the reference's own loader writes: DEFAULT blocks and START / MIDDLE / END splits of one user
        blocks = cases.user_blocks(45, nu, ni, ni, seed, split_every=4)
        test = cases.user_blocks(20, nu, ni, ni, seed + 1)
    extra = kw.pop("extra", [])
    conf = _conf(num_user=nu, num_item=ni, num_factor=kw.pop("num_factor", 12), num_ufeedback=ni, wd_ufeedback=0.004, wd_ufeedback_bias=0.002,
                 scale_lr_ufeedback=0.7, ufeedback_init_sigma=0.02, learning_rate=0.01, **kw) + list(extra)
    return dict(conf=conf, format_type=1, active_type=0, extend_type=2, rounds=3, train_blocks=blocks, test_blocks=test)


def _scn_bilinear(tmp, **kw):
    s = _scn_svdpp_random(tmp, **kw)
    s["conf"] = s["conf"] + [("num_bi_feedback", "6"), ("start_ufeedback", "2"), ("reg_bi_feedback", "2"), ("wd_bi_feedback", "0.01"),
                             ("slr_bi_feedback", "0.5")]
    s["extend_type"] = 15
    return s


VARIANT_SCENARIOS = {
    "imfb_loader_shapes": lambda t: _scn_imfb(t, nested=False),                    # == SVD++ on such data
    "imfb_nested": lambda t: _scn_imfb(t),
    "imfb_nested_nobias_k33": lambda t: _scn_imfb(t, seed=41, no_user_bias=1, num_factor=33),
    "imfb_nested_disable_level1": lambda t: _scn_imfb(t, seed=51, extra=[("ufeedback_disable_level", "1")]),
    "imfb_nested_disable_level0": lambda t: _scn_imfb(t, seed=61, extra=[("ufeedback_disable_level", "0")]),
    "imfb_nested_l1_ranges": lambda t: _scn_imfb(t, seed=71, reg_method=1, wd_user=0.02, wd_item=0.03),
    "imfb_nested_lazy": lambda t: _scn_imfb(t, seed=81, reg_method=5),
    "bilinear_is_svdpp_plus_file_tail": _scn_bilinear,
    "extend1_is_svdpp": lambda t: dict(_scn_svdpp_random(t), extend_type=1),
}
SCENARIOS_ALL = dict(SCENARIOS, **VARIANT_SCENARIOS)


def run_scenario(name, make_trainer, chunk=None):
    """make_trainer(format_type, active_type) -> engine with the OracleTrainer method set.
    chunk: if set, feed training rows through update_batch in chunks of this many rows
    (exercises staging/flush boundaries); None feeds each round as one batch."""
    with tempfile.TemporaryDirectory() as tmp:
        s = SCENARIOS_ALL[name](tmp)
        if "extend_type" in s:
            tr = make_trainer(s["format_type"], s["active_type"], s["extend_type"])
        else:
            tr = make_trainer(s["format_type"], s["active_type"])
        tr.seed(SEED)
        for k, v in s["conf"]:
            tr.set_param(k, v)
        tr.init_model()
        tr.init_trainer()
        path0 = os.path.join(tmp, "0000.model")
        tr.save_model(path0)
        model0 = open(path0, "rb").read()
        for r in range(s["rounds"]):
            tr.set_round(r)
            if "train" in s:
                d = s["train"]
                if chunk:
                    for st in range(0, d.num_row, chunk):
                        tr.update_batch(d.slice_rows(st, st + chunk))
                else:
                    tr.update_batch(d)
            else:
                for b in s["train_blocks"]:
                    tr.update_block(b)
            tr.finish_round()
        if "test" in s:
            pred = tr.predict_batch(s["test"])
            label = s["test"].row_label
        else:
            pred = np.concatenate([tr.predict_block(b) for b in s["test_blocks"]])
            label = np.concatenate([b.data.row_label for b in s["test_blocks"]])
        path1 = os.path.join(tmp, "final.model")
        tr.save_model(path1)
        model1 = open(path1, "rb").read()
        tr.close()
    return dict(model0=model0, model=model1, pred=np.asarray(pred, np.float32).copy(), label=label.copy(),
                rmse=cases.rmse(pred, label))


def digest(res):
    """Small, committable summary of a scenario result."""
    m = np.frombuffer(res["model"][4 + 1056:], dtype=np.float32)
    step = max(1, m.size // 256)
    return dict(model0_md5=hashlib.md5(res["model0"]).hexdigest(), model_md5=hashlib.md5(res["model"]).hexdigest(),
                model_len=len(res["model"]), model_sample=m[::step][:256].copy(), sample_step=step,
                pred=res["pred"][:4096].copy(), pred_md5=hashlib.md5(res["pred"].tobytes()).hexdigest(),
                rmse=np.float64(res["rmse"]))
