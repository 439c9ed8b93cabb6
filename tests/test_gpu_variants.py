"""GPU parity of the variant solvers: the HIP engine with extend_type 2 (multi-level implicit feedback,
k_imfb), 15 (bilinear) and 1 (SVD++) against tests/golden/variants.npz (compiled default factory of the reference) and
against the C oracle on staged / chunked / resident / interleaved-predict paths -- byte-identical model files."""
import numpy as np
import pytest

import cases
import scenarios
import svdfeature_amd as sa
from oracle import oracle
from test_variants import check_against_golden

pytestmark = pytest.mark.gpu

VIEWS = ("W_user", "W_item", "W_ufeedback", "ufeedback_bias", "u_bias", "i_bias")


def hip(f, a, e=0):
    return sa.Trainer(f, a, e)


def port(f, a, e=0):
    return oracle.OracleTrainer("port", f, a, e)


def _ready(mk, conf, ext, seed=10):
    t = mk(1, 0, ext)
    t.seed(seed)
    for k, v in conf:
        t.set_param(k, v)
    t.init_model()
    t.init_trainer()
    return t


@pytest.mark.parametrize("name", list(scenarios.VARIANT_SCENARIOS))
def test_variant_scenarios_match_reference_golden(name):
    check_against_golden(name, hip)


@pytest.mark.parametrize("k", [7, 16, 64, 100, 300])
@pytest.mark.parametrize("window", [1 << 22, 9])
def test_multi_level_feedback_every_path_matches_the_oracle(k, window):
    """Nested spans through per-block calls (staging windows of 9 rows cut units in the middle of open levels: the stack then
    travels through the device state slot), through a resident dataset, with predictions between the rounds (predict pushes
    and pops levels too) -- parameters and predictions equal to the oracle's, bit for bit; all lane-group widths incl. wide rows."""
    nu, ni = 90, 60
    blocks = cases.nested_blocks(60, nu, ni, ni, seed=k)
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=k, num_ufeedback=ni, wd_ufeedback=0.004,
                           ufeedback_init_sigma=0.02, wd_ufeedback_bias=0.001, scale_lr_ufeedback=0.8, learning_rate=0.02)
    o, t, r = _ready(port, conf, 2), _ready(hip, conf, 2), _ready(hip, conf, 2)
    t.set_knob("stage_window", window)
    ds = r.dataset_from_blocks(blocks)
    assert ds.kind == 4 and ds.num_row == sum(b.data.num_row for b in blocks)
    for rnd in range(2):
        for b in blocks:
            o.update_block(b)
            t.update_block(b)
        r.train_dataset(ds)
        po = np.concatenate([o.predict_block(b) for b in blocks])
        pt = np.concatenate([t.predict_block(b) for b in blocks])
        np.testing.assert_array_equal(po.view(np.uint32), pt.view(np.uint32))
        np.testing.assert_array_equal(po.view(np.uint32), r.predict_dataset(ds).view(np.uint32))
    for name in VIEWS:
        np.testing.assert_array_equal(t.view(name).view(np.uint32), o.view(name).view(np.uint32))
        np.testing.assert_array_equal(r.view(name).view(np.uint32), o.view(name).view(np.uint32))


def test_nesting_deeper_than_four_levels_takes_the_deep_kernel_build():
    """spans nested up to 11 deep (the usual build of k_imfb holds 4 levels in registers; deeper data switches the engine to the
    16-level build): parameters and predictions byte-identical to the oracle, per-block calls with a small staging window that cuts
    units with open levels, and the same pass as a resident data set"""
    nu, ni = 60, 40
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=12, num_ufeedback=ni, wd_ufeedback=0.004, ufeedback_init_sigma=0.01)
    rng = np.random.default_rng(31)
    e0 = np.zeros(0, np.uint32), np.zeros(0, np.float32)

    def fb():
        n = int(rng.integers(0, 5))
        return np.sort(rng.choice(ni, size=n, replace=False)).astype(np.uint32), np.full(n, 1.0 / np.sqrt(max(n, 1)), np.float32)

    def rows(uid):
        rs = [(float(rng.integers(1, 6)), [], [(int(uid), 1.0)], [(int(rng.integers(0, ni)), 1.0)]) for _ in range(int(rng.integers(0, 4)))]
        return sa.CSRData.from_rows(rs) if rs else sa.CSRData.empty()
    blocks, peak = [], 0
    for span in range(12):
        uid, d = int(rng.integers(0, nu)), int(rng.integers(3, 12))
        peak = max(peak, d + 1)
        lists = [fb() for _ in range(d)]
        for l in range(d):                                           # d nested STARTs ...
            blocks.append(sa.PlusBlock(lists[l][0], lists[l][1], rows(uid), 1))
            if rng.integers(0, 2):
                blocks.append(sa.PlusBlock(e0[0], e0[1], rows(uid), 3))
        f = fb()
        blocks.append(sa.PlusBlock(f[0], f[1], rows(uid), 0))       # ... a DEFAULT block on top (push + pop at once) ...
        for l in reversed(range(d)):                                 # ... and the ENDs, each scattering through its own list
            blocks.append(sa.PlusBlock(lists[l][0], lists[l][1], rows(uid), 2))
    assert peak == 12
    t, r, o = _ready(hip, conf, 2), _ready(hip, conf, 2), _ready(port, conf, 2)
    t.set_knob("stage_window", 7)
    ds = r.dataset_from_blocks(blocks)
    for _ in range(2):
        for b in blocks:
            t.update_block(b)
            o.update_block(b)
        t.finish_round()
        r.train_dataset(ds)
    for name in VIEWS:
        np.testing.assert_array_equal(t.view(name).view(np.uint32), o.view(name).view(np.uint32))
        np.testing.assert_array_equal(r.view(name).view(np.uint32), o.view(name).view(np.uint32))
    po = np.concatenate([o.predict_block(b) for b in blocks])
    pt = np.concatenate([t.predict_block(b) for b in blocks])
    np.testing.assert_array_equal(po.view(np.uint32), pt.view(np.uint32))


def test_multi_level_feedback_errors_and_limits():
    nu, ni = 30, 20
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=8, num_ufeedback=ni)
    t = _ready(hip, conf, 2)
    e = sa.PlusBlock(np.zeros(0, np.uint32), np.zeros(0, np.float32), sa.CSRData.empty(), 2)
    with pytest.raises(sa.SvdfError, match="start tag,end tag error in implicit feedback"):
        t.update_block(e)   # END without START (apex_multi_imfb.h:183)
    s = sa.PlusBlock(np.zeros(0, np.uint32), np.zeros(0, np.float32), sa.CSRData.empty(), 1)
    for _ in range(16):
        t.update_block(s)
    with pytest.raises(sa.SvdfError, match="more than 16 nested"):   # the reference's stack is an unbounded std::vector (apex_multi_imfb.h:41-58)
        t.update_block(s)
    with pytest.raises(sa.SvdfError, match="extend_type 30 is not supported"):
        sa.Trainer(0, 0, 30)
    with pytest.raises(sa.SvdfError, match="need the user-group format"):
        _x = sa.Trainer(0, 0, 2)
        for k, v in conf:
            _x.set_param(k, v)
        _x.init_model()
        _x.init_trainer()


def test_bilinear_trains_like_svdpp_and_keeps_its_file_tail(tmp_path):
    res = scenarios.run_scenario("bilinear_is_svdpp_plus_file_tail", hip)
    base = scenarios.run_scenario("svdpp_random", lambda f, a: sa.Trainer(f, a))
    assert res["model"][4:len(base["model"])] == base["model"][4:]
    p = str(tmp_path / "m.model")
    open(p, "wb").write(res["model"])
    t = hip(1, 0, 15)
    t.load_model(p)
    t.init_trainer()
    p2 = str(tmp_path / "m2.model")
    t.save_model(p2)
    assert open(p2, "rb").read() == res["model"]
