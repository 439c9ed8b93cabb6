"""Window-minibatch step for USER-GROUP blocks and for rows with global features on the CPU: the block checker
step (oracle/svdf_oracle.c: svdo_update_block_stale) pinned to the compiled reference's own SVDPPFeature, its defining properties, and
the oracle-backed simulation the HIP kernels (svdf_k_wunit.hip) are compared with in tests/test_gpu_wunit.py.
Reference path: /root/reference/solvers/base-solver/apex_svd_base.h:506-554 (feedback hooks), :568-582 (update(block)), :188-210
(reg_global), :313-353 (bias terms)."""
import numpy as np
import pytest

import cases
from oracle import oracle
from svdfeature_amd import BlockArrays, CSRData
from svdfeature_amd.data import PlusBlock, TAG_DEFAULT

SVDPP_EXTRA = [("wd_ufeedback", "0.004"), ("ufeedback_init_sigma", "0.01")]


def _make(kind, conf, fmt=1, active=0, seed=10):
    t = oracle.OracleTrainer(kind, fmt, active)
    t.seed(seed)
    for k, v in conf:
        t.set_param(k, v)
    t.init_model()
    t.init_trainer()
    return t


def _blocks_with_globals(nblocks, nu, ni, ng, seed, split_every=3):
    """user blocks whose rows also carry global entries and sometimes two item entries"""
    rng = np.random.default_rng(seed)
    base = cases.user_blocks(nblocks, nu, ni, ni, seed=seed, max_rows=5, max_fb=4, split_every=split_every)
    out = []
    for b in base:
        d = b.data
        rows = []
        for r in range(d.num_row):
            p0, p1, p2, p3 = d.row_ptr[3 * r:3 * r + 4]
            gl = [(int(g), float(rng.uniform(0.2, 1.0))) for g in sorted(rng.choice(ng, size=int(rng.integers(0, 3)), replace=False))]
            us = [(int(d.feat_index[j]), float(d.feat_value[j])) for j in range(p1, p2)]
            it = [(int(d.feat_index[j]), float(d.feat_value[j])) for j in range(p2, p3)]
            if rng.random() < 0.3:
                extra = int(rng.integers(0, ni))
                if extra != it[0][0]:
                    it = sorted(it + [(extra, -0.5)])
            rows.append((float(d.row_label[r]), gl, us, it))
        out.append(PlusBlock(b.index_ufeedback, b.value_ufeedback, CSRData.from_rows(rows), b.extend_tag))
    return out


@pytest.mark.skipif(not oracle.have_reference(), reason="compiled reference (oracle/_ref) not present")
@pytest.mark.parametrize("active,extra", [(0, []), (2, [("base_score", "0.5")]), (0, [("reg_method", "1"), ("reg_global", "1")]),
                                          (0, [("no_user_bias", "1")]), (0, [("scale_lr_ufeedback", "0.5"), ("wd_ufeedback_bias", "0.01")])])
def test_block_checker_step_equals_the_compiled_reference(active, extra):
    """svdo_update_block_stale of the C port == the same step driven through the reference's own SVDPPFeature (START block without rows,
    one MIDDLE block per row with the replicated side put back through save_model / load_model in between, END block without rows),
    bit for bit: all five delta arrays, the private user side, and the untouched replicated side.  DEFAULT and START / MIDDLE / END
    blocks, users without feedback, rows with global entries and two item entries."""
    nu, ni, ng = 30, 12, 5
    blocks = _blocks_with_globals(24, nu, ni, ng, seed=3 + active)
    if active == 2:
        for b in blocks:
            b.data.row_label[:] = (b.data.row_label > 3).astype(np.float32)
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=6, num_global=ng, num_ufeedback=ni, wd_global="0.002") + SVDPP_EXTRA + extra
    got = {}
    for kind in ("port", "reference"):
        t = _make(kind, conf, 1, active)
        init = t.view("W_item").copy(), t.view("W_ufeedback").copy()
        delta = None
        for rep in range(2):   # a second window's worth accumulates into the same arrays
            for b in blocks:
                delta = t.update_block_stale(b, delta)
        np.testing.assert_array_equal(t.view("W_item"), init[0])        # the replicated side does not move
        np.testing.assert_array_equal(t.view("W_ufeedback"), init[1])
        got[kind] = delta + (t.view("W_user"), t.view("u_bias"), t.view("i_bias"), t.view("g_bias"), t.view("ufeedback_bias"))
    for a, b in zip(got["port"], got["reference"]):
        np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32))
    p = got["port"]
    assert np.abs(p[0]).max() > 0 and np.abs(p[2]).max() > 0 and np.abs(p[3]).max() > 0
    if not any(k == "no_user_bias" for k, _ in extra):
        assert np.abs(p[4]).max() > 0


def test_one_block_per_window_is_the_sequential_reference_on_disjoint_rows():
    """A window of ONE block whose rows touch distinct items: the stale step's deltas ARE what update(block) changes, and the private side
    is identical -- the step differs from the reference only in WHEN the replicated side moves."""
    nu, ni = 40, 200
    rng = np.random.default_rng(5)
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=8, num_ufeedback=ni) + SVDPP_EXTRA
    a, b = _make("port", conf), _make("port", conf)
    for step in range(60):
        uid = int(rng.integers(0, nu))
        items = rng.choice(ni, size=5, replace=False)
        fb = np.sort(rng.choice(ni, size=4, replace=False)).astype(np.uint32)
        blk = PlusBlock(fb, np.full(4, 0.5, np.float32), CSRData.from_rows([(float(rng.integers(1, 6)), [], [(uid, 1.0)], [(int(x), 1.0)]) for x in items]), TAG_DEFAULT)
        before = {n: b.view(n).copy() for n in ("W_item", "i_bias", "W_ufeedback", "ufeedback_bias")}
        b.update_block(blk)
        dW, db, dg, dF, dfb = a.update_block_stale(blk)
        np.testing.assert_array_equal(dW, b.view("W_item") - before["W_item"])
        np.testing.assert_array_equal(db, b.view("i_bias") - before["i_bias"])
        np.testing.assert_array_equal(dF, b.view("W_ufeedback") - before["W_ufeedback"])
        np.testing.assert_array_equal(dfb, b.view("ufeedback_bias") - before["ufeedback_bias"])
        np.testing.assert_array_equal(a.view("W_user"), b.view("W_user"))
        for n in before:
            a.set_view(n, b.view(n))


@pytest.mark.skipif(not oracle.have_reference(), reason="compiled reference (oracle/_ref) not present")
def test_bf16_contribution_rounding_of_the_checker_equals_the_same_rounding_over_the_reference():
    """`amd:contrib = bf16`: row contributions are rounded to bfloat16 (nearest even) before they are summed; the C port and the step
    driven through the reference's classes apply the same rounding and agree bit for bit; the result differs from the fp32 step and every
    summed delta of ONE contribution is a bfloat16 number."""
    nu, ni, ng = 30, 12, 5
    blocks = _blocks_with_globals(24, nu, ni, ng, seed=9)
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=6, num_global=ng, num_ufeedback=ni, wd_global="0.002") + SVDPP_EXTRA
    got = {}
    for kind in ("port", "reference", "fp32"):
        t = _make("port" if kind == "fp32" else kind, conf, 1, 0)
        if kind != "fp32":
            t.set_stale_rounding(True)
        delta = None
        for b in blocks:
            delta = t.update_block_stale(b, delta)
        got[kind] = delta + (t.view("W_user"),)
    for a, b in zip(got["port"], got["reference"]):
        np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32))
    assert not np.array_equal(got["port"][0], got["fp32"][0])
    np.testing.assert_allclose(got["port"][0], got["fp32"][0], rtol=0, atol=1e-4)
    one = _make("port", conf, 1, 0)
    one.set_stale_rounding(True)
    d1 = one.update_block_stale(blocks[0])
    assert np.all((d1[3].view(np.uint32) & 0xFFFF) == 0) and np.abs(d1[3]).max() > 0   # feedback rows: one contribution each
