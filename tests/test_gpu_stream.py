"""The conflict DAG inside ONE launch per pass (svdf_k_stream.hip / svdf_stream.cpp; knob stream_exec): tiles of the level-sorted
arrays handed out in order to a persistent grid, every instance waiting for the tiles that hold the previous touchers of its rows.  Same arithmetic,
same partial order as the level-by-level pass: the model must equal it -- and the oracle -- bit for bit, whatever the number of persistent waves."""
import numpy as np
import pytest

import cases
import svdfeature_amd as sa

pytestmark = pytest.mark.gpu
NAMES = ("W_user", "W_item", "u_bias", "i_bias")


def _run(u, i, r, nu, ni, stream, waves=0, passes=2, k=64):
    t = sa.Trainer(0, 0)
    t.seed(10)
    for kk, v in cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=k):
        t.set_param(kk, v)
    t.init_model()
    t.init_trainer()
    t.set_knob("stream_exec", stream)
    t.set_knob("pivot_exec", 0)   # (hot rows would otherwise be walked as units: svdf_pivot.cpp, another kind of data set)
    t.set_knob("runs_exec", 0)    # (... and large data sets of this configuration scheduled as runs: svdf_runs.cpp)
    t.set_knob("stream_spin_limit", 400000)
    if waves:
        t.set_knob("stream_waves", waves)
    ds = t.dataset_from_triples(u, i, r)
    for _ in range(passes):
        t.train_dataset(ds)
    t.synchronize()
    return {n: t.view(n).copy() for n in NAMES}, t.counter(21), ds.num_batches


@pytest.mark.parametrize("nu,ni,n,zipf,waves", [(20000, 2000, 200000, False, 0), (3000, 300, 60000, True, 0), (50000, 4000, 400000, False, 7),
                                                (500, 40, 20000, True, 1), (200000, 20000, 1500000, False, 0), (64, 64, 4000, False, 2048)])
def test_stream_pass_equals_the_level_by_level_pass(nu, ni, n, zipf, waves):
    u, i, r = cases.planted_triples(n, nu, ni, seed=nu + n, zipf=zipf)
    a, sp0, levels = _run(u, i, r, nu, ni, 0)
    b, sp1, _ = _run(u, i, r, nu, ni, 1, waves)
    assert sp0 == 0 and sp1 == 2, (sp0, sp1, levels)
    for name in NAMES:
        assert np.array_equal(a[name].view(np.uint32), b[name].view(np.uint32)), (name, levels)


def test_stream_pass_equals_the_oracle():
    from oracle import oracle
    oracle.build()
    nu, ni, n = 5000, 700, 120000
    u, i, r = cases.planted_triples(n, nu, ni, seed=5, zipf=True)
    got, sp, _ = _run(u, i, r, nu, ni, 1, passes=1)
    assert sp == 1
    o = oracle.OracleTrainer("port", 0, 0)
    o.seed(10)
    for kk, v in cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=64):
        o.set_param(kk, v)
    o.init_model()
    o.init_trainer()
    o.update_batch(sa.CSRData.from_triples(u, i, r))
    for name in NAMES:
        assert np.array_equal(got[name].view(np.uint32), o.view(name).view(np.uint32)), name


def test_configurations_without_a_stream_kernel_keep_the_level_loop():
    nu, ni, n = 2000, 300, 30000
    u, i, r = cases.planted_triples(n, nu, ni, seed=1)
    _, sp, _ = _run(u, i, r, nu, ni, 1, k=32)
    assert sp == 0
