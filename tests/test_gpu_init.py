"""SVDModel::rand_init ON THE DEVICE (svdf_k_init.hip): init_model fills W_user / W_item / W_ufeedback in HBM from the libc rand()
stream the reference's sample_normal() loop (apex-tensor/apex_random.h:67-77, apex_svd_model.h:665-705) would have consumed.  Checked here against the
host loop of the same engine (knob device_init = 0: the reference's operations one draw at a time, the path every golden model0 digest was produced
with until round 4) bit for bit, including where libc's generator stands afterwards; every scenario of tests/test_gpu_parity.py additionally compares
the device-initialised model0 with the reference's own digest."""
import ctypes

import numpy as np
import pytest

import svdfeature_amd as sa

pytestmark = pytest.mark.gpu

libc = ctypes.CDLL(None)
libc.rand.restype = ctypes.c_int


def _init(fmt, conf, seed, device_init, margin=None):
    t = sa.Trainer(fmt, 0)
    t.set_knob("device_init", device_init)
    if margin is not None:
        t.set_knob("device_init_margin_log2", margin)
    t.seed(seed)
    for k, v in conf:
        t.set_param(k, str(v))
    t.init_model()
    nxt = [libc.rand() for _ in range(5)]   # where libc's generator stands after init_model
    views = {}
    for name in ("W_user", "W_item", "W_ufeedback", "u_bias", "i_bias", "g_bias"):
        v = t.view(name)
        if v is not None:
            views[name] = v.copy()
    stats = (t.counter(13), t.counter(14))
    t.init_trainer()   # the device model is what the trainer starts from
    after = {n: t.view(n).copy() for n in views}
    for n in views:
        assert np.array_equal(views[n].view(np.uint32), after[n].view(np.uint32)), n
    return views, nxt, stats


def _same(a, b):
    assert a.keys() == b.keys()
    for n in a:
        assert a[n].shape == b[n].shape, n
        assert np.array_equal(a[n].view(np.uint32), b[n].view(np.uint32)), "%s differs: %d of %d words" % (
            n, int((a[n].view(np.uint32) != b[n].view(np.uint32)).sum()), a[n].size)


CASES = [
    # (format_type, conf)
    (0, dict(num_user=943, num_item=1682, num_global=0, num_factor=64)),
    (0, dict(num_user=300, num_item=77, num_global=5, num_factor=7, u_init_sigma="0.05", i_init_sigma="0.2")),
    (0, dict(num_user=1, num_item=1, num_global=0, num_factor=1)),
    (0, dict(num_user=50, num_item=40, num_global=0, num_factor=33, user_nonnegative=1, item_nonnegative=1)),
    (0, dict(num_user=500, num_item=300, num_global=0, num_factor=16, num_randinit_ufactor=100, num_randinit_ifactor=7)),
    (0, dict(num_user=200, num_item=200, num_global=3, num_factor=100, ui_init_sigma="0")),
    (1, dict(num_user=400, num_item=250, num_global=0, num_factor=128, num_ufeedback=250)),   # sigma 0 on W_ufeedback: the draws are consumed, +-0 stored
    (1, dict(num_user=400, num_item=250, num_global=2, num_factor=20, num_ufeedback=300, ufeedback_init_sigma="0.003")),
    (0, dict(num_user=0, num_item=10, num_global=0, num_factor=8)),
]


@pytest.mark.parametrize("case", range(len(CASES)))
def test_device_init_equals_the_host_loop(case):
    fmt, kw = CASES[case]
    conf = list(kw.items())
    for seed in (10, 12345):
        host, nxt_h, st_h = _init(fmt, conf, seed, 0)
        dev, nxt_d, st_d = _init(fmt, conf, seed, 1)
        _same(host, dev)
        assert nxt_h == nxt_d, "libc's generator stands elsewhere after the device init"
        assert st_h[1] == 0
        n = sum(v.size for k, v in host.items() if k.startswith("W_"))
        if "num_randinit_ufactor" in kw:
            n = (kw["num_randinit_ufactor"] + kw["num_randinit_ifactor"]) * kw["num_factor"]
        assert st_d[1] >= 2 * n and (n == 0 or st_d[1] < 4 * n + 64), "draws consumed: %d for %d normals" % (st_d[1], n)


def test_values_near_a_rounding_boundary_go_to_the_host_libm():
    """a wide margin (2^-12 relative) sends ~ 2^12 x the usual share of the values through the host's log(): same model"""
    conf = list(dict(num_user=3000, num_item=500, num_global=0, num_factor=64).items())
    host, nxt_h, _ = _init(0, conf, 7, 0)
    dev, nxt_d, st = _init(0, conf, 7, 1, margin=12)
    _same(host, dev)
    assert nxt_h == nxt_d
    assert st[0] > 100, "expected many reported values with a 2^-12 margin, got %d" % st[0]
    dev2, _, st2 = _init(0, conf, 7, 1)
    _same(host, dev2)
    assert st2[0] < 20


def test_several_tiles_and_the_contract_shape():
    """more accepted attempts than one tile holds (2^25 attempts ~ 26 M normals): 600 K x 64 user rows + items"""
    conf = list(dict(num_user=600_000, num_item=100_000, num_global=0, num_factor=64).items())
    host, nxt_h, _ = _init(0, conf, 10, 0)
    dev, nxt_d, st = _init(0, conf, 10, 1)
    _same(host, dev)
    assert nxt_h == nxt_d
    assert st[1] > 2 * (1 << 25)


def test_unusual_libc_state_takes_the_host_loop():
    """initstate() with a 128-byte table (TYPE_2 / 15 words... any mode but the default 31-word one): the device path declines, the host loop runs"""
    buf = ctypes.create_string_buffer(64)
    libc.initstate.restype = ctypes.c_void_p
    libc.initstate.argtypes = [ctypes.c_uint, ctypes.c_char_p, ctypes.c_size_t]
    libc.setstate.restype = ctypes.c_void_p
    libc.setstate.argtypes = [ctypes.c_void_p]
    old = libc.initstate(5, buf, 64)
    try:
        t = sa.Trainer(0, 0)
        for k, v in dict(num_user=20, num_item=10, num_global=0, num_factor=8).items():
            t.set_param(k, str(v))
        t.init_model()
        assert t.counter(14) == 0
        w = t.view("W_user")
        assert w.shape == (20, 8) and np.all(np.isfinite(w)) and np.any(w != 0)
    finally:
        libc.setstate(old)


# ---- load_model straight into HBM (file -> pinned chunks -> device; knob device_load = 0: through a host copy of the model)
@pytest.mark.parametrize("fmt,kw", [(0, dict(num_user=700, num_item=333, num_global=4, num_factor=64)),
                                    (0, dict(num_user=90, num_item=50, num_global=0, num_factor=7)),          # pitch != k: pad floats stay 0
                                    (1, dict(num_user=300, num_item=200, num_global=2, num_factor=33, num_ufeedback=150, ufeedback_init_sigma="0.01")),
                                    (0, dict(num_user=120, num_item=120, num_global=0, num_factor=16, common_latent_space=1, common_feedback_space=1))])
def test_load_model_streams_into_hbm(tmp_path, fmt, kw):
    a = sa.Trainer(fmt, 0)
    a.seed(3)
    for k, v in kw.items():
        a.set_param(k, str(v))
    a.init_model()
    a.init_trainer()
    names = [n for n in ("W_user", "W_item", "W_ufeedback", "u_bias", "i_bias", "g_bias", "ufeedback_bias") if a.view(n) is not None]
    rng = np.random.default_rng(1)
    for n in ("u_bias", "i_bias", "g_bias"):   # biases are 0 after init: give them values
        if a.view(n) is not None and a.view(n).size:
            a.set_view(n, rng.normal(size=a.view(n).shape).astype(np.float32))
    f0 = str(tmp_path / "m0.model")
    a.save_model(f0)
    want = {n: a.view(n).copy() for n in names}
    out = []
    for dev in (0, 1):
        b = sa.Trainer(fmt, 0)
        b.set_knob("device_load", dev)
        b.load_model(f0)
        for k, v in kw.items():
            b.set_param(k, str(v))
        got_before = {n: b.view(n).copy() for n in names}
        b.init_trainer()
        for n in names:
            assert np.array_equal(want[n].view(np.uint32), got_before[n].view(np.uint32)), (dev, n)
            assert np.array_equal(want[n].view(np.uint32), b.view(n).view(np.uint32)), (dev, n)
        f1 = str(tmp_path / ("m1_%d.model" % dev))
        b.save_model(f1)
        out.append(open(f1, "rb").read())
    assert out[0] == out[1] == open(f0, "rb").read()
    # a second model over an existing device model, then training goes on from it
    b.load_model(f0)
    for n in names:
        assert np.array_equal(want[n].view(np.uint32), b.view(n).view(np.uint32)), n


def test_load_model_shape_checks_survive(tmp_path):
    a = sa.Trainer(0, 0)
    for k, v in dict(num_user=30, num_item=20, num_global=0, num_factor=8).items():
        a.set_param(k, str(v))
    a.init_model()
    f0 = str(tmp_path / "m.model")
    a.save_model(f0)
    data = bytearray(open(f0, "rb").read())
    bad = str(tmp_path / "short.model")
    open(bad, "wb").write(bytes(data[:len(data) // 2]))
    b = sa.Trainer(0, 0)
    with pytest.raises(sa.SvdfError, match="load_from_file"):
        b.load_model(bad)
    with pytest.raises(sa.SvdfError):   # no half-loaded model is left behind
        b.save_model(str(tmp_path / "after_failure.model"))
    b.load_model(f0)                    # and the handle takes a good file afterwards
    b.init_trainer()
    assert np.array_equal(a.view("W_user").view(np.uint32), b.view("W_user").view(np.uint32))


def test_the_model_file_written_beside_the_next_pass_equals_the_synchronous_one(tmp_path):
    """svdf_save_model_begin / _end (round 5): the snapshot taken at _begin is what the file holds, whatever trains meanwhile"""
    import cases
    nu, ni, n = 30000, 3000, 600000
    u, i, r = cases.planted_triples(n, nu, ni, seed=8)
    for fmt, conf in ((0, cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=64, num_global=5)),):
        t = sa.Trainer(fmt, 0)
        t.seed(10)
        for k, v in conf:
            t.set_param(k, str(v))
        t.init_model()
        t.init_trainer()
        ds = t.dataset_from_triples(u, i, r)
        t.train_dataset(ds)
        a, b, c = str(tmp_path / "sync.model"), str(tmp_path / "async.model"), str(tmp_path / "after.model")
        t.save_model(a)
        t.save_model_begin(b)
        for _ in range(3):
            t.train_dataset(ds)          # the model moves on while the writer streams the snapshot
        t.save_model_end()
        t.save_model(c)
        assert open(a, "rb").read() == open(b, "rb").read()
        assert open(a, "rb").read() != open(c, "rb").read()
        t.save_model_begin(b)
        fo = sa._libc.fopen(c.encode(), b"wb")
        assert t.lib.svdf_save_model_begin(t.h, fo) != 0 and "has not been ended" in t.lib.svdf_last_error().decode()   # one save in flight per handle
        sa._libc.fclose(fo)
        t.save_model_end()
        t.save_model_end()               # idempotent
