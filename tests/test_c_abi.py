"""The boundary is a C ABI: compile a plain-C program against include/svdfeature_amd.h with gcc and run it.
CPU: host-only handle (config parsing, rand_init, model file) checked against the oracle's bytes.
GPU: the same program trains through per-instance svdf_update_csr calls; model checked against the oracle."""
import os
import subprocess

import numpy as np
import pytest

import cases
from oracle import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c", "c_abi_smoke.c")
LIBDIR = os.path.join(ROOT, "svdfeature_amd")
CONF = [("base_score", "3"), ("learning_rate", "0.005"), ("wd_item", "0.004"), ("wd_user", "0.004"), ("num_user", "50"),
        ("num_item", "40"), ("num_global", "0"), ("num_factor", "12")]


def _build(tmp_path):
    exe = str(tmp_path / "c_abi_smoke")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), SRC, "-o", exe,
                           "-L", LIBDIR, "-lsvdfeature_amd", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def _oracle(train):
    o = oracle.OracleTrainer("port", 0, 0)
    o.seed(10)
    for k, v in CONF:
        o.set_param(k, v)
    o.init_model()
    o.init_trainer()
    if train:
        for r in range(500):
            o.update_csr(float(1 + r % 5), 0, 1, 1, [(r * 7) % 50, (r * 13) % 40], [1.0, 1.0])
    return o


def test_plain_c_consumer_host_only(tmp_path):
    exe = _build(tmp_path)
    out = str(tmp_path / "c.model")
    p = subprocess.run([exe, "-2", out], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert p.returncode == 0, p.stdout.decode()
    ref = str(tmp_path / "o.model")
    _oracle(False).save_model(ref)
    assert open(out, "rb").read() == open(ref, "rb").read()


def test_reference_error_behaviour_exit_minus_one(tmp_path):
    """Default error mode is the reference's: message on stderr, exit(-1) (apex-utils/apex_utils.h:47-58)."""
    prog = tmp_path / "die.c"
    prog.write_text('#include <svdfeature_amd.h>\nint main(void){ svdf_trainer *t = svdf_create(0,0,0,0,-2);'
                    ' svdf_set_param(t, "up:bound", "0"); return 0; }\n')
    exe = str(tmp_path / "die")
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(prog), "-o", exe, "-L", LIBDIR, "-lsvdfeature_amd",
                           "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib"])
    p = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 255 and b"can't give 0 as bound" in p.stderr


@pytest.mark.gpu
def test_plain_c_consumer_trains_on_gpu(tmp_path):
    exe = _build(tmp_path)
    out = str(tmp_path / "c.model")
    p = subprocess.run([exe, "-1", out], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert p.returncode == 0, p.stdout.decode()
    text = p.stdout.decode()
    assert "user feature index exceed bound" in text and "instances 500" in text
    o = _oracle(True)
    ref = str(tmp_path / "o.model")
    o.save_model(ref)
    assert open(out, "rb").read() == open(ref, "rb").read()
    pred = float(text.split("pred ")[1].split()[0])
    assert np.float32(pred) == np.float32(o.predict_csr(0.0, 0, 1, 1, [3, 4], [1.0, 1.0]))
