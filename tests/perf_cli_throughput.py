#!/usr/bin/env python3
"""(not collected by pytest; lives under tests/ because it executes the reference binaries in oracle/_ref)
End-to-end throughput of the reference's trainer CLI: unmodified (oracle/_ref/svd_feature) vs linked
against the MI355X engine (oracle/_ref/svd_feature_amd).  Same config, same binary buffer, wall clock of
whole rounds including the reference's loader thread, per-instance update() calls and model saves."""
import os, subprocess, sys, tempfile, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from svdfeature_amd import data as D

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
NU, NI, K = 1_000_000, 100_000, 64
REFDIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref")
u, i, r = bench.synth_triples(N, NU, NI)
tmp = tempfile.mkdtemp()
buf = os.path.join(tmp, "train.buffer")
t0 = time.time()
with open(buf, "wb") as fo:   # CSR buffer, 1000 rows per block (tools/make_feature_buffer.cpp:42)
    nb = (N + 999) // 1000
    np.array([nb, 1000, 2000], np.int32).tofile(fo)
    for s in range(0, N, 1000):
        e = min(N, s + 1000); m = e - s
        np.array([m, 2 * m], np.int32).tofile(fo)
        base = 2 * np.arange(m, dtype=np.int32)
        ptr = np.empty(3 * m + 1, np.int32); ptr[0:3*m:3] = base; ptr[1:3*m:3] = base; ptr[2:3*m:3] = base + 1; ptr[3*m] = 2 * m
        ptr.tofile(fo); r[s:e].tofile(fo)
        idx = np.empty(2 * m, np.uint32); idx[0::2] = u[s:e]; idx[1::2] = i[s:e]; idx.tofile(fo)
        np.ones(2 * m, np.float32).tofile(fo)
print("buffer written in %.1fs (%d MB)" % (time.time() - t0, os.path.getsize(buf) >> 20), flush=True)
conf = os.path.join(tmp, "run.conf")
with open(conf, "w") as f:
    f.write("base_score = 3\nlearning_rate = 0.005\nwd_item = 0.004\nwd_user = 0.004\nnum_item = %d\nnum_user = %d\nnum_global = 0\n"
            "num_factor = %d\nactive_type = 0\nbuffer_feature = \"%s\"\nmodel_out_folder = \"./\"\n" % (NI, NU, K, buf))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
bulk = os.path.join(tmp, "svdf_train_bulk")   # the patched round loop in plain C (integration/svdf_train_bulk.c): whole passes
subprocess.check_call(["gcc", "-O2", "-std=c99", "-D_POSIX_C_SOURCE=200809L", "-w", "-I", os.path.join(ROOT, "include"),
                       os.path.join(ROOT, "integration", "svdf_train_bulk.c"), "-o", bulk, "-L", os.path.join(ROOT, "svdfeature_amd"),
                       "-lsvdfeature_amd", "-Wl,-rpath," + os.path.join(ROOT, "svdfeature_amd"), "-Wl,-rpath,/opt/rocm/lib"])
res = {}
# "<binary> gpusN": the same binary with amd:gpus=N on its command line (ranks share the GPU when fewer devices are visible: the
# same sharding, windows, exchange kernels and events as N devices -- what it measures on a one-GPU box is the handle's overhead)
for name in ("svdf_train_bulk", "svdf_train_bulk gpus2", "svdf_train_bulk gpus8", "svd_feature_amd", "svd_feature_amd gpus2", "svd_feature"):
    times = {}
    exe, _, gp = name.partition(" gpus")
    extra = ["amd:gpus=%s" % gp] if gp else []
    # the per-instance CLIs: 2 rounds against 0 (the reference takes 4.7 s per round).  The bulk loop reports its own clock over rounds
    # 2..10 (round 1 captures the pass as a hipGraph, ~0.2 s once; process start-up varies by tenths of a second with the number of ranks)
    R0, R = (0, 10) if exe == "svdf_train_bulk" else (0, 2)
    own = None
    for rounds in (R0, R):
        d = os.path.join(tmp, "%s_%d" % (name.replace(" ", "_"), rounds)); os.makedirs(d)
        t0 = time.time()
        p = subprocess.run([bulk if exe == "svdf_train_bulk" else os.path.join(REFDIR, exe), conf, "num_round=%d" % rounds, "silent=1"] + extra, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                           env=dict(os.environ, SVDF_PROFILE="1"))
        times[rounds] = time.time() - t0
        if rounds == R: print(p.stdout.decode()[-400:].strip(), flush=True)
        assert p.returncode == 0, p.stdout.decode()[-2000:]
        for line in p.stdout.decode().splitlines():
            if "seconds per round" in line:
                own = float(line.rsplit(":", 1)[1])
    per_round = own if own is not None else (times[R] - times[R0]) / (R - R0)
    res[name] = {"start_s": times[R0] - R0 * per_round, "s_per_round": per_round, "inst_per_s": N / per_round}
    print(name, json.dumps(res[name]), flush=True)
a = open(os.path.join(tmp, "svd_feature_amd_2", "0002.model"), "rb").read()
b = open(os.path.join(tmp, "svd_feature_2", "0002.model"), "rb").read()
c = open(os.path.join(tmp, "svdf_train_bulk_10", "0002.model"), "rb").read()
print("bulk loop: models byte-identical to the reference CLI's:", c == b, " %.1fx the reference CLI end to end (model save every round included)"
      % (res["svdf_train_bulk"]["inst_per_s"] / res["svd_feature"]["inst_per_s"]))
for nm in ("svdf_train_bulk gpus2", "svdf_train_bulk gpus8", "svd_feature_amd gpus2"):
    print("%s: %.2fx its single-GPU form end to end" % (nm, res[nm]["inst_per_s"] / res[nm.split(" ")[0]]["inst_per_s"]))
print("models after 2 rounds byte-identical:", a == b, " speedup end-to-end: %.1fx" % (res["svd_feature_amd"]["inst_per_s"] / res["svd_feature"]["inst_per_s"]))
