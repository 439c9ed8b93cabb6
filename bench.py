#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: training instances/sec (SGD updates/s), basicMF k=64.

One "step" = one full pass of the apex_svd SGD hot path (SVDTrainTask::update's inner loop,
svd_feature.cpp:220-248) over the synthetic workload of BASELINE configs[1]: 1M users x 100K items,
100M (user, item, rating) triples in uniform random order, k=64 fp32, demo/basicMF hyper-parameters.
Inputs (the scheduled instance stream and the model) are resident in HBM when the timed region
starts; every pass performs all 100M sequentially-consistent SGD updates (the result is bit-identical
to the reference's one-instance-at-a-time loop -- checked in-run on a prefix, see "parity").

    python bench.py                       # 1 GPU, 5 timed passes, 1 warm-up, + the secondary workloads
    python bench.py --gpus N              # spawns its own N ranks (torch.distributed.run) when WORLD_SIZE is unset
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --workload pairwise|svdpp|neighbourhood [--gpus N]   # the other BASELINE configs as the main line

Prints ONE JSON line on rank 0 (contract in the task statement) with extra objects:
  "roofline"      algorithmic HBM bytes (1072 B/instance at k=64) / HIP-event time of
                  the dominant kernel's launches, against the 8 TB/s HBM3E peak
  "cpu_baseline"  the reference's own solver (oracle/_ref/libsvdf_ref.so, kind "reference"; or the C
                  port when that is absent) timed on this box's host, 1 thread, on a prefix sample
  "secondary"     (N=1 default run) BASELINE configs[3] / configs[4] at their configured sizes: pairwise rank pairs
                  k=128 (200 M pairs), SVD++ user blocks k=128, neighbourhood (4 of 10 K global ids) k=128 -- each with
                  value, ms_per_step, roofline, cpu_baseline, in-run parity against the CPU path and, where the data's
                  dependency depth is the bound, the DAG bound (levels x latency of one unit); plus SURVEY 8 f3: the evaluator
                  (evaluate_k64) and the ranker (ranker_k128_positions / _top10) with roofline, CPU baseline and identity check
"""
import argparse
import ctypes as C
import json
import os
import socket
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


# =============================================================================== synthetic data: benchlib/synth.py
from benchlib.synth import Planted, cached, synth_triples, synth_pairs, synth_user_blocks, synth_neighbourhood, _DATA_CACHE  # noqa: E402,F401




# =============================================================================== configuration per workload
def conf_for(a):
    return [("base_score", "3"), ("learning_rate", "0.005"), ("wd_item", "0.004"), ("wd_user", "0.004"),
            ("num_item", str(a.items)), ("num_user", str(a.users)), ("num_global", "0"),
            ("num_factor", str(a.factor)), ("active_type", "0")]   # demo/basicMF/basicMF.conf:4-23


WORKLOADS = {
    # name: (format_type, active_type, default factor, dominant kernel, unit name)
    "basicmf": (0, 0, 64, "k_basicmf_runs_soa<8,2,4,1> at k=64 (a lane group of 8 lanes x 2 chunks walks a RUN of up to 4 consecutive ratings of one item with the item's row in registers; "
                          "one launch per conflict-free level of runs; knob runs_exec=0: k_basicmf_slots<8,2,4>, one instance per lane group), k_basicmf<k/4,G,...> at other widths", "instances/s"),
    "pairwise": (0, 3, 128, "k_fewrow_slots<16,2,1,2> (few-row kernel, 3 rows per pair, 16 lanes x 2 chunks per row)", "pairs/s"),
    "svdpp": (1, 0, 128, "k_svdpp_wave<2,true,true,true,false,8> (one wave per user for the row recurrence + 7 helper waves for the feedback phases)", "instances/s"),
    "neighbourhood": (0, 0, 128, "k_fewrow_gslots<16,2> (few-row kernel stripped for one user id + one item id + up to 4 inline global slots; k_fused<32,1,1,1,...> with knob fewrow_gslots=0)", "instances/s"),
}


def workload_conf(name, a, factor):
    base = [("learning_rate", "0.005"), ("wd_item", "0.004"), ("wd_user", "0.004"), ("num_item", str(a.items)), ("num_user", str(a.users)),
            ("num_factor", str(factor))]
    if name == "basicmf":
        return conf_for(a)
    if name == "pairwise":     # demo/pairwiseRank/pairwiseRank.conf: sigmoid rank loss (active_type 3), no user bias
        return base + [("num_global", "0"), ("no_user_bias", "1")]
    if name == "svdpp":        # demo/implicitFeedback/implicitFeedback.conf
        return base + [("base_score", "3"), ("num_global", "0"), ("num_ufeedback", str(a.items)), ("wd_ufeedback", "0.004")]
    if name == "neighbourhood":   # demo/neighborhoodModel: global neighbourhood weights + wd_global
        return base + [("base_score", "3"), ("num_global", str(a.globals)), ("wd_global", "0.001")]
    raise ValueError(name)


PMC_KERNEL = {"basicmf": "k_basicmf", "pairwise": "k_fewrow_slots", "svdpp": "k_svdpp_wave", "neighbourhood": "k_fewrow_gslots"}


def ranker_roofline(matrix_bytes, cand, dt, nsec, tiles, top_k, tile_traffic, k=128):
    """One launch over the prepared candidate matrix serves a TILE of up to 32 user sections (k_rank_score_tile<NS, MODE>, round 5), so the launch's
    algorithmic bytes are the matrix ONCE plus a score / key slice per section -- not the reference's matrix-per-user stream.  With 32 dots per
    candidate the pass is fp32 VALU work (unfused multiplies and adds in the reference's four-chain order: half the FMA peak), not HBM; both
    fractions are in the line."""
    per_tile = nsec / max(tiles, 1) if tiles else 1.0
    launch_bytes = matrix_bytes + per_tile * cand * 4
    achieved = launch_bytes / (dt * per_tile) / 1e9
    flops = 2.0 * k * cand   # per section: one multiply and one add per element, not fused (the reference's SSE order)
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "kernel": ("k_rank_score_tile<16, 1, ...> (a tile of up to 32 user sections per launch, 16 dots per lane and candidate, two section groups) + "
                       "k_rank_tile_select (top_k of every section of the tile from per-wave minimum keys)" if top_k else
                       "k_rank_score_tile<8, 0, ...> (a tile of up to 32 user sections per launch, 8 dots per lane and candidate, four section groups; positions counted in the pass)"),
            "algorithmic_bytes_per_launch": launch_bytes, "sections_per_launch": per_tile,
            "valu_fp32": {"achieved_tflops": flops / dt / 1e12, "peak_tflops_unfused": 157.3 / 2, "frac": flops / dt / 1e12 / (157.3 / 2),
                          "note": "the scoring pass as arithmetic: 2 k flop per candidate and section over the whole call's time per section"},
            "reference_bytes_per_section": matrix_bytes,
            "reference_stream_equivalent_GBps": matrix_bytes / dt / 1e9,
            "timing": "host clock over the whole svdf_ranker_process_rows call / sections: uploads, opening kernel, scoring pass, selection or counting, readback, "
                      "tied sections on the host's sort pool; reference_stream_equivalent = what streaming the matrix once per section, as the reference does, "
                      "would have to sustain for the same sections/s",
            "traffic": tile_traffic,
            "traffic_source": "profiles/hbm_traffic.json: k_rank_score_tile, (2*FETCH+WRITE)*1024 per launch = per TILE of sections (null until measured for the 32-section tile)"}


def measure_traffic(name, a, log, deadline=None):
    """--pmc: HBM bytes per launch of the workload's dominant kernel, measured NOW: two rocprofv3 passes (--kernel-trace --pmc FETCH_SIZE,
    then WRITE_SIZE: the TCC block cannot hold both, MI355X_MICROARCH.md) over one pass of the same workload in a child process of this
    script; (2 * FETCH_SIZE + WRITE_SIZE) * 1024 per launch -- the correction the guide prescribes for gfx950, checked on known byte
    counts in this engine's access patterns by tools/pmc_calib (profiles/r03_pmc_calibration.txt).  None when rocprofv3 is missing or fails."""
    import csv, glob, shutil, signal, subprocess, tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, None
    per = {}
    size_args = {"basicmf": ["--ratings", str(a.ratings)], "pairwise": ["--pairs", str(a.pairs)],
                 "svdpp": ["--svdpp-users", str(a.svdpp_users), "--svdpp-per-user", str(a.svdpp_per_user)],
                 "neighbourhood": ["--neighbour-rows", str(a.neighbour_rows), "--globals", str(a.globals)]}[name]
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        out = tempfile.mkdtemp(prefix="svdf_pmc_", dir="/tmp")
        cmd = [exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out, "-o", "p", "--", sys.executable, os.path.abspath(__file__),
               "--workload", name, "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--pmc", "off", "--secondary", "", "--users", str(a.users), "--items", str(a.items)] + size_args
        if a.factor and name == a.workload:   # secondary workloads run at their own configured width (128), like in this process
            cmd += ["--factor", str(a.factor)]
        env = dict(os.environ, TMPDIR="/tmp", SVDF_BENCH_PMC_CHILD="1")
        env.pop("SVDF_BENCH_DATA_CACHE_WRITE", None)
        left = 600 if deadline is None else deadline - time.time()
        if left < 20:
            log("%s: PMC pass %s skipped: the time cap of the in-run counter passes is used up" % (name, counter))
            shutil.rmtree(out, ignore_errors=True)
            return None, None
        try:
            p = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                p.wait(timeout=left)
            except subprocess.TimeoutExpired:
                os.killpg(p.pid, signal.SIGKILL)   # exactly the group this call started
                p.wait()
                return None, None
            tot, cnt = 0.0, 0
            for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
                for r in csv.DictReader(open(f)):
                    if PMC_KERNEL[name] in r["Kernel_Name"] and r["Counter_Name"] == counter:
                        tot += float(r["Counter_Value"]); cnt += 1
            if cnt == 0:
                return None, None
            per[counter] = (tot / cnt, cnt)
        finally:
            shutil.rmtree(out, ignore_errors=True)
    traffic = (2.0 * per["FETCH_SIZE"][0] + per["WRITE_SIZE"][0]) * 1024.0
    log("%s: PMC passes: FETCH_SIZE %.1f KB, WRITE_SIZE %.1f KB per launch over %d launches -> %.3f MB per launch" % (
        name, per["FETCH_SIZE"][0], per["WRITE_SIZE"][0], per["FETCH_SIZE"][1], traffic / 1e6))
    return traffic, ("measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) over one pass of this workload "
                     "in a child process, kernels matching '%s', %d launches; (2*FETCH_SIZE + WRITE_SIZE)*1024 per launch" % (PMC_KERNEL[name], per["FETCH_SIZE"][1]))


class HipEvents:
    """HIP events recorded on the engine's own stream (torch.cuda.Event only sees torch's stream)."""

    def __init__(self):
        self.hip = C.CDLL("libamdhip64.so")
        self.hip.hipEventCreate.argtypes = [C.POINTER(C.c_void_p)]
        self.hip.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
        self.hip.hipEventSynchronize.argtypes = [C.c_void_p]
        self.hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]

    def new(self):
        ev = C.c_void_p()
        assert self.hip.hipEventCreate(C.byref(ev)) == 0
        return ev

    def record(self, ev, stream):
        assert self.hip.hipEventRecord(ev, C.c_void_p(stream)) == 0

    def elapsed_ms(self, a, b):
        assert self.hip.hipEventSynchronize(b) == 0
        ms = C.c_float()
        assert self.hip.hipEventElapsedTime(C.byref(ms), a, b) == 0
        return float(ms.value)


def rmse(pred, label):
    d = pred.astype(np.float64) - label.astype(np.float64)
    return float(np.sqrt(np.mean(d * d)))


def make_trainer(sa, name, a, factor, device, oracle_kind=None, extra=()):
    fmt, act = WORKLOADS[name][0], WORKLOADS[name][1]
    if oracle_kind:
        from oracle import oracle   # checker only: never on the measured GPU path
        t = oracle.OracleTrainer(oracle_kind, fmt, act)
    else:
        t = sa.Trainer(fmt, act, device=device)
    t.seed(10)   # svd_feature.cpp:293
    for k, v in list(workload_conf(name, a, factor)) + list(extra):
        t.set_param(k, v)
    t0 = time.perf_counter()
    t.init_model()
    t1 = time.perf_counter()
    t.init_trainer()
    # SVDModel::rand_init: on the device since round 4 (svdf_k_init.hip); the oracle trainers run the reference's own loop
    t.init_model_s, t.init_trainer_s = t1 - t0, time.perf_counter() - t1
    return t


# =============================================================================== CPU baseline + in-run parity
def cpu_baseline_and_parity(sa, name, a, factor, trainer, sample, n_total, log):
    """Times the reference CPU path (1 thread) on a prefix `sample` of the workload and runs the same prefix through the
    GPU bench path; every parameter must be identical bit for bit.  sample: ("triples", u, i, r) | ("pairs", u, p, q) |
    ("blocks", BlockArrays) | ("csr", CSRData).  Returns (cpu_baseline dict, parity dict)."""
    from oracle import oracle   # checker only: never on the measured GPU path
    oracle.build()
    kind = "reference" if oracle.have_reference() else "port"
    cpu = make_trainer(sa, name, a, factor, 0, oracle_kind=kind)
    what = sample[0]
    if what == "triples":
        d, S = sa.CSRData.from_triples(*sample[1:]), len(sample[3])
        ds = trainer.dataset_from_triples(*sample[1:])
    elif what == "pairs":
        d, S = sa.pairs_as_csr(*sample[1:]), len(sample[1])
        ds = trainer.dataset_from_pairs(*sample[1:])
    elif what == "csr":
        d, S = sample[1], sample[1].num_row
        ds = trainer.dataset_from_csr(d)
    else:
        d, S = sample[1].to_blocks(), sample[1].num_row
        ds = trainer.dataset_from_blocks(sample[1])
    t0 = time.time()
    if what == "blocks":
        for b in d:
            cpu.update_block(b)
    else:
        cpu.update_batch(d)
    dt = time.time() - t0
    unit = WORKLOADS[name][4]
    log("%s cpu baseline (%s): %d in %.2fs = %.3f M %s" % (name, kind, S, dt, S / dt / 1e6, unit))
    base = {"value": S / dt, "unit": unit, "cores": 1, "kind": kind,
            "sample": "first %d of the %d %s (same stream, same seed-10 init), 1 pass, data preloaded in memory, "
                      "model init and I/O excluded; host has %d logical cores" % (S, n_total, unit.split("/")[0], os.cpu_count())}
    trainer.train_dataset(ds)
    ok, checked = True, []
    for vname in ("W_item", "i_bias", "u_bias", "W_user", "g_bias", "W_ufeedback", "ufeedback_bias"):
        g, c = trainer.view(vname), cpu.view(vname)
        if g is None or c is None or g.size == 0:
            continue
        checked.append(vname)
        ok = ok and np.array_equal(g.view(np.uint32), c.view(np.uint32))
    par = {"checked": "%s after %d sequential SGD updates vs the %s CPU path" % ("/".join(checked), S, kind), "bit_exact": bool(ok)}
    ds.close()
    cpu.close()
    log("%s parity on the prefix: bit_exact=%s" % (name, ok))
    return base, par


# =============================================================================== one workload
def run_workload(name, a, env, steps, warmup, main_line):
    """Builds the workload, trains warmup + steps passes, returns the result dict (rank 0) or None."""
    import svdfeature_amd as sa
    from svdfeature_amd.multi_gpu import (HipShard, Pairs, ShardedTrainer, defer_tails, shard_block_windows, shard_pair_windows,
                                          shard_windows, split_by_item_range)
    torch, dist, rank, world, local_rank, log = env["torch"], env["dist"], env["rank"], env["world"], env["local_rank"], env["log"]
    factor = a.factor if (main_line and a.factor) else WORKLOADS[name][2]
    unit = WORKLOADS[name][4]
    t0 = time.time()
    quality = {}
    # ---- data (every rank generates the same stream from the same seed, then keeps its shard)
    if name == "basicmf":
        n = a.ratings
        u, i, r = cached(synth_triples, n + 1_000_000, a.users, a.items, 12345 + a.data_seed)
        test = (u[n:n + 200000], i[n:n + 200000], r[n:n + 200000])
        u, i, r = u[:n], i[:n], r[:n]
        per_item = n / max(a.items, 1)
        sample = ("triples", u[:a.cpu_sample], i[:a.cpu_sample], r[:a.cpu_sample])
    elif name == "pairwise":
        n = a.pairs
        u, p, q = cached(synth_pairs, n + 200_000, a.users, a.items, 777 + a.data_seed)
        test = (u[n:], p[n:], q[n:])
        u, p, q = u[:n], p[:n], q[:n]
        per_item = 2.0 * n / max(a.items, 1)
        S = min(n, a.cpu_sample // 4)
        sample = ("pairs", u[:S], p[:S], q[:S])
    elif name == "svdpp":
        nblk = a.svdpp_users
        train, test = cached(synth_user_blocks, nblk, a.svdpp_per_user, a.users, a.items, 4242 + a.data_seed)
        n = train.num_row
        per_item = n / max(a.items, 1)
        sample = ("blocks", train.slice(0, min(nblk, max(1, (a.cpu_sample // 8) // a.svdpp_per_user))))
    else:
        n = a.neighbour_rows
        d_all = cached(synth_neighbourhood, n + 100_000, a.users, a.items, a.globals, 4, 99 + a.data_seed)
        test = d_all.slice_rows(n, n + 100_000)
        d_all = d_all.slice_rows(0, n)
        per_item = n / max(a.items, 1)
        sample = ("csr", d_all.slice_rows(0, min(n, a.cpu_sample // 8)))
    log("%s: synthetic data (%d %s per pass) in %.1fs" % (name, n, unit.split("/")[0], time.time() - t0))

    t0 = time.time()
    # auto: bf16 where it pays -- the all-reduce step at any N (one rank's share 2.98 -> 2.66 ms at N = 8, 22.1 -> 16.0 ms at full size) and the stratified
    # schedule's large windows at 2 ranks (11.4 -> 9.8 ms); its small windows at 4 / 8 ranks are launch-latency bound and run 0 ... 18 % SLOWER with
    # 8-byte row pieces (profiles/r04_shard_scale_probe.txt), so they stay fp32.  The 3-seed contract is the same either way (to 1e-9).
    strat_default = name == "basicmf" and a.exchange in ("auto", "stratified") and world > 1
    contrib_fmt = a.contrib if a.contrib != "auto" else ("fp32" if world == 1 or (strat_default and world > 2) else "bf16")
    contrib = [("amd:contrib", contrib_fmt)] if contrib_fmt != "fp32" else []
    tr = make_trainer(sa, name, a, factor, local_rank, extra=contrib)
    if name == "basicmf" and a.groups_per_wave:
        tr.set_knob("groups_per_wave", a.groups_per_wave)
    tr.set_knob("use_graph", a.use_graph)
    for kv in a.knob:
        kname, value = kv.split("=")
        tr.set_knob(kname, int(value))
    log("%s: model init (libc rand) + upload: %.1fs" % (name, time.time() - t0))

    cpu_base, parity = None, None
    if rank == 0 and not a.no_cpu_baseline:
        cpu_base, parity = cpu_baseline_and_parity(sa, name, a, factor, tr, sample, n, log)
        if world > 1:
            parity = None   # the N-rank result is window-synchronous SGD: accuracy contract, not bit parity
        # fresh model for the measured run: the parity prefix trained this one
        tr.close()
        tr = make_trainer(sa, name, a, factor, local_rank, extra=contrib)
        tr.set_knob("use_graph", a.use_graph)
        if name == "basicmf" and a.groups_per_wave:
            tr.set_knob("groups_per_wave", a.groups_per_wave)
        for kv in a.knob:
            kname, value = kv.split("=")
            tr.set_knob(kname, int(value))

    # ---- schedule the instance stream once and keep it in HBM
    t0 = time.time()
    # N > 1, ratings: the exchange of a window is cut into item-range pieces so that every piece's all-reduce runs while the
    # next piece trains (svdf_item_delta_select; multi_gpu.ShardedTrainer(parts=p)); 1 = one synchronous all-reduce per window
    # how a window is trained when the exchange runs (N > 1 or --force-exchange): "minibatch" = the window-minibatch step
    # (svdf_k_window.hip: user side exact, item side one minibatch step per window; three launches per window), "levels" = the
    # round-2 scheme (exact conflict-free levels per rank, item side stale across ranks only)
    exchanging = world > 1 or a.force_exchange
    minibatch = exchanging and name in ("basicmf", "pairwise", "svdpp") and a.exchange != "levels"
    # ratings on N > 1 ranks: the stratified schedule unless another one is asked for (no all-reduce: DESIGN.md section 6f)
    stratified = exchanging and name == "basicmf" and (a.exchange == "stratified" or (a.exchange == "auto" and world > 1))
    auto_parts = 1 if world <= 2 else 2
    parts = (a.exchange_parts or auto_parts) if (name == "basicmf" and exchanging) else 1
    if a.exchange_transport in ("ipc", "native"):
        parts = 1   # the IPC exchange is synchronous per window (its reduce runs over all links at once; nothing to hide behind pieces)
    adaptor = HipShard(tr, torch, torch.device("cuda", local_rank), parts=parts, minibatch=minibatch)
    if a.windows > 0:
        nwin = a.windows
    else:
        # updates per item per window that keep the accuracy contract: window-minibatch step 32 at any number of ranks
        # (tools/minibatch_calibration.py: the result does not depend on the rank count); level scheme 64 at 2 ranks, 42 at 3-4, 32
        # beyond for ratings (tools/rmse_contract_fullsize.py); 50 for rank pairs (tests/test_multi_rank.py)
        # rank pairs through the window step: calibrated at the demo learning rate on the full configs[4] stream (profiles/r04_pairs_windows_demo_rate.txt:
        # 12 windows per pass = 333 updates per item per window cost 3.9e-4 of held-out pair accuracy and 0.02 % of mean margin against the exact pass --
        # the contract is 3e-3 / 2 % -- and even 3 windows stay inside it); round 3's 32 per window (125 windows) came from a run at ten times that rate
        tgt = (320.0 if name == "pairwise" else 32.0) if minibatch else (50.0 if name == "pairwise" else (64.0 if world <= 2 else (42.0 if world <= 4 else 32.0)))
        nwin = max(1, int(np.ceil(per_item / tgt)))
        if minibatch and name == "svdpp":
            # user-group blocks: the feedback rows bind -- a block of n rows pushes n |value| instance-sized updates into every row of its
            # list at once; sum m_f^2 / sum m_f of that mass is kept at 16 per window (profiles/r04_wstep_calibration.txt; svdf_wunit.cpp)
            rows_of_block = np.diff(train.block_row_ptr).astype(np.float64)
            mass = np.bincount(train.fb_index, weights=np.repeat(rows_of_block, np.diff(train.fb_ptr)) * np.abs(train.fb_value), minlength=a.items)
            nwin = max(nwin, int(np.ceil(float((mass * mass).sum() / max(mass.sum(), 1e-30)) / 16.0)))
    if world == 1 and not a.force_exchange:
        nwin = 1
    if name == "basicmf":
        shards = shard_windows(u, i, r, rank, world, nwin)
    elif name == "pairwise":
        shards = shard_pair_windows(u, p, q, rank, world, nwin)
    elif name == "svdpp":
        shards = shard_block_windows(train, rank, world, nwin) if (world > 1 or nwin > 1) else [train]
    else:
        assert world == 1, "the neighbourhood workload is single-GPU (BASELINE configs[3])"
        shards = [d_all]
    if parts > 1:   # piece p of every window, as a window sequence of its own for the tail deferral
        pieces = [split_by_item_range(uu, ii, rr, a.items, parts) for (uu, ii, rr) in shards]
        per_part = [[pc[q] for pc in pieces] for q in range(parts)]
        if nwin > 1 and a.defer_tails > 0 and not minibatch:
            per_part = [defer_tails(seq, a.users, a.items, a.defer_tails) for seq in per_part]
        shards = [[per_part[q][w] for q in range(parts)] for w in range(nwin)]
    elif nwin > 1 and a.defer_tails > 0 and name in ("basicmf", "pairwise") and not minibatch:
        shards = defer_tails(shards, a.users, a.items, a.defer_tails)
    bpr = 1
    if stratified:
        from svdfeature_amd.multi_gpu import StratifiedTrainer, default_chunks, stratified_plan
        bpr = max(1, a.blocks_per_rank) if world > 1 else 1
        # accuracy defaults from the 3-seed contract at the full configs[2] size (profiles/r04_contract_seeds.txt): below 8 ranks a stratum holds
        # many updates per item and its order is far from the file's -- 8 chunks per pass and <= 16 updates per item per window keep every cell
        # <= 6.0e-5 (4 chunks / 32: up to 1.10e-4 at 2 and 4 ranks); at 8 ranks 4 chunks / 32 measure <= 6.3e-5 and the steps are already short
        # skewed catalogues (hottest item >= 16 x a mean item): 12 chunks at every rank count (3 seeds of Zipf(0.7) at the configs[2] size on 8 ranks:
        # 4 chunks 9.0e-5, 12 chunks 4.4e-5; profiles/r06_contract_zipf_c2.txt)
        if a.chunks <= 0:
            a.chunks = default_chunks(i, a.items, world)
        strat_per_item = a.stratified_per_item if a.stratified_per_item > 0 else (16.0 if world < 8 else 32.0)
        plan = [[adaptor.make_windows(sub) for sub in chunk] for chunk in stratified_plan(u, i, r, rank, world, a.chunks, a.items, strat_per_item, bpr)]
        wins = [w for chunk in plan for sub in chunk for w in sub]
        nwin = len(wins)
        parts = 1
    elif name == "neighbourhood":
        wins = [tr.dataset_from_csr(d_all)]
    else:
        wins = adaptor.make_windows(shards)
    sched_s = time.time() - t0
    flat = [d for w in wins for d in (w if isinstance(w, list) else [w])]
    n_batches = sum(w.num_batches for w in flat)
    alg_bytes = sum(w.algorithmic_bytes for w in flat)
    my_n = sum(w.num_row for w in flat)
    log("%s: scheduled %d into %d conflict-free batches (largest %d) in %.1fs" % (name, my_n, n_batches, max(w.max_batch for w in flat), sched_s))
    use_ipc = a.exchange_transport == "ipc" and world > 1 and minibatch and parts == 1
    # the rank's own RCCL communicator driven from C++ (svdf_rccl.cpp): no torch.distributed call on the data path
    use_native = a.exchange_transport == "native" and world > 1 and minibatch and parts == 1
    if stratified:
        adaptor.set_wire_half(False)
        if use_ipc:
            open_transport_agreed(env, "ipc", lambda: adaptor.ipc_open(dist, rank, world, blocks=world * bpr, barrier=False))
            dist.barrier()   # nobody signals into a page that is not mapped yet
        if use_native:
            open_transport_agreed(env, "native", lambda: adaptor.rccl_open(dist, rank, world))
        st = StratifiedTrainer(adaptor, plan, world, rank, dist if world > 1 else None, blocks_per_rank=bpr)
    else:
        st = ShardedTrainer(adaptor, wins, world, dist, force_exchange=a.force_exchange, half_delta=(a.delta_dtype == "fp16"), parts=parts)
        if use_ipc:
            open_transport_agreed(env, "ipc", lambda: adaptor.ipc_open(dist, rank, world, barrier=False))
            dist.barrier()
        if use_native:
            open_transport_agreed(env, "native", lambda: adaptor.rccl_open(dist, rank, world))

    def sync_all():
        tr.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(warmup):
        st.train_pass()
    ev = HipEvents()
    e0, e1 = ev.new(), ev.new()
    launches0 = tr.counter(1)
    sync_all()
    t0 = time.perf_counter()
    ev.record(e0, tr.stream())
    for _ in range(steps):
        st.train_pass()
    enqueue_s = time.perf_counter() - t0   # the host's share: every launch / collective of the timed passes has been ENQUEUED (nothing waited for)
    ev.record(e1, tr.stream())
    sync_all()
    elapsed = time.perf_counter() - t0
    ev_ms = ev.elapsed_ms(e0, e1)
    launches = tr.counter(1) - launches0
    per_rank = None
    if dist is not None:
        # every rank's own clock, stream time, algorithmic bytes and instance count of the timed region: the line reports the MAX clock
        # (contract) plus the spread over the ranks and the aggregate roofline
        mine_ = torch.tensor([elapsed, ev_ms, float(alg_bytes), float(my_n), float(launches), enqueue_s], dtype=torch.float64, device="cuda")
        allr = [torch.zeros_like(mine_) for _ in range(world)]
        dist.all_gather(allr, mine_)
        per_rank = [[float(x) for x in t_.tolist()] for t_ in allr]
        elapsed = max(pr[0] for pr in per_rank)

    # ---- host cost of a launch with the device queue DRAINED: enqueue_ms_per_pass cannot tell a host-bound pass from a GPU-bound one once thousands
    # of launches are in flight (the launch call blocks when the queue is full, so both read ~ms_per_step).  Here a short prefix of the pass
    # (at most 256 launches: nothing can push back) is enqueued onto an idle stream and only the host side is timed.
    host_enq = None
    if world == 1 and not exchanging and name in ("basicmf", "pairwise") and not os.environ.get("SVDF_BENCH_PMC_CHILD"):   # (not in the --pmc children: their launches are counted)
        m = min(n, 4_000_000)
        tr_main, tr = tr, make_trainer(sa, name, a, factor, local_rank, extra=contrib)   # a trainer of its own: the measured model is not trained further
        tr.set_knob("use_graph", a.use_graph)
        for kv in a.knob:
            tr.set_knob(kv.split("=")[0], int(kv.split("=")[1]))
        small = tr.dataset_from_triples(u[:m], i[:m], r[:m]) if name == "basicmf" else tr.dataset_from_pairs(u[:m], p[:m], q[:m])
        nl = small.num_batches
        if 0 < nl <= 4096:
            best = None
            for _ in range(3):
                tr.synchronize()
                l0 = tr.counter(1)
                t0_ = time.perf_counter()
                tr.train_dataset(small)
                dt_ = time.perf_counter() - t0_
                tr.synchronize()
                per = dt_ / max(tr.counter(1) - l0, 1)
                best = per if best is None else min(best, per)
            lpp = launches / max(steps, 1)
            host_enq = {"us_per_launch": best * 1e6, "launches_measured": nl, "launches_per_pass": lpp, "ms_per_pass_estimate": best * lpp * 1e3,
                        "host_bound": bool(best * lpp > 0.8 * elapsed / steps),
                        "what": "host time to enqueue one launch onto an IDLE stream (a %d-launch prefix of the pass, best of 3); x launches per pass = what the host "
                                "needs for a pass when nothing pushes back -- well below ms_per_step = the pass is GPU-bound" % nl}
        small.close()
        tr.close()
        tr = tr_main

    # ---- one more pass with a HIP event after every phase (outside the timed region): stream time per phase of the exchange
    phase_ms = None
    if exchanging and (name in ("basicmf", "pairwise") or minibatch):
        marks = []

        def mark(phase):
            e = ev.new()
            ev.record(e, tr.stream())
            marks.append((phase, e))
        sync_all()
        first = ev.new()
        ev.record(first, tr.stream())
        st.train_pass(mark)
        sync_all()
        phase_ms = {"compute": 0.0, "pack": 0.0, "allreduce": 0.0, "unpack": 0.0}
        prev = first
        for phase, e in marks:
            phase_ms[phase] += ev.elapsed_ms(prev, e)
            prev = e
        if stratified:
            phase_ms["unpack"] = 0.0
        phase_ms["what"] = ("stream time of ONE extra pass (not in the timed region), HIP events on the trainer's stream after each phase was "
                            "enqueued; compute = %s, pack = %s, allreduce = the collective (incl. waiting for the slowest rank), unpack = %s" % (
                                ("k_window_users (user walks)", "k_window_items (per-item sums into the wire buffer)", "k_delta_addto") if minibatch else
                                ("the conflict-free levels of the window", "k_delta_pack", "k_delta_unpack")))
        steps_done = warmup + steps + 1
    else:
        steps_done = warmup + steps

    if stratified and world > 1:
        st.gather_blocks()   # between passes a rank holds one valid item block: complete the item side before scoring
        sync_all()

    # ---- held-out quality after the run; with N ranks every rank scores the test rows of the users it owns
    def reduce_sum(vals):
        if dist is None:
            return vals
        acc = torch.tensor(vals, dtype=torch.float64, device="cuda")
        dist.all_reduce(acc)
        return [float(x) for x in acc.tolist()]
    if name == "basicmf":
        tu, ti, tl = test
        mine = (tu % world) == rank
        pred = tr.predict_batch(sa.CSRData.from_triples(tu[mine], ti[mine], tl[mine]))
        sse, cnt = reduce_sum([float(np.sum((pred.astype(np.float64) - tl[mine].astype(np.float64)) ** 2)), float(mine.sum())])
        quality = {"rmse_test_after_run": float(np.sqrt(sse / max(cnt, 1.0))), "passes_before_rmse": steps_done}
        if exchanging and rank == 0 and not a.no_sequential_reference:
            # the contract of the exchange (|dRMSE| <= 1e-4): the same passes as exact sequential SGD (the reference's result) on this GPU
            t0 = time.time()
            sq = make_trainer(sa, name, a, factor, local_rank)
            dsq = sq.dataset_from_triples(u, i, r)
            for _ in range(steps_done):
                sq.train_dataset(dsq)
            ps = sq.predict_batch(sa.CSRData.from_triples(tu, ti, tl))
            quality["rmse_sequential_reference"] = rmse(ps, tl)
            quality["rmse_minus_sequential"] = quality["rmse_test_after_run"] - quality["rmse_sequential_reference"]
            dsq.close()
            sq.close()
            log("%s: sequential reference of the same %d passes on rank 0: rmse %.6f (this run %.6f) in %.1fs" % (
                name, steps_done, quality["rmse_sequential_reference"], quality["rmse_test_after_run"], time.time() - t0))
    elif name == "pairwise":
        tu, tp, tq = test
        mine = (tu % world) == rank
        margin = tr.predict_batch(sa.pairs_as_csr(tu[mine], tp[mine], tq[mine]))   # score(pos) - score(neg): active_type 3 predicts the raw score
        right, msum, cnt = reduce_sum([float(np.sum(margin > 0)), float(np.sum(margin, dtype=np.float64)), float(mine.sum())])
        quality = {"pair_accuracy_test_after_run": right / max(cnt, 1.0), "mean_margin_test_after_run": msum / max(cnt, 1.0)}
    elif name == "svdpp":
        owner = test.block_user() % np.uint32(world)
        mine = test.select(owner == rank)
        dt_ = tr.dataset_from_blocks(mine)
        pred = tr.predict_dataset(dt_)
        dt_.close()
        sse, cnt = reduce_sum([float(np.sum((pred.astype(np.float64) - mine.row_label.astype(np.float64)) ** 2)), float(mine.num_row)])
        quality = {"rmse_test_after_run": float(np.sqrt(sse / max(cnt, 1.0))), "passes_before_rmse": steps_done}
        if exchanging and rank == 0 and not a.no_sequential_reference:   # the contract of the exchange: the same passes as exact sequential SGD on this GPU
            sq = make_trainer(sa, name, a, factor, local_rank)
            dsq = sq.dataset_from_blocks(train)
            for _ in range(steps_done):
                sq.train_dataset(dsq)
            dte = sq.dataset_from_blocks(test)
            quality["rmse_sequential_reference"] = rmse(sq.predict_dataset(dte), test.row_label)
            quality["rmse_minus_sequential"] = quality["rmse_test_after_run"] - quality["rmse_sequential_reference"]
            for x in (dsq, dte):
                x.close()
            sq.close()
    else:
        quality = {"rmse_test_after_run": rmse(tr.predict_batch(test), test.row_label)}

    # ---- the dependency (DAG) bound of exact sequential semantics: levels x latency of ONE unit launched alone
    dag = None
    if rank == 0 and world == 1 and name in ("svdpp", "neighbourhood", "pairwise"):
        if name == "svdpp":
            one = tr.dataset_from_blocks(train.slice(0, 1))
        elif name == "pairwise":
            one = tr.dataset_from_pairs(u[:1], p[:1], q[:1])
        else:
            one = tr.dataset_from_csr(d_all.slice_rows(0, 1))
        reps = 200
        for _ in range(20):
            tr.train_dataset(one)
        tr.synchronize()
        ev.record(e0, tr.stream())
        for _ in range(reps):
            tr.train_dataset(one)
        ev.record(e1, tr.stream())
        lat_us = ev.elapsed_ms(e0, e1) * 1e3 / reps
        one.close()
        dag = {"levels_per_pass": n_batches, "unit_latency_us": lat_us, "bound_ms_per_pass": n_batches * lat_us * 1e-3,
               "measured_over_bound": (elapsed * 1e3 / steps) / max(n_batches * lat_us * 1e-3, 1e-9),
               "what": "conflict-free levels of exact sequential semantics x the latency of one %s launched alone "
                       "(back to back on one stream); a pass cannot be faster than this however fast the kernel streams"
                       % ("user unit (%d rows + its feedback list)" % a.svdpp_per_user if name == "svdpp" else "instance")}

    res = None
    if rank == 0:
        value = steps * n / elapsed
        # dominant kernel: one launch per conflict-free batch.  algorithmic bytes per launch and average launch duration
        # measured with HIP events on the engine's stream over the timed region (launch gaps included, so this is a lower
        # bound on the in-kernel rate).
        per_launch_bytes = alg_bytes * steps / max(launches, 1)
        per_launch_us = ev_ms * 1e3 / max(launches, 1)
        achieved = per_launch_bytes / (per_launch_us * 1e-6) / 1e9
        traffic, traffic_src = None, None
        tfile = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if world == 1 and not exchanging and os.path.exists(tfile):
            try:
                traffic = json.load(open(tfile)).get(name if name != "basicmf" else "hbm_bytes_per_launch")
                if isinstance(traffic, dict):
                    traffic = traffic.get("hbm_bytes_per_launch")
                traffic_src = "profiles/hbm_traffic.json (builder's rocprofv3 PMC run of this command, not measured in this run)"
            except Exception:
                traffic = None
        if world == 1 and not exchanging and getattr(a, "pmc_results", {}).get(name, (None, None))[0] is not None:
            traffic, traffic_src = a.pmc_results[name]
        res = {
            "value": value, "unit": unit, "ms_per_step": elapsed * 1e3 / steps,
            "workload": {"basicmf": "basicMF synthetic %dx%d, %d ratings, k=%d fp32 (BASELINE configs[%d])" % (a.users, a.items, n, factor, 1 if world == 1 else 2),
                         "pairwise": "pairwiseRank synthetic %dx%d, %d (user, pos, neg) pairs, k=%d fp32, active_type=3, no user bias (BASELINE configs[4])" % (a.users, a.items, n, factor),
                         "svdpp": "implicitFeedback (SVD++) %d users x %d ratings, feedback set = own items, k=%d fp32 (BASELINE configs[3])" % (a.svdpp_users, a.svdpp_per_user, factor),
                         "neighbourhood": "neighborhoodModel shape: %d ratings + 4 of %d global ids each, k=%d fp32 (BASELINE configs[3])" % (n, a.globals, factor)}[name],
            "order": "uniform random (file order preserved: result == sequential SGD)" if not exchanging else
                     ("stratified: %d file-order chunks x %d steps per pass, in step t rank r trains (user block r) x (item block (%d r + t) %% %d) with the "
                      "window-minibatch step, the item block is handed to rank r - 1 afterwards (fp32, no all-reduce); %d window steps per rank and pass"
                      % (a.chunks, world * bpr, bpr, world * bpr, nwin)) if stratified else
                     "user-sharded, %s, item-delta all-reduce (%s on the wire) every 1/%d pass%s" % (
                         "window-minibatch step (user side exact, item side applied at the window's end)" if minibatch else "exact conflict-free levels per rank",
                         a.delta_dtype, nwin, (", in %d item-range pieces overlapped with training" % parts) if parts > 1 else ""),
            "exchange": None if not exchanging else {
                "path": ("torch.distributed %s: ring hand-over of item blocks (batch_isend_irecv, rank r -> r - 1) after every sub-epoch, broadcasts of the "
                         "blocks before scoring; no all-reduce" % dist.get_backend()) if (stratified and dist is not None and world > 1) else
                        ("torch.distributed %s all_reduce (RCCL over xGMI)" % dist.get_backend()) if (dist is not None and world > 1) else
                        ("torch.distributed %s all_reduce with one rank (identity)" % dist.get_backend() if dist is not None else "none (one rank)"),
                "transport": ("ipc: IPC-mapped wire buffers / inboxes, k_delta_reduce_gather + sequence flags (svdf_ipc.cpp); torch.distributed only carries the handles, "
                              "the final block broadcasts and the timing reductions") if use_ipc else
                             ("native: the rank's own RCCL communicator driven from C++ (svdf_rccl.cpp: ncclAllReduce / ncclSend + ncclRecv on a side stream, ordered by "
                              "HIP events, one C call per hand-over); torch.distributed only carries the unique id, the final block broadcasts and the timing reductions")
                             if use_native else "torch.distributed",
                "world_size_reported": (dist.get_world_size() if dist is not None else 1),   # what the process group (RCCL with backend nccl) says, not what --gpus asked for
                "contributions": contrib_fmt if minibatch else None,
                "step": "stratified" if stratified else ("minibatch" if minibatch else "levels"), "windows": nwin, "parts": parts,
                "handoffs_per_pass": a.chunks * world * bpr if (stratified and world > 1) else 0,
                "bytes_per_window": int(tr.item_delta_count() * (2 if a.delta_dtype == "fp16" else 4)) if not stratified else
                                    int(tr.item_delta_count() * 4 // max(world * bpr, 1)),
                "updates_per_item_per_window": per_item / nwin},
            "phase_ms": phase_ms,
            "enqueue_ms_per_pass": enqueue_s * 1e3 / steps,   # host time until a pass is enqueued; under queue back-pressure (thousands of launches in flight) this reads ~ms_per_step whoever is the limit: host_enqueue is the field that tells
            "host_enqueue": host_enq,
            # N > 1: the spread of the ranks' own clocks over the timed region, the aggregate roofline (sum of the ranks' algorithmic bytes
            # over the contract's max-over-ranks time against N x 8 TB/s) and what DESIGN.md's model expects for this line
            "per_rank_ms": None if (per_rank is None or world == 1) else {
                "min": min(pr[0] for pr in per_rank) * 1e3 / steps, "max": max(pr[0] for pr in per_rank) * 1e3 / steps,
                "stream_min": min(pr[1] for pr in per_rank) / steps, "stream_max": max(pr[1] for pr in per_rank) / steps,
                "instances_min": min(pr[3] for pr in per_rank), "instances_max": max(pr[3] for pr in per_rank),
                "enqueue_min": min(pr[5] for pr in per_rank) * 1e3 / steps, "enqueue_max": max(pr[5] for pr in per_rank) * 1e3 / steps,
                "what": "per pass; min / max over the ranks of the host clock between the two barriers (the line's ms_per_step is the max), of the HIP-event time on each rank's "
                        "stream, and of the host time until every launch / collective of the pass was ENQUEUED (enqueue well below the clock = the host thread was NOT the limit; "
                        "close to the clock = either host-bound or the device queue filled up and pushed back, as it does over the ~1 900 launches of an exact one-GPU pass)"},
            "roofline_aggregate": None if (per_rank is None or world == 1) else {
                "bound": "hbm", "achieved": sum(pr[2] for pr in per_rank) * steps / elapsed / 1e9, "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                "frac": sum(pr[2] for pr in per_rank) * steps / elapsed / 1e9 / (HBM_PEAK_GBS * world),
                "what": "sum over the ranks of the algorithmic bytes of their passes (SURVEY 8d4 per instance + the window step's contribution slots) / the max-over-ranks elapsed time, against N x 8 TB/s"},
            "model_ms": None if not exchanging else model_ms(
                name, world, "stratified" if stratified else ("minibatch" if minibatch else "levels"), n, a.items, factor, nwin,
                world * bpr if stratified else 1, a.chunks * world * bpr if (stratified and world > 1) else 0,
                (17.1 * n / 1e8 * factor / 64.0) if name == "basicmf" else (177.3 * n / 2e8 * factor / 128.0 if name == "pairwise" else None)),
            "conflict_free_batches_per_pass": n_batches, "schedule_build_s": round(sched_s, 2),
            "parallelism": "1 GPU" if world == 1 else "dp%d user shards + RCCL all-reduce" % world,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": (("k_window_users_slots<8,2,G,1,0,true> + k_window_items<16,HALF> + k_delta_addto<HALF>" if name == "basicmf" else
                                     "k_window_users_slots<16,2,1,2,3,false> + k_window_items<32,HALF> + k_delta_addto<HALF>") +
                                    " (window-minibatch step, 3 launches per window)" if (minibatch and name != "svdpp") else
                                    ("k_wunit_wave<2,8,BF16,true> + k_wunit_sum<32,HALF,false> + k_delta_addto<HALF> (window-minibatch step for user units, one wave per unit, 3 launches per window)"
                                     if minibatch else WORKLOADS[name][3])), "launches": launches, "avg_launch_us": per_launch_us,
                         "algorithmic_bytes_per_launch": per_launch_bytes,
                         "algorithmic_bytes_per_instance": alg_bytes / max(my_n, 1)},
            # what bounds one launch of this kernel: its measured HBM traffic at the achievable streaming rate + one dependent
            # kernel boundary (/opt/skills/guides/MI355X_MICROARCH.md: 6.3 TB/s achievable, 1.7-1.9 us between streaming kernels)
            "launch_model": None if not traffic else {
                "hbm_traffic_bytes_per_launch": traffic, "stream_us_at_6300GBps": traffic / 6.3e12 * 1e6, "boundary_us": 1.8,
                "model_us": traffic / 6.3e12 * 1e6 + 1.8, "measured_us": per_launch_us,
                "measured_over_model": per_launch_us / (traffic / 6.3e12 * 1e6 + 1.8)},
            "cpu_baseline": cpu_base, "parity": parity, "dag_bound": dag,
            # end to end: what a training run of `rounds` passes sees when the one-off schedule build is counted in
            "end_to_end": {"rounds": 40, "value": 40 * n / (sched_s + 40 * elapsed / steps), "unit": unit,
                           "what": "40 passes (demo/basicMF num_round) + the one-off schedule build + upload of this run"},
            # SVDModel::rand_init of this run's model: on the device (svdf_k_init.hip), the reference's draws and values bit for bit
            # (the in-run parity above starts the reference's classes from their own rand_init: equal results = equal start)
            "init_model": {"seconds": getattr(tr, "init_model_s", None), "rand_draws": tr.counter(14), "values_decided_by_host_libm": tr.counter(13),
                           "where": "device" if tr.counter(14) > 0 else "host loop"},
        }
        res.update(quality)
    if use_ipc:
        if dist is not None:
            dist.barrier()   # nobody unmaps while a peer may still read
        tr.ipc_close()
    if use_native:
        if dist is not None:
            dist.barrier()
        adaptor.rccl_close()
    for w in flat:
        w.close()
    tr.close()
    return res


def run_f3_secondary(a, env):
    """SURVEY 8 f3 next to the training numbers (N=1 default run only): the evaluator (svdf_eval_dataset = RMSEEvaluator over a
    resident test set, the read-only half of the hot path) and the ranker (svdf_ranker_process_rows, 100 K candidates, k=128),
    each with its HBM roofline figure, the CPU path on a bounded sample and an in-run identity check."""
    import tempfile
    import svdfeature_amd as sa
    from oracle import oracle   # checker / CPU baseline only
    oracle.build()
    log = env["log"]
    out = {}
    try:   # HBM traffic of the f3 kernels from the builder's PMC passes (tools/r03_pmc_f3.sh), per launch
        f3_traffic = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
    except Exception:
        f3_traffic = {}

    def traffic_of(key):
        v = f3_traffic.get(key)
        return (v or {}).get("hbm_bytes_per_launch") if isinstance(v, dict) else None
    # ---- evaluator: basicMF shape of the main line, 20 M held-out-style ratings
    n, k = 20_000_000, 64
    rng = np.random.default_rng(31)
    u = rng.integers(0, a.users, n, dtype=np.uint32)
    i = rng.integers(0, a.items, n, dtype=np.uint32)
    r = rng.integers(1, 6, n).astype(np.float32)
    conf = [(kk, v) for kk, v in conf_for(a) if kk != "num_factor"] + [("num_factor", str(k)), ("ui_init_sigma", "0.1")]
    f3_kind = "reference" if oracle.have_reference() else "port"   # the reference's own classes where oracle/_ref is present
    t, o = sa.Trainer(0, 0), oracle.OracleTrainer(f3_kind, 0, 0)
    for x in (t, o):
        x.seed(10)
        for kk, v in conf:
            x.set_param(kk, v)
        x.init_model()
        x.init_trainer()
    ds = t.dataset_from_triples(u, i, r)
    t.eval_dataset(ds)
    reps = 10
    t0 = time.time()
    for _ in range(reps):
        sse, cnt = t.eval_dataset(ds)
    dt = (time.time() - t0) / reps
    S = 1_000_000
    t0 = time.time()
    cpu_pred = np.asarray(o.predict_batch(sa.CSRData.from_triples(u[:S], i[:S], r[:S])), np.float32)
    dt_cpu = time.time() - t0
    gpu_pred = t.predict_dataset(ds)[:S]
    byts = 2 * k * 4 + 8 + 12   # two rows, two bias words, the (user, item, label) record
    out["evaluate_k64"] = {
        "workload": "RMSEEvaluator over %d resident (user, item, rating) instances, basicMF k=64" % n,
        "value": n / dt, "unit": "instances/s", "ms_per_step": dt * 1e3, "rmse": float(np.sqrt(sse / cnt)),
        "roofline": {"bound": "hbm", "achieved": n * byts / dt / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": n * byts / dt / 1e9 / HBM_PEAK_GBS,
                     "kernel": "k_predict_basic", "algorithmic_bytes_per_instance": byts,
                     "timing": "host clock around svdf_eval_dataset (one launch + the partial-sum reduction + sync)",
                     "traffic": traffic_of("evaluate_k64"), "traffic_source": "profiles/hbm_traffic.json: k_predict_basic over the same 20 M instances, (2*FETCH+WRITE)*1024 per launch"},
        "cpu_baseline": {"value": S / dt_cpu, "unit": "instances/s", "cores": 1, "kind": f3_kind, "sample": "predict of the first %d instances" % S},
        "parity": {"predictions_bit_exact_on_sample": bool(np.array_equal(gpu_pred.view(np.uint32), cpu_pred.view(np.uint32)))}}
    log("f3 evaluate: %.2f G inst/s (%.1f%% of peak), cpu %.2f M inst/s" % (n / dt / 1e9, 100 * out["evaluate_k64"]["roofline"]["frac"], S / dt_cpu / 1e6))
    path = os.path.join(tempfile.mkdtemp(), "rank.model")
    t.close()
    o.close()
    # ---- ranker: 100 K candidates, k=128, 5 positives per user section, positions and top-10
    cand, k, nsec, ncpu = 100_000, 128, 1200, 12   # enough sections that the rare tied ones (an 8 ms host sort each, on helper threads) do not set the wall clock
    tr = oracle.OracleTrainer("port", 0, 0)
    tr.seed(7)
    for kk, v in [("num_user", "100000"), ("num_item", str(cand)), ("num_factor", str(k)), ("base_score", "0"), ("ui_init_sigma", "0.1"),
                  ("num_global", "0"), ("active_type", "0")]:
        tr.set_param(kk, v)
    tr.init_model()
    tr.init_trainer()
    tr.save_model(path)
    items = sa.CSRData.from_rows([(0.0, [], [], [(c, 1.0)]) for c in range(cand)])
    secs = []
    for s_ in range(nsec):
        pos = rng.choice(cand, size=5, replace=False)
        secs.append(sa.CSRData.from_rows([(2.0, [], [(int(rng.integers(0, 100_000)), 1.0)], []), (1.0, [], [(int(x), 1.0) for x in pos], []),
                                          (4.0, [], [], [])]))
    bulk = sa.CSRData.concat(secs)
    byts = cand * (k * 4 + 4 + 4 + 1)
    for top_k in (0, 10):
        g = sa.Ranker(0, 0)
        c = oracle.OracleRanker(f3_kind, 0, 0)
        for x in (g, c):
            x.set_param("top_k", str(top_k))
            x.load_model(path)
            x.init_ranker(cand)
            x.process_rows(items)
        g.process_rows(secs[0])
        t0 = time.time()
        got = g.process_rows(bulk)
        dt = (time.time() - t0) / nsec
        t0 = time.time()
        ref = np.concatenate([c.process_rows(s_) for s_ in secs[:ncpu]])
        dt_cpu = (time.time() - t0) / ncpu
        out["ranker_k128_top%d" % top_k if top_k else "ranker_k128_positions"] = {
            "workload": "ISVDRanker: %d candidates, k=128, %d user sections in one svdf_ranker_process_rows call, %s" % (
                cand, nsec, "top_k=%d" % top_k if top_k else "rank positions of 5 positives"),
            "value": 1.0 / dt, "unit": "user sections/s", "ms_per_step": dt * 1e3, "sections_finished_by_host_sort": g.counter(1),
            "tiles_of_up_to_32_sections": g.counter(3),
            "roofline": ranker_roofline(byts, cand, dt, nsec, g.counter(3), top_k, traffic_of("ranker_k128_positions_tile")),
            "cpu_baseline": {"value": 1.0 / dt_cpu, "unit": "user sections/s", "cores": 1, "kind": f3_kind, "sample": "the first %d sections" % ncpu},
            "parity": {"results_identical_on_sample": bool(np.array_equal(got[:len(ref)], ref))}}
        log("f3 ranker top_k=%d: %.1f us/section, cpu %.2f ms/section" % (top_k, dt * 1e6, dt_cpu * 1e3))
        g.close()
    return out


# =============================================================================== N > 1: watchdog, rendezvous ladder, preflight, model: benchlib/multi.py
from benchlib.multi import (ATTEMPT_ENV, FALLBACK_ENV, LADDER, Watchdog, fallback_log, escalate, rendezvous, store_agree, preflight,  # noqa: E402,F401
                            TransportUnavailable, open_transport_agreed, choose_schedule, model_ms)




def run_window_step(sa, name, a, device, log, steps=3, warmup=1, seq_quality=None):
    """secondary.<workload>_window_step: BASELINE configs[3] through the OPT-IN window-minibatch step on ONE GPU (`amd:step = minibatch`;
    svdf_k_wunit.hip) -- next to, never instead of, the exact line.  A user's unit is exact on its private state,
    the shared rows (W_item / i_bias, W_ufeedback / ufeedback_bias, g_bias) move once per window; the result is NOT the reference's bit for
    bit, the contract is |dRMSE| <= 1e-4 against the exact pass of the same epochs, measured here (rmse_minus_sequential)."""
    factor = WORKLOADS[name][2]
    tri = None
    if name == "basicmf":   # the contract workload through the same opt-in step (k_window_users_slots + k_window_items in place), bf16 contribution rows
        n = a.ratings
        u, i, r = cached(synth_triples, n + 1_000_000, a.users, a.items, 12345 + a.data_seed)
        test = sa.CSRData.from_triples(u[n:n + 200000], i[n:n + 200000], r[n:n + 200000])
        tri = (u[:n], i[:n], r[:n])
    elif name == "pairwise":   # configs[4]: two signed item entries per instance; windows from the demo-rate calibration (320 updates per item per window)
        n = a.pairs
        pu, pp, pq = cached(synth_pairs, n + 200_000, a.users, a.items, 777 + a.data_seed)
        test = sa.pairs_as_csr(pu[n:], pp[n:], pq[n:])
        tri = (pu[:n], pp[:n], pq[:n])
    elif name == "svdpp":
        train, test = cached(synth_user_blocks, a.svdpp_users, a.svdpp_per_user, a.users, a.items, 4242 + a.data_seed)
        n = train.num_row
    else:
        n = a.neighbour_rows
        d_all = cached(synth_neighbourhood, n + 100_000, a.users, a.items, a.globals, 4, 99 + a.data_seed)
        test = d_all.slice_rows(n, n + 100_000)
        d_all = d_all.slice_rows(0, n)
    extra = [("amd:step", "minibatch")] + ([("amd:contrib", "bf16")] if (getattr(a, "contrib", "fp32") == "bf16" or (name in ("basicmf", "pairwise", "svdpp") and getattr(a, "contrib", "auto") == "auto")) else [])
    if a.step_window > 0:
        extra.append(("amd:window", str(a.step_window)))
    elif name == "pairwise":
        nw = max(1, int(np.ceil(2.0 * n / max(a.items, 1) / 320.0)))
        extra.append(("amd:window", str(-(-n // nw))))
    t0 = time.time()
    tr = make_trainer(sa, name, a, factor, device, extra=extra)
    if a.step_per_target > 0:
        tr.set_knob("window_per_target", a.step_per_target)
        tr.set_knob("window_per_target_fb", a.step_per_target)
    if os.environ.get("SVDF_WUNIT_FAST"):   # A/B of the user-unit kernels (tools/wstep_probe.py): 0 lane groups, 1 slot kernel, 2 one wave per unit
        tr.set_knob("wunit_fast", int(os.environ["SVDF_WUNIT_FAST"]))
    for env, knob in (("SVDF_WUNIT_INPLACE", "wunit_inplace"), ("SVDF_WUNIT_DEFER_FB", "wunit_defer_fb")):
        if os.environ.get(env):
            tr.set_knob(knob, int(os.environ[env]))
    ds = (tr.dataset_from_pairs(*tri) if name == "pairwise" else tr.dataset_from_triples(*tri)) if tri else (tr.dataset_from_blocks(train) if name == "svdpp" else tr.dataset_from_csr(d_all))
    build_s = time.time() - t0
    assert ds.kind == 8
    ev = HipEvents()
    e0, e1 = ev.new(), ev.new()
    for _ in range(warmup):
        tr.train_dataset(ds)
    tr.synchronize()
    launches0 = tr.counter(1)
    t0 = time.perf_counter()
    ev.record(e0, tr.stream())
    for _ in range(steps):
        tr.train_dataset(ds)
    ev.record(e1, tr.stream())
    tr.synchronize()
    elapsed = time.perf_counter() - t0
    ev_ms = ev.elapsed_ms(e0, e1)
    launches = tr.counter(1) - launches0

    def score(t):
        if name == "pairwise":   # (held-out pair accuracy, mean margin): active_type 3 predicts the raw score difference
            m = t.predict_batch(test)
            return (float(np.mean(m > 0)), float(np.mean(m, dtype=np.float64)))
        if name == "svdpp":
            dt_ = t.dataset_from_blocks(test)
            p = t.predict_dataset(dt_)
            dt_.close()
            return rmse(p, test.row_label)
        return rmse(t.predict_batch(test), test.row_label)
    # the same epochs as exact sequential SGD (the reference's result) on this GPU; the minibatch model is scored through an exact-mode twin
    # that loads its model file (a minibatch handle builds window sequences, which are training sets)
    import tempfile
    sq = dsq = None
    if seq_quality is not None and seq_quality[0] == warmup + steps:
        rm_seq = seq_quality[1]   # the exact secondary of this run trained the same passes over the same stream (7 s per pass at 1 M SVD++ users: not twice)
    else:
        sq = make_trainer(sa, name, a, factor, device)
        dsq = (sq.dataset_from_pairs(*tri) if name == "pairwise" else sq.dataset_from_triples(*tri)) if tri else (sq.dataset_from_blocks(train) if name == "svdpp" else sq.dataset_from_csr(d_all))
        for _ in range(warmup + steps):
            sq.train_dataset(dsq)
        rm_seq = score(sq)
    path = os.path.join(tempfile.mkdtemp(), "wstep.model")
    tr.save_model(path)
    tw = sa.Trainer(WORKLOADS[name][0], WORKLOADS[name][1], device=device)
    tw.load_model(path)
    tw.init_trainer()
    rm_run = score(tw)
    os.remove(path)
    alg = ds.algorithmic_bytes
    try:
        wtraffic = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json"))).get(name + "_window_step", {}).get("hbm_bytes_per_launch")
    except Exception:
        wtraffic = None
    res = {"value": steps * n / elapsed, "unit": WORKLOADS[name][4], "ms_per_step": elapsed * 1e3 / steps, "steps": steps, "warmup": warmup,
           "windows_per_pass": ds.num_batches, "launches_per_pass": launches / steps, "build_s": round(build_s, 2),
           "semantics": "OPT-IN amd:step = minibatch: window-minibatch SGD (user units exact, shared rows applied at the window's end, deterministic), NOT the reference's "
                        "sequential result; contract |dRMSE| <= 1e-4 against the exact pass of the same epochs",
           **({"pair_accuracy_test_after_run": rm_run[0], "mean_margin_test_after_run": rm_run[1], "pair_accuracy_sequential_reference": rm_seq[0],
               "mean_margin_sequential_reference": rm_seq[1], "pair_accuracy_minus_sequential": rm_run[0] - rm_seq[0],
               "mean_margin_relative_change": rm_run[1] / rm_seq[1] - 1.0, "contract": "accuracy within 3e-3, mean margin within 2 % of the exact pass (DESIGN.md 6b)"}
              if name == "pairwise" else
              {"rmse_test_after_run": rm_run, "rmse_sequential_reference": rm_seq, "rmse_minus_sequential": rm_run - rm_seq}), "passes_before_rmse": warmup + steps,
           "roofline": {"bound": "hbm", "achieved": alg * steps / (ev_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": alg * steps / (ev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "kernel": ("k_wunit_wave<2,8,BF16,true> (one wave per user unit)" if name == "svdpp" else "k_wunit_fast<16,2,false,4,1>" if name == "neighbourhood" else "k_wunit_fast<LANES,2,false,0,4>") + " + k_wunit_sum<32,false,true> (two launches per window)",
                        "launches": launches, "avg_launch_us": ev_ms * 1e3 / max(launches, 1), "algorithmic_bytes_per_launch": alg * steps / max(launches, 1),
                        "algorithmic_bytes_per_instance": alg / max(n, 1), "traffic": wtraffic,
                        "traffic_source": "profiles/hbm_traffic.json: (2*FETCH_SIZE + WRITE_SIZE)*1024 of the unit kernel (k_wunit_wave / k_wunit_fast) + k_wunit_sum per window / 2 launches (builder's rocprofv3 PMC passes, tools/profile_round.sh)"}}
    log("%s window step: %.2f ms per pass = %.1f M inst/s (%.1f%% of peak), %d windows, quality %s vs sequential %s" % (
        name, res["ms_per_step"], res["value"] / 1e6, 100 * res["roofline"]["frac"], ds.num_batches, rm_run, rm_seq))
    for x in (ds, dsq):
        if x is not None:
            x.close()
    for t in (tr, sq, tw):
        if t is not None:
            t.close()
    return res


def run_single_process_handle(sa, a, world, xch, device, log):
    """secondary.single_process_handle: ONE svdf_trainer with amd:gpus = N (svdf_multi.cpp: N engines on N devices, one host thread per rank,
    HIP events between the ranks' streams) trains the main line's ratings from a resident data set -- the window-minibatch step with the
    handle's own exchange: amd:exchange = p2p (hipDeviceEnablePeerAccess + k_delta_reduce_gather through peer pointers over all xGMI links)
    or rccl (ncclCommInitAll + grouped ncclAllReduce).  Host clock around svdf_train_dataset + synchronize, 1 warm-up + 3 timed passes."""
    n = a.ratings
    u, i, r = cached(synth_triples, n + 1_000_000, a.users, a.items, 12345 + a.data_seed)
    test = (u[n:n + 200000], i[n:n + 200000], r[n:n + 200000])
    u, i, r = u[:n], i[:n], r[:n]
    nwin = max(1, int(np.ceil(n / max(a.items, 1) / 32.0)))
    t0 = time.time()
    t = sa.Trainer(0, 0, device=device)
    t.seed(10)
    for k, v in conf_for(a) + [("amd:gpus", str(world)), ("amd:exchange", xch), ("amd:step", "minibatch"), ("amd:window", str(-(-n // nwin)))]:
        t.set_param(k, v)
    t.init_model()
    t.init_trainer()
    ds = t.dataset_from_triples(u, i, r)
    build_s = time.time() - t0
    t.train_dataset(ds)
    t.synchronize()
    steps = 3
    t0 = time.perf_counter()
    for _ in range(steps):
        t.train_dataset(ds)
    t.synchronize()
    dt = (time.perf_counter() - t0) / steps
    pred = t.predict_batch(sa.CSRData.from_triples(*test))
    res = {"value": n / dt, "unit": "instances/s", "ms_per_step": dt * 1e3, "steps": steps, "warmup": 1, "windows": nwin,
           "exchanges": t.counter(8), "exchange_path": "rccl" if t.counter(12) == 1 else "p2p", "devices_distinct": bool(t.counter(10)),
           "window_minibatch_windows": t.counter(11), "rmse_test_after_run": rmse(pred, test[2]), "passes_before_rmse": steps + 1,
           "build_s": round(build_s, 1),
           "what": "one process, one C-ABI handle, amd:gpus = %d, amd:exchange = %s, fp16 wire format, resident data set sharded by user inside the handle" % (world, xch)}
    log("single-process handle (%s): %.2f ms per pass = %.2f G inst/s, rmse %.6f" % (xch, dt * 1e3, n / dt / 1e9, res["rmse_test_after_run"]))
    ds.close()
    t.close()
    return res


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=list(WORKLOADS), default="basicmf", help="which BASELINE config is the main JSON line")
    ap.add_argument("--ratings", type=int, default=100_000_000)
    ap.add_argument("--pairs", type=int, default=200_000_000, help="pairwise workload: rank pairs per pass (BASELINE configs[4])")
    ap.add_argument("--svdpp-users", type=int, default=1_000_000, help="SVD++ workload: user blocks per pass (round 5: 1 M x 100 = 100 M instances; rounds 1-4 ran 40 K)")
    ap.add_argument("--svdpp-per-user", type=int, default=100)
    ap.add_argument("--neighbour-rows", type=int, default=4_000_000)
    ap.add_argument("--globals", type=int, default=10_000)
    ap.add_argument("--users", type=int, default=1_000_000)
    ap.add_argument("--items", type=int, default=100_000)
    ap.add_argument("--factor", type=int, default=0, help="0 = the workload's configured width (64 for basicMF, 128 otherwise)")
    ap.add_argument("--windows", type=int, default=0,
                    help="item-delta exchanges per pass when --gpus > 1 (0 = chosen from the data density so that the "
                         "RMSE stays within 1e-4 of the sequential reference: about 64 ratings per item per window at "
                         "2 ranks, 32 at 4+ ranks; calibration in DESIGN.md section 6)")
    ap.add_argument("--exchange-parts", type=int, default=0,
                    help="N>1, ratings: item-range pieces per window exchange; piece p's all-reduce overlaps with training piece p+1 "
                         "(1 = one synchronous all-reduce per window; 0 = auto: 1 at 2 ranks, 2 beyond -- a piece keeps the window's item-chain "
                         "depth, so pieces double a rank's launches: 12.9 -> 32.7 ms per pass at 2 ranks, 7.9 -> 10.6 at 4, 5.3 -> 7.7 at 8 "
                         "(tools/shard_parts_probe.sh), which only pays once the exchange it hides is the larger part)")
    ap.add_argument("--stratified-per-item", type=float, default=0.0, help="--exchange stratified: updates per item per window inside a stratum (0 = 16 below 8 ranks, 32 from 8 ranks)")
    ap.add_argument("--contrib", choices=["auto", "fp32", "bf16"], default="auto",
                    help="window-minibatch step: storage format of the contribution rows (amd:contrib; sums stay fp32).  auto = bf16 on N > 1 ranks (a contribution is "
                         "written once and read once: half the bytes; the 3-seed contract is unchanged to 1e-9, profiles/r04_contract_seeds.txt), fp32 otherwise")
    ap.add_argument("--chunks", type=int, default=0,
                    help="--exchange stratified: file-order chunks per pass (a chunk = N sub-epochs; more chunks keep the training order closer to "
                         "the file order: tools/stratified_calibration.py, tools/contract_seeds.py); 0 = 8 below 8 ranks, 4 from 8 ranks")
    ap.add_argument("--blocks-per-rank", type=int, default=2,
                    help="--exchange stratified: item blocks per rank (1: a block is handed over between two steps; 2: the hand-over of a block "
                         "runs beside the training of the rank's next block)")
    ap.add_argument("--exchange", choices=["auto", "minibatch", "levels", "stratified"], default="auto",
                    help="N>1 (or --force-exchange): how the ranks train and exchange.  auto = stratified for ratings on N > 1 ranks, minibatch "
                         "otherwise.  minibatch: the window-minibatch step, user side "
                         "exact, item side one minibatch step per window, three launches per window (svdf_k_window.hip), all-reduce per window; levels: the round-2 scheme, "
                         "exact conflict-free levels per rank with the item side stale across ranks only; stratified: no all-reduce -- item blocks are "
                         "owned exclusively and handed from rank to rank (DSGD-style strata, window-minibatch step inside a stratum; ratings only)")
    ap.add_argument("--exchange-transport", choices=["rccl", "ipc", "native"], default="rccl",
                    help="N>1: what carries the exchange between the processes.  rccl: torch.distributed collectives / point-to-point (default until a hardware "
                         "run says otherwise); ipc: every rank's wire buffer and flag page IPC-mapped into every process, peer-pointer reduce-scatter + "
                         "all-gather kernel and inbox stores ordered by sequence flags in device memory (svdf_ipc.cpp, DESIGN.md section 6i)")
    ap.add_argument("--no-sequential-reference", action="store_true",
                    help="N>1: skip the exact single-GPU run of the same passes on rank 0 that rmse_sequential_reference comes from")
    ap.add_argument("--delta-dtype", choices=["fp16", "fp32"], default="fp16",
                    help="wire format of the item-side window deltas when --gpus > 1 (parameters stay fp32)")
    ap.add_argument("--cpu-sample", type=int, default=20_000_000, help="basicMF ratings of the CPU baseline sample (other workloads scale it down)")
    ap.add_argument("--groups-per-wave", type=int, default=0)
    ap.add_argument("--knob", action="append", default=[], help="extra tuning knob name=value (svdf_set_knob), repeatable")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pmc", choices=["auto", "all", "off"], default="auto", nargs="?", const="all",
                    help="N = 1: measure roofline.traffic in this run with two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over one pass of the workload in "
                         "a child process.  auto (default): the MAIN workload, capped at --pmc-cap seconds (a pass that does not fit quotes the committed "
                         "profiles/hbm_traffic.json and says so); all: the secondary workloads too, no cap; off: quote the committed file")
    ap.add_argument("--pmc-cap", type=float, default=150.0, help="--pmc auto: seconds the in-run counter passes may take in total")
    ap.add_argument("--no-auto-schedule", action="store_true", help="N>1, --exchange auto: keep the stratified ring as the main line whatever the preflight measured")
    ap.add_argument("--no-preflight", action="store_true", help="N>1: skip the checked all_reduce / ring hand-over before the run")
    ap.add_argument("--preflight-timeout", type=float, default=150.0, help="N>1: watchdog of rendezvous + preflight (seconds); on expiry the rank re-executes one ladder rung lower")
    ap.add_argument("--run-timeout", type=float, default=900.0, help="N>1: watchdog of the main workload (seconds)")
    ap.add_argument("--secondary-timeout", type=float, default=420.0, help="N>1: watchdog of each secondary (seconds); on expiry the contract line is printed without it")
    ap.add_argument("--data-seed", type=int, default=0, help="added to the generator seed of every synthetic stream (0 = the streams of SURVEY 8d2 / earlier rounds)")
    ap.add_argument("--no-orders", action="store_true", help="N=1: skip secondary.orders (Zipf items, generator-order pairs; benchlib/orders.py)")
    ap.add_argument("--no-window-step", action="store_true", help="N=1: skip the opt-in window-minibatch lines of the SVD++ / neighbourhood secondaries")
    ap.add_argument("--step-window", type=int, default=0, help="window-step secondaries: rows per window (amd:window); 0 = the engine's choice from the data")
    ap.add_argument("--step-per-target", type=int, default=0, help="window-step secondaries: updates a shared row meets per window (knob window_per_target); 0 = default")
    ap.add_argument("--multi-secondary", choices=["allreduce", "all", "none"], default="allreduce",
                    help="N>1, ratings: what runs besides the main line in the same command.  allreduce (default since round 6): the OTHER schedule on the same data -- "
                         "north_star's all-reduce window step when the ring is the main line, the ring otherwise -- so that allreduce_step is always measured; "
                         "all: also the native-RCCL / IPC transports of both steps and the single-process amd:gpus handle (five more runs, each under "
                         "--secondary-timeout: none of them has ever run on more than one device, and an 8-GPU driver run must not spend its wall clock on them); none = --no-multi-secondary")
    ap.add_argument("--no-multi-secondary", action="store_true",
                    help="N>1, ratings: skip secondary.allreduce_minibatch (the RCCL all-reduce step on the same data) and secondary.single_process_handle (rank 0 alone "
                         "drives all N devices through one amd:gpus handle, amd:exchange = p2p and rccl)")
    ap.add_argument("--secondary", default="auto", help="comma list of secondary workloads for the N=1 line (auto = all three with the default main line, none otherwise)")
    ap.add_argument("--secondary-steps", type=int, default=2)
    ap.add_argument("--defer-tails", type=float, default=0.05,
                    help="N>1: batches smaller than this fraction of a window's largest, at the end of the window, move to the next window (0 = off)")
    ap.add_argument("--use-graph", type=int, default=0, help="replay each resident dataset's pass as a captured hipGraph (0 = plain launches)")
    ap.add_argument("--force-exchange", action="store_true",
                    help="debug: run the item-delta exchange path even with one rank (exercises the N>1 code on one GPU)")
    a = ap.parse_args()
    if not a.factor and a.workload == "basicmf":
        a.factor = 64

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as plain `python bench.py --gpus N`: become the launcher of our own N ranks (one process per GPU, RCCL)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        os.execv(sys.executable, cmd)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    attempt = int(os.environ.get(ATTEMPT_ENV, "0"))
    if world != a.gpus:
        print("[bench] WORLD_SIZE=%d overrides --gpus %d" % (world, a.gpus), file=sys.stderr)
        a.gpus = world

    def log(msg):
        if rank == 0:
            print("[bench] " + msg, file=sys.stderr, flush=True)

    metric = {"basicmf": "training instances/sec (SGD updates/s), basicMF k=%d" % (a.factor or 64),
              "pairwise": "training pairs/sec (SGD updates/s), pairwiseRank k=%d" % (a.factor or 128),
              "svdpp": "training instances/sec (SGD updates/s), SVD++ implicit feedback k=%d" % (a.factor or 128),
              "neighbourhood": "training instances/sec (SGD updates/s), neighbourhood model k=%d" % (a.factor or 128)}[a.workload]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this driver
    a.pmc_results = {}
    if a.pmc != "off" and world == 1 and not a.force_exchange:
        # before this process opens the device, so that the counted child is alone on the GPU (the same commands as tools/profile_round.sh).
        # auto: the main workload only, under a 150 s cap (a pass that does not fit falls back to the committed profiles/hbm_traffic.json);
        # the children read this process's synthetic stream from /dev/shm instead of drawing it again
        import shutil, tempfile
        sec0 = a.secondary
        if sec0 == "auto":
            sec0 = "pairwise,svdpp,neighbourhood" if (a.workload == "basicmf" and a.ratings == 100_000_000) else ""
        names = [a.workload] + ([x for x in sec0.split(",") if x and x != a.workload] if a.pmc == "all" else [])
        cache = tempfile.mkdtemp(prefix="svdf_bench_data_", dir="/dev/shm" if os.path.isdir("/dev/shm") else "/tmp")
        try:
            os.environ["SVDF_BENCH_DATA_CACHE"] = cache
            if a.workload == "basicmf":
                os.environ["SVDF_BENCH_DATA_CACHE_WRITE"] = "1"
                cached(synth_triples, a.ratings + 1_000_000, a.users, a.items)
                os.environ.pop("SVDF_BENCH_DATA_CACHE_WRITE", None)
            deadline = None if a.pmc == "all" else time.time() + a.pmc_cap
            for nm in names:
                if nm in PMC_KERNEL:
                    a.pmc_results[nm] = measure_traffic(nm, a, log, deadline)
        finally:
            os.environ.pop("SVDF_BENCH_DATA_CACHE", None)
            shutil.rmtree(cache, ignore_errors=True)
    import torch
    import svdfeature_amd as sa   # noqa: F401  (fails loudly when the HIP library is missing: there is no CPU fallback)
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU fallback)"
    # debug only (1-GPU boxes): SVDF_BENCH_SHARE_GPU=1 puts every rank on GPU 0 and exchanges through gloo, so the
    # whole N>1 flow (sharding, windows, exchange, timing, RMSE reduction) can be exercised without N GPUs
    share_gpu = os.environ.get("SVDF_BENCH_SHARE_GPU", "0") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist, store, backend, pf, wd = None, None, None, None, None
    schedule_choice = None

    def esc(reason):
        escalate(reason, rank, world, attempt, metric)
    if world > 1:
        # ---- the N > 1 ladder: rendezvous + preflight under a watchdog; a rung that fails or hangs is left through
        # os.execv, so ONE driver command always ends in a JSON line
        wd = Watchdog(log)
        requested = a.exchange
        if attempt >= 1 and a.exchange in ("auto", "stratified"):
            a.exchange = "minibatch"
        if attempt >= 1:
            log("ladder rung %d: %s (after: %s)" % (attempt, LADDER[attempt], json.dumps(fallback_log())))
        wd.arm(a.preflight_timeout, "rendezvous + preflight", esc)
        try:
            dist, store, backend = rendezvous(torch, rank, world, local_rank, attempt, share_gpu)
            ring = a.workload == "basicmf" and a.exchange in ("auto", "stratified")
            ok, err = True, None
            try:
                if not a.no_preflight:
                    if os.environ.get("SVDF_BENCH_TEST_FAIL_PREFLIGHT") == str(attempt) and rank == world - 1:
                        raise RuntimeError("injected preflight failure (test hook)")
                    if os.environ.get("SVDF_BENCH_TEST_HANG_PREFLIGHT") == str(attempt) and rank == world - 1:
                        time.sleep(10 ** 6)
                    pf = preflight(torch, dist, rank, world, torch.device("cuda", local_rank), ring, log)
            except Exception as e:
                ok, err = False, repr(e)
            store_agree(store, rank, world, "preflight", ok, seconds=a.preflight_timeout)
            if not ok:
                raise RuntimeError(err)
            # (gloo stages through the host -- its timings say nothing about xGMI; SVDF_BENCH_TEST_AUTO_SCHEDULE lets the one-GPU tests walk the path)
            if ring and pf and a.exchange == "auto" and (backend == "nccl" or os.environ.get("SVDF_BENCH_TEST_AUTO_SCHEDULE")) and not a.no_auto_schedule \
                    and "handoff_1.6MB_us" in pf:
                if rank == 0:
                    store.set("schedule_choice", json.dumps(choose_schedule(pf, world, a)))
                schedule_choice = json.loads(store.get("schedule_choice").decode())   # blocks until rank 0 has decided
                log("schedule: %s" % json.dumps(schedule_choice))
                if schedule_choice["pick"] == "minibatch":
                    a.exchange = "minibatch"
        except Exception as e:
            esc("rendezvous / preflight: %r" % (e,))
        wd.arm(a.run_timeout, "main workload", esc)
    elif a.force_exchange:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    env = {"torch": torch, "dist": dist, "rank": rank, "world": world, "local_rank": local_rank, "log": log, "store": store}

    try:
        main_res = run_workload(a.workload, a, env, a.steps, a.warmup, True)
    except Exception as e:
        if world > 1:
            import traceback
            traceback.print_exc()
            esc("main workload: %r" % (e,))
        raise
    out = None
    if rank == 0:
        m = main_res
        out = {
            "metric": metric, "value": m["value"], "unit": m["unit"], "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": m["ms_per_step"], "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": m["workload"], "order": m["order"],
                       "conflict_free_batches_per_pass": m["conflict_free_batches_per_pass"], "schedule_build_s": m["schedule_build_s"],
                       "parallelism": m["parallelism"]},
            "roofline": m["roofline"], "cpu_baseline": m["cpu_baseline"], "parity": m["parity"],
            "end_to_end": m["end_to_end"],
        }
        for k in ("rmse_test_after_run", "passes_before_rmse", "rmse_sequential_reference", "rmse_minus_sequential", "exchange", "phase_ms",
                  "per_rank_ms", "roofline_aggregate", "model_ms",
                  "pair_accuracy_test_after_run", "mean_margin_test_after_run", "dag_bound", "launch_model", "init_model", "enqueue_ms_per_pass", "host_enqueue"):
            if m.get(k) is not None:
                out[k] = m[k]
        if world > 1:
            out["exchange"] = dict(out.get("exchange") or {}, backend=backend, ladder_rung=attempt, ladder=LADDER[attempt],
                                   fallback=fallback_log() or None, preflight_us=pf, schedule_choice=schedule_choice)

    def emit(extra=None):
        if rank == 0:
            if extra:
                out.update(extra)
            # the full object (any size) -> bench_secondary.json + a short table on stderr; stdout gets the contract head with one compact
            # tuple per secondary, asserted <= 6 000 bytes (benchlib/contract.py: the driver keeps the last 8 000 bytes of stdout)
            from benchlib.contract import compact_line, stderr_table
            side = os.environ.get("SVDF_BENCH_SIDE_FILE", os.path.join(ROOT, "bench_secondary.json"))
            try:
                with open(side, "w") as f:
                    json.dump(out, f, indent=1)
                out["details"] = os.path.basename(side)
            except OSError as e:
                out["details"] = "not written: %r" % (e,)
            table = stderr_table(out)
            if table:
                print("[bench] secondaries (full objects: %s)\n%s" % (out["details"], table), file=sys.stderr, flush=True)
            line, shed = compact_line(out)
            if shed:
                print("[bench] contract line: shed to fit: %s" % shed, file=sys.stderr, flush=True)
            sys.stdout.flush()
            C.CDLL(None).fflush(None)   # RCCL prints its version banner through C stdio: push it out BEFORE the JSON line, which stays the last line
            print(line, flush=True)

    def finish_now(reason):
        """watchdog action of the secondaries: the contract line goes out with what is there, every rank leaves without another collective"""
        emit({"secondary_error": reason})
        sys.stderr.flush()
        os._exit(0)

    secondary = {}
    if out is not None:
        out["secondary"] = secondary
    sec = a.secondary
    if sec == "auto":
        sec = "pairwise,svdpp,neighbourhood" if (a.workload == "basicmf" and world == 1 and a.ratings == 100_000_000) else ""
    if wd is not None:
        wd.arm(a.secondary_timeout, "secondary workloads", finish_now)
    for name in [s for s in sec.split(",") if s]:
        if name == a.workload or (world > 1 and name == "neighbourhood"):
            continue
        t0 = time.time()
        # SVD++ at 1 M users: an exact pass is 159 K levels x 46 us = 7.4 s whatever the kernel does (dag_bound): one timed pass after one warm-up
        r = run_workload(name, a, env, 1 if (name == "svdpp" and a.svdpp_users >= 200_000) else a.secondary_steps, 1, False)
        if r is not None:
            r["wall_s"] = round(time.time() - t0, 1)
            secondary["%s_k%d" % (name, WORKLOADS[name][2])] = r
    if rank == 0 and world == 1 and not a.no_window_step:
        for name in (["basicmf"] if (a.workload == "basicmf" and sec) else []) + [s for s in sec.split(",") if s in ("pairwise", "svdpp", "neighbourhood")]:
            try:   # extras: never lose the contract line over them
                exact = secondary.get("%s_k%d" % (name, WORKLOADS[name][2])) or {}
                big = name == "svdpp" and a.svdpp_users >= 200_000
                seq = (exact["passes_before_rmse"], exact["rmse_test_after_run"]) if (big and "passes_before_rmse" in exact) else None
                secondary["%s_k%d_window_step" % (name, WORKLOADS[name][2])] = run_window_step(sa, name, a, local_rank, log, steps=1 if big else 3, seq_quality=seq)
            except Exception as e:
                secondary["%s_window_step_error" % name] = repr(e)
    if rank == 0 and world == 1 and a.secondary == "auto" and secondary:
        try:
            secondary.update(run_f3_secondary(a, env))
        except Exception as e:   # the f3 rows are extras: never lose the contract line over them
            secondary["f3_error"] = repr(e)
    if rank == 0 and world == 1 and a.secondary == "auto" and secondary and not a.no_orders:
        try:   # the BASELINE workloads on the reference's own data orders, exact / window / auto (benchlib/orders.py)
            import types
            from benchlib import orders
            ctx = types.SimpleNamespace(Planted=Planted, HipEvents=HipEvents, rmse=rmse, make_trainer=make_trainer, WORKLOADS=WORKLOADS,
                                        HBM_PEAK_GBS=HBM_PEAK_GBS, cpu_baseline_and_parity=cpu_baseline_and_parity, log=log)
            secondary["orders"] = orders.run_orders(ctx, sa, a, local_rank)
        except Exception as e:
            secondary["orders_error"] = repr(e)

    # ---- N > 1, ratings: the other exchange designs on the SAME data in the SAME driver command
    if a.no_multi_secondary:
        a.multi_secondary = "none"
    if world > 1 and a.workload == "basicmf" and a.multi_secondary != "none":
        import argparse as _ap
        main_step = (main_res or {}).get("exchange", {}).get("step") if rank == 0 else None
        # (1) north_star's step: RCCL all-reduce of the per-item sums every window (window-minibatch step), when the main line was the ring
        # (when the preflight's timings made the all-reduce step the main line, the ring over torch.distributed is measured here instead)
        other = ("allreduce_minibatch", "minibatch") if a.exchange in ("auto", "stratified") else \
                (("stratified_ring", "stratified") if (schedule_choice or {}).get("pick") == "minibatch" else None)
        if other is not None:
            wd.arm(a.secondary_timeout, "secondary: %s" % other[0], finish_now)
            a2 = _ap.Namespace(**vars(a))
            a2.exchange, a2.no_cpu_baseline = other[1], True
            t0 = time.time()
            try:
                r = run_workload("basicmf", a2, env, 3, 1, False)
            except Exception as e:
                import traceback
                traceback.print_exc()
                finish_now("%s on rank %d: %r" % (other[0], rank, e))
            if r is not None:
                keep = ("value", "unit", "ms_per_step", "order", "exchange", "phase_ms", "per_rank_ms", "roofline", "roofline_aggregate", "model_ms",
                        "rmse_test_after_run", "passes_before_rmse", "rmse_sequential_reference", "rmse_minus_sequential")
                secondary[other[0]] = dict({k: r[k] for k in keep if r.get(k) is not None}, wall_s=round(time.time() - t0, 1), steps=3, warmup=1)
        # (1a) both steps with the rank's OWN RCCL communicator driven from C++ (svdf_rccl.cpp): the same RCCL, no Python / c10d call per collective --
        # what the host thread costs a pass shows in per_rank_ms.enqueue_* of this entry against the main line's; (1b) both steps again with the
        # DIRECT exchange between the processes (IPC-mapped buffers, no collective library on the data path)
        variants = [("allreduce_minibatch_ipc", "minibatch", "ipc"), ("stratified_ipc", "stratified", "ipc")]
        if backend == "nccl" or os.environ.get("SVDF_BENCH_TEST_NATIVE_ON_GLOO"):   # RCCL refuses two ranks on one device: only with a device per rank (the hook: tests of the skip path)
            variants = [("stratified_native", "stratified", "native"), ("allreduce_minibatch_native", "minibatch", "native")] + variants
        if a.multi_secondary != "all":
            variants = []
        for key, exch, transport in variants:
            wd.arm(a.secondary_timeout, "secondary: %s" % key, finish_now)
            a3 = _ap.Namespace(**vars(a))
            a3.exchange, a3.no_cpu_baseline, a3.exchange_transport = exch, True, transport
            t0 = time.time()
            try:
                r = run_workload("basicmf", a3, env, 3, 1, False)
            except TransportUnavailable as e:   # every rank is here: the next secondary can run
                log("secondary %s skipped: %s" % (key, e))
                secondary[key] = {"error": str(e)[:300]}
                continue
            except Exception as e:
                import traceback
                traceback.print_exc()
                finish_now("%s on rank %d: %r" % (key, rank, e))
            if r is not None:
                keep = ("value", "unit", "ms_per_step", "order", "exchange", "phase_ms", "per_rank_ms", "roofline_aggregate", "model_ms",
                        "rmse_test_after_run", "passes_before_rmse", "rmse_sequential_reference", "rmse_minus_sequential")
                secondary[key] = dict({k: r[k] for k in keep if r.get(k) is not None}, wall_s=round(time.time() - t0, 1), steps=3, warmup=1)
        # (2) the same algorithm behind ONE C-ABI handle: rank 0 alone drives all N devices from C++ (svdf_multi.cpp), direct peer exchange
        # and RCCL from the engine; the other ranks idle on the host (their GPUs hold no running kernel) until rank 0 says so through the store
        wd.arm(a.secondary_timeout, "secondary: single-process amd:gpus handle", finish_now)
        if a.multi_secondary != "all":
            pass
        elif rank == 0:
            sp = {}
            for xch in ("p2p", "rccl"):
                try:
                    sp[xch] = run_single_process_handle(sa, a, world, xch, 0 if share_gpu else local_rank, log)
                except Exception as e:
                    sp[xch] = {"error": repr(e)[:600]}
                    log("single-process handle, amd:exchange = %s: %r" % (xch, e))
            secondary["single_process_handle"] = sp
            store.set("single_process_done", "1")
        else:
            import datetime
            try:
                store.wait(["single_process_done"], datetime.timedelta(seconds=a.secondary_timeout + 30))
            except Exception:
                pass
    if out is not None and world > 1 and a.workload == "basicmf":
        # north_star's step -- user shards + an RCCL all-reduce of the item-side sums every window -- at the TOP level of every N > 1 line, whichever
        # schedule the preflight made the main line (the first hardware run decides between them from these two entries)
        keep = ("value", "unit", "ms_per_step", "per_rank_ms", "roofline_aggregate", "phase_ms", "model_ms", "rmse_test_after_run", "rmse_sequential_reference",
                "rmse_minus_sequential", "passes_before_rmse")
        if (out.get("exchange") or {}).get("step") == "minibatch":
            src, where = out, "the main line of this run"
        else:
            src, where = secondary.get("allreduce_minibatch") or {}, "secondary.allreduce_minibatch of this run (3 passes after 1 warm-up, same data, same ranks)"
        ns = {k: src[k] for k in keep if src.get(k) is not None}
        if ns:
            xs = src.get("exchange") or {}
            ns.update({"step": "window-minibatch step + all-reduce (SUM) of the per-item sums every window", "measured_as": where, "n_gpus": world,
                       "world_size_reported": xs.get("world_size_reported"), "backend": xs.get("backend", backend), "windows": xs.get("windows"),
                       "bytes_per_window": xs.get("bytes_per_window"), "transport": xs.get("transport")})
            out["allreduce_step"] = ns
        else:
            out["allreduce_step"] = {"error": "not measured in this run (secondary skipped or failed: see secondary / secondary_error)"}
    if wd is not None:
        wd.disarm()
    if out is not None and not secondary:
        out.pop("secondary", None)
    emit()
    if dist is not None:
        if wd is not None:
            wd.arm(60, "shutdown", lambda reason: os._exit(0))
        try:
            dist.barrier()
            dist.destroy_process_group()
        except Exception:
            pass


if __name__ == "__main__":
    main()
