"""bench.py --gpus N: what one driver command needs around the measured passes -- the watchdog, the rendezvous ladder
(every rung a fresh process image of the same rank), the checked preflight of the transports, the agreement helpers, the schedule choice from
the preflight numbers and the pass model the line is compared with.  No measurement happens here."""
import json
import os
import sys
import time

import numpy as np

ATTEMPT_ENV, FALLBACK_ENV = "SVDF_BENCH_ATTEMPT", "SVDF_BENCH_FALLBACK_LOG"
# the ladder of one driver command: every rung is a fresh process image of THIS rank (os.execv keeps the pid, so the
# torch.distributed.run agent sees nothing) that meets the others again under a new key prefix of the same TCP store
LADDER = ["as requested", "exchange = minibatch (RCCL all-reduce only, no point-to-point ring)", "backend = gloo (host-staged exchange), exchange = minibatch"]


class Watchdog:
    """One deadline at a time, watched by a daemon thread: when the armed phase does not finish in time the thread dumps every Python
    stack (faulthandler) and runs the phase's action -- which never returns (os.execv to the next ladder rung, or the JSON line + os._exit).
    A hung RCCL call blocks the main thread inside C; this thread does not need it."""

    def __init__(self, log):
        import threading
        self.log, self.deadline, self.what, self.action = log, None, None, None
        self.cv = threading.Condition()
        th = threading.Thread(target=self._run, daemon=True)
        th.start()

    def arm(self, seconds, what, action):
        with self.cv:
            self.deadline, self.what, self.action = time.time() + seconds, what, action
            self.cv.notify()

    def disarm(self):
        with self.cv:
            self.deadline = None
            self.cv.notify()

    def _run(self):
        import faulthandler
        while True:
            with self.cv:
                if self.deadline is None:
                    self.cv.wait()
                    continue
                left = self.deadline - time.time()
                if left > 0:
                    self.cv.wait(left)
                    continue
                what, action = self.what, self.action
                self.deadline = None
            print("[bench] WATCHDOG: '%s' did not finish in time; stacks follow" % what, file=sys.stderr, flush=True)
            try:
                faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
            except Exception:
                pass
            action("watchdog: '%s' timed out" % what)


def fallback_log():
    try:
        return json.loads(os.environ.get(FALLBACK_ENV, "[]"))
    except Exception:
        return []


def escalate(reason, rank, world, attempt, metric="training instances/sec (SGD updates/s), basicMF k=64"):
    """This rank gives up the current ladder rung: re-execute bench.py one rung lower (the other ranks follow through their own
    watchdogs / store timeouts), or, below the last rung, print the contract line with value 0 and the reasons -- a record is never lost."""
    reasons = fallback_log() + [{"attempt": attempt, "rung": LADDER[min(attempt, len(LADDER) - 1)], "rank": rank, "reason": str(reason)[:400]}]
    print("[bench] rank %d attempt %d failed: %s" % (rank, attempt, reason), file=sys.stderr, flush=True)
    if attempt + 1 < len(LADDER):
        os.environ[ATTEMPT_ENV] = str(attempt + 1)
        os.environ[FALLBACK_ENV] = json.dumps(reasons)
        sys.stdout.flush(); sys.stderr.flush()
        os.execv(sys.executable, [sys.executable] + sys.argv)
    if rank == 0:
        from benchlib.contract import compact_line
        line, _ = compact_line({"metric": metric, "value": 0.0, "unit": "instances/s", "n_gpus": world, "steps": 0, "warmup": 0, "ms_per_step": None,
                                "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                                "config": {"workload": "FAILED: no exchange path worked on this node"}, "roofline": None, "cpu_baseline": None,
                                "parity": None, "exchange": {"fallback": reasons}})
        print(line, flush=True)
    os._exit(3)


def rendezvous(torch, rank, world, local_rank, attempt, share_gpu):
    """(dist, key-value store, backend): the process group of this ladder rung.  Under torch.distributed.run the agent hosts the TCP store
    (workers are clients), so a re-executed rank can meet the others again: every rung uses its own key prefix."""
    import datetime
    import torch.distributed as dist
    host, port = os.environ.setdefault("MASTER_ADDR", "127.0.0.1"), int(os.environ.setdefault("MASTER_PORT", "29533"))
    agent = os.environ.get("TORCHELASTIC_USE_AGENT_STORE") == "True"
    if not agent:
        port += attempt   # rank 0 hosts the store itself: a fresh port per rung
    tcp = dist.TCPStore(host, port, world, is_master=(not agent and rank == 0), timeout=datetime.timedelta(seconds=300), wait_for_workers=False)
    store = dist.PrefixStore("svdf_bench_try%d" % attempt, tcp)
    backend = "gloo" if (share_gpu or attempt >= 2) else "nccl"
    kw = {} if backend == "gloo" else {"device_id": torch.device("cuda", local_rank)}
    dist.init_process_group(backend, store=dist.PrefixStore("pg", store), rank=rank, world_size=world, timeout=datetime.timedelta(seconds=600), **kw)
    return dist, store, backend


def store_agree(store, rank, world, key, ok, seconds=120):
    """every rank publishes ok / not ok under `key` and reads everybody's: host-side only (no collective), so it also works when the
    collective under test is what is broken.  Raises when a rank said no or did not answer."""
    import datetime
    store.set("%s/%d" % (key, rank), "1" if ok else "0")
    store.wait(["%s/%d" % (key, r) for r in range(world)], datetime.timedelta(seconds=seconds))
    bad = [r for r in range(world) if store.get("%s/%d" % (key, r)) != b"1"]
    if bad:
        raise RuntimeError("%s failed on ranks %s" % (key, bad))


def preflight(torch, dist, rank, world, device, ring, log):
    """Before any training: the collectives the run will use, on small and on run-sized buffers, results CHECKED.
    (1) all_reduce of 1 K floats and of a window's wire buffer (13 MB fp16 at configs[2]);
    (2) when the stratified ring is the plan: batch_isend_irecv rank r -> r - 1 exactly as HipShard.handoff_start / handoff_wait issue it
        (inside a side stream's context, wait() = stream wait), 1 K floats and one item block (1.6 MB).
    Returns the measured times (they sit next to the model numbers in the JSON line)."""
    out = {}
    stream = torch.cuda.Stream(device=device)

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e6
    with torch.cuda.stream(stream):
        small = torch.full((1024,), float(rank + 1), device=device)
        dist.all_reduce(small)
        stream.synchronize()
        want = world * (world + 1) / 2.0
        if not bool((small == want).all().item()):
            raise RuntimeError("preflight all_reduce: wrong sum %r (want %r)" % (float(small[0].item()), want))
        big = torch.zeros(13 * 1024 * 1024 // 2, device=device, dtype=torch.float16)
        out["allreduce_13MB_fp16_us"] = timed(lambda: dist.all_reduce(big), 10)
        out["allreduce_4KB_us"] = timed(lambda: dist.all_reduce(small), 20)
    if ring:
        dst, src = (rank - 1) % world, (rank + 1) % world
        for nfl, key, reps in ((1024, "handoff_4KB_us", 20), (400 * 1024, "handoff_1.6MB_us", 10)):
            snd = torch.full((nfl,), float(rank), device=device)
            rcv = torch.full((nfl,), -1.0, device=device)

            def once():
                with torch.cuda.stream(stream):
                    reqs = dist.batch_isend_irecv([dist.P2POp(dist.isend, snd, dst), dist.P2POp(dist.irecv, rcv, src)])
                    for q in reqs:
                        q.wait()
            once()
            stream.synchronize()
            if not bool((rcv == float(src)).all().item()):
                raise RuntimeError("preflight ring hand-over: rank %d received %r from rank %d" % (rank, float(rcv[0].item()), src))
            out[key] = timed(once, reps)
    log("preflight ok: %s" % json.dumps({k: round(v, 1) for k, v in out.items()}))
    return out


class TransportUnavailable(RuntimeError):
    """a secondary's transport could not be opened on some rank; EVERY rank knows (agreed through the store), so the secondary is skipped, not fatal"""


_open_counter = [0]


def open_transport_agreed(env, what, open_fn):
    """opens an optional exchange transport (native RCCL communicator, IPC mapping) on every rank and makes the ranks agree on the outcome through the
    store before anyone trains through it: a rank that could not open it takes the others out of the secondary together.  (A rank that HANGS inside
    the open is the watchdog's business.)"""
    ok, err = True, None
    try:
        open_fn()
    except Exception as e:   # noqa: BLE001
        ok, err = False, repr(e)
    store = env.get("store")
    if store is not None and env["world"] > 1:
        _open_counter[0] += 1
        try:
            store_agree(store, env["rank"], env["world"], "open_%s_%d" % (what, _open_counter[0]), ok, seconds=120)
        except Exception as e:
            raise TransportUnavailable("%s: %s" % (what, err or e))
    elif not ok:
        raise TransportUnavailable("%s: %s" % (what, err))


def choose_schedule(pf, world, a):
    """--exchange auto on real devices (backend nccl): the stratified ring or the all-reduce step as the MAIN line, from what the preflight just measured
    on this node.  Both keep the accuracy contract; which one is faster depends on what a hand-over of one item block costs the rank that
    issues it (host + link, not hidden behind a ~40 us step at 8 ranks: DESIGN.md 6j) against what an all-reduce of a window's sums costs.
    est(stratified) = max(compute share, hand-overs x measured hand-over); est(all-reduce step) = compute share + windows x measured all-reduce.
    The other schedule is still measured in the same command as a secondary."""
    n, items, factor = a.ratings, a.items, (a.factor or 64)
    t1 = 17.1 * n / 1e8 * factor / 64.0   # the exact one-GPU pass (round 5: runs; 23.5 ms in rounds 2-4)
    contract = n == 100_000_000 and items == 100_000 and factor == 64
    share_s = ({2: 9.78, 4: 5.92, 8: 2.71}.get(world) if contract else None) or t1 / world
    share_a = ({2: 8.26, 4: 4.26, 8: 2.66}.get(world) if contract else None) or t1 / world
    bpr = max(1, a.blocks_per_rank)
    chunks = a.chunks if a.chunks > 0 else (8 if world < 8 else 4)
    handoffs = chunks * world * bpr
    nwin = a.windows if a.windows > 0 else max(1, int(np.ceil(n / max(items, 1) / 32.0)))
    block_bytes = items * (factor + 1) * 4.0 / (world * bpr)
    ho_ms = pf["handoff_1.6MB_us"] * 1e-3 * max(block_bytes / (400 * 1024 * 4.0), 0.25)    # measured on a 1.6 MB block; small blocks keep the fixed part
    ar_ms = pf["allreduce_13MB_fp16_us"] * 1e-3 * max(items * (factor + 1) * 2.0 / (13 * 1024 * 1024.0), 0.25)
    est_s, est_a = max(share_s, handoffs * ho_ms), share_a + nwin * ar_ms
    return {"pick": "stratified" if est_s <= est_a else "minibatch", "est_stratified_ms": est_s, "est_allreduce_step_ms": est_a,
            "handoffs_per_pass": handoffs, "handoff_ms": ho_ms, "windows": nwin, "allreduce_ms": ar_ms,
            "what": "decided by rank 0 from the preflight's timings of this node; the other schedule is measured as a secondary of the same command"}


def model_ms(name, world, exchange_step, n, items, factor, nwin, blocks, handoffs, t1_ms):
    """What DESIGN.md sections 6c / 6d / 6f expect for this line, so that a hardware curve can be checked against the model from the line
    itself.  compute share = the one-GPU share table of 6c / 6f where the workload is the contract one (100 M ratings, k = 64), else T1 / N;
    all-reduce schemes: per-link-bound ring, 2 (N-1)/N x bytes / 153 GB/s + 30 us per window, not overlapped;
    stratified: `handoffs` point-to-point transfers per rank and pass of one item block (NI / blocks rows, fp32) over one xGMI link,
    bytes / 64 GB/s + 25 us each -- hidden behind the next step's training at 2 blocks per rank (lower figure), serial at 1 (upper)."""
    contract = name == "basicmf" and n == 100_000_000 and items == 100_000 and factor == 64
    table = {"minibatch": {2: 8.26, 4: 4.26, 8: 2.66}, "stratified": {2: 9.78, 4: 5.92, 8: 2.71}}   # round-4 defaults, profiles/r04_shard_scale_probe.txt
    share = table.get(exchange_step, {}).get(world) if contract else None
    src = "compute share: one rank's share measured on one GPU (DESIGN.md 6c / 6f)"
    if share is None:
        share, src = (t1_ms / world if t1_ms else None), "compute share: T1 / N with T1 = 17.1 ms per 100 M ratings (k = 64; the exact one-GPU pass of round 5) scaled by size and width"
    if share is None:
        return None
    if exchange_step == "stratified":
        per = items * (factor + 1) * 4.0 / max(blocks, 1) / 64e9 * 1e3 + 0.025
        lo, hi = share, share + handoffs * per
        return {"compute_share_ms": share, "handoff_ms_each": per, "handoffs_per_rank": handoffs, "total_ms": [lo, hi],
                "speedup_over_one_gpu": [t1_ms / hi, t1_ms / lo] if t1_ms else None, "source": src + "; hand-overs: DESIGN.md 6f"}
    t_ar = 2.0 * (world - 1) / world * items * (factor + 1) * 2.0 / 153e9 * 1e3 + 0.030
    total = share + nwin * t_ar
    return {"compute_share_ms": share, "allreduce_ms_per_window": t_ar, "exchange_ms": nwin * t_ar, "total_ms": total,
            "speedup_over_one_gpu": (t1_ms / total) if t1_ms else None, "source": src + "; ring all-reduce: DESIGN.md 6d"}
