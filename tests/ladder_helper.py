"""Helper of tests/test_bench_ladder.py: one rank of bench.py's N > 1 ladder on the CPU (gloo).  Plays rendezvous -> agreement -> escalate
exactly as bench.main() does; MODE says what goes wrong on the last rank at attempt 0: "fail" (an exception before the agreement),
"hang" (never answers: the watchdog has to fire), "ok"."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    import torch
    mode = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    attempt = int(os.environ.get(bench.ATTEMPT_ENV, "0"))
    wd = bench.Watchdog(lambda m: None)

    def esc(reason):
        bench.escalate(reason, rank, world, attempt, "test metric")
    wd.arm(8.0, "rendezvous + preflight", esc)
    try:
        dist, store, backend = bench.rendezvous(torch, rank, world, 0, attempt, True)
        ok = True
        if attempt == 0 and rank == world - 1:
            if mode == "fail":
                ok = False
            elif mode == "hang":
                time.sleep(10 ** 6)
        x = torch.full((8,), float(rank + 1))
        if ok:
            dist.all_reduce(x) if (mode == "ok" or attempt > 0) else None
        bench.store_agree(store, rank, world, "preflight", ok, seconds=6)
    except Exception as e:
        esc("rendezvous / preflight: %r" % (e,))
    wd.disarm()
    t = torch.full((4,), float(rank + 1))
    dist.all_reduce(t)
    if rank == 0:
        print(json.dumps({"rung": attempt, "ladder": bench.LADDER[attempt], "fallback": bench.fallback_log(), "sum": float(t[0])}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
