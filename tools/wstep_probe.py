#!/usr/bin/env python3
"""Window-minibatch step on ONE GPU at the full BASELINE configs[3] sizes (bench.run_window_step): throughput and the accuracy contract
|dRMSE| <= 1e-4 against the exact pass, over data seeds and window sizes.  usage: wstep_probe.py svdpp|neighbourhood SEEDS PER_TARGETS [steps]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    import argparse
    import svdfeature_amd as sa
    name = sys.argv[1]
    seeds = [int(x) for x in sys.argv[2].split(",")]
    targets = [int(x) for x in sys.argv[3].split(",")]
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    a = argparse.Namespace(users=1_000_000, items=100_000, globals=10_000, svdpp_users=int(os.environ.get("WSTEP_USERS", "40000")), svdpp_per_user=100, neighbour_rows=4_000_000,
                           step_window=0, step_per_target=0, data_seed=0, contrib=os.environ.get("WSTEP_CONTRIB", "fp32"))
    log = lambda m: print("[probe] " + m, file=sys.stderr, flush=True)
    for seed in seeds:
        for pt in targets:
            a.data_seed, a.step_per_target = seed, pt   # (sets both knobs: window_per_target and window_per_target_fb)
            r = bench.run_window_step(sa, name, a, 0, log, steps=steps)
            print(json.dumps({"workload": name, "seed": seed, "per_target": pt, "windows": r["windows_per_pass"], "ms_per_pass": r["ms_per_step"],
                              "M_inst_s": r["value"] / 1e6, "frac": r["roofline"]["frac"], "rmse": r["rmse_test_after_run"],
                              "rmse_seq": r["rmse_sequential_reference"], "d": r["rmse_minus_sequential"], "build_s": r["build_s"]}), flush=True)
        bench._DATA_CACHE.clear()


if __name__ == "__main__":
    main()
