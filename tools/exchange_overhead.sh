#!/bin/bash
# Cost of the item-delta exchange path on ONE GPU (the collective is the identity with one rank): bench.py with
# --force-exchange at several window counts next to the plain run.  Usage: tools/exchange_overhead.sh [dtype]
dt=${1:-fp16}
show='import sys,json; d=json.loads(sys.stdin.readline()); print(sys.argv[1], "ms/pass %.2f" % d["ms_per_step"], "rmse %.6f" % d["rmse_test_after_run"])'
python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "$show" "plain"
for w in 16 32 64; do
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --force-exchange --windows $w --delta-dtype $dt 2>/dev/null | python -c "$show" "windows=$w($dt)"
done
