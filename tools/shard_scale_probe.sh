#!/bin/bash
# What ONE rank of an N-GPU run does per pass, measured on one GPU without the collective's cost: the rank's shard
# (ratings/N, users/N, all items), the window count bench.py would pick, exchange forced on (identity all-reduce).
show='import sys,json; d=json.loads(sys.stdin.readline()); c=d["config"]; print(sys.argv[1], "ms/pass %.2f" % d["ms_per_step"], "batches/pass", c["conflict_free_batches_per_pass"])'
for n in 2 4 8; do
  w=16; [ $n -ge 4 ] && w=32
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --force-exchange --windows $w --ratings $((100000000/n)) --users $((1000000/n)) --defer-tails 0 2>/dev/null | python -c "$show" "rank-of-$n windows=$w plain"
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --force-exchange --windows $w --ratings $((100000000/n)) --users $((1000000/n)) 2>/dev/null | python -c "$show" "rank-of-$n windows=$w tails-deferred"
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --ratings $((100000000/n)) --users $((1000000/n)) 2>/dev/null | python -c "$show" "rank-of-$n no-exchange"
done
