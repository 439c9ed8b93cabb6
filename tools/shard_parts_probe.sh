#!/bin/bash
# one rank's compute share per pass (see shard_scale_probe.sh) with the window exchange in one piece or in two item-range pieces
cd ${GRAFT_REPO_ROOT:-/root/repo}
show='import sys,json; d=json.loads(sys.stdin.readline()); c=d["config"]; print(sys.argv[1], "ms/pass %.2f" % d["ms_per_step"], "batches/pass", c["conflict_free_batches_per_pass"])'
for n in 2 4 8; do
  w=16; [ $n -eq 4 ] && w=24; [ $n -eq 8 ] && w=32
  for p in 1 2; do
    python bench.py --steps 3 --warmup 1 --no-cpu-baseline --force-exchange --windows $w --exchange-parts $p --ratings $((100000000/n)) --users $((1000000/n)) 2>/dev/null | python -c "$show" "rank-of-$n windows=$w parts=$p"
  done
done
