#!/bin/bash
# one rank's share of an 8-GPU (and 2-/4-GPU) basicMF run on one GPU (see shard_scale_probe.sh) under kernel-shape knobs
cd ${GRAFT_REPO_ROOT:-/root/repo}
show='import sys,json; d=json.loads(sys.stdin.readline()); c=d["config"]; print(sys.argv[1], "ms/pass %.2f" % d["ms_per_step"], "batches/pass", c["conflict_free_batches_per_pass"])'
for n in 8 4 2; do
  w=16; [ $n -ge 4 ] && w=32
  for cfg in "groups_per_wave=0" "groups_per_wave=4" "groups_per_wave=2" "groups_per_wave=1" "basic_i8=0"; do
    k=""; for x in $cfg; do k="$k --knob $x"; done
    python bench.py --steps 3 --warmup 1 --no-cpu-baseline --force-exchange --windows $w --ratings $((100000000/n)) --users $((1000000/n)) $k 2>/dev/null | python -c "$show" "rank-of-$n $cfg"
  done
done
