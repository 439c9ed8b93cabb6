#!/bin/bash
# What ONE rank of an N-GPU run does per pass, measured on one GPU without the collective's cost: the rank's shard
# (ratings/N, users/N, all items), the window count bench.py would pick, exchange forced on (one-rank all-reduce = identity).
# minibatch = the window-minibatch step (svdf_k_window.hip, 3 launches per window), levels = the round-2 scheme.
cd ${GRAFT_REPO_ROOT:-/root/repo}
show='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); c=d["config"]; p=d.get("phase_ms") or {}; print(sys.argv[1], "ms/pass %.2f" % d["ms_per_step"], "launches/pass", d["roofline"]["launches"]//d["steps"], "phase_ms", {k: round(v,3) for k,v in p.items() if k!="what"}, "rmse", d.get("rmse_test_after_run"))'
for n in ${RANKS:-2 4 8}; do
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sequential-reference --force-exchange --exchange minibatch --windows 32 --ratings $((100000000/n)) --users $((1000000/n)) $EXTRA 2>/dev/null | python -c "$show" "rank-of-$n minibatch (32 windows)"
  # stratified schedule, two item blocks per rank: a step's window lies on items / (2 N) items; 64 window steps per pass (32 with one block per rank)
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sequential-reference --force-exchange --exchange stratified --chunks 64 --ratings $((100000000/n)) --users $((1000000/n)) --items $((100000/(2*n))) 2>/dev/null | python -c "$show" "rank-of-$n stratified (64 steps, items/2N per block)"
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sequential-reference --force-exchange --exchange stratified --chunks 32 --ratings $((100000000/n)) --users $((1000000/n)) --items $((100000/n)) 2>/dev/null | python -c "$show" "rank-of-$n stratified (32 steps, items/N per block)"
  w=16; [ $n -eq 4 ] && w=24; [ $n -eq 8 ] && w=32
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sequential-reference --force-exchange --exchange levels --windows $w --ratings $((100000000/n)) --users $((1000000/n)) 2>/dev/null | python -c "$show" "rank-of-$n levels windows=$w tails-deferred"
done
