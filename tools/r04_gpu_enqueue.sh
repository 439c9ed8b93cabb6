#!/bin/bash
# host enqueue time per pass of one rank's share (rank-of-N settings on one GPU, exchange forced on with one rank): is the Python loop ahead of the GPU?
cd "${GRAFT_REPO_ROOT:-/root/repo}"
show='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(sys.argv[1], "ms/pass %.2f" % d["ms_per_step"], "enqueue ms/pass %.2f" % d["enqueue_ms_per_pass"], "launches/pass", d["roofline"]["launches"]//d["steps"])'
for n in 2 8; do
  python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-sequential-reference --pmc off --force-exchange --exchange minibatch --windows 32 --ratings $((100000000/n)) --users $((1000000/n)) 2>/dev/null | python -c "$show" "rank-of-$n all-reduce step (32 windows)"
  c=8; [ $n -eq 8 ] && c=4
  python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-sequential-reference --pmc off --force-exchange --exchange stratified --chunks $((c*2*n)) --ratings $((100000000/n)) --users $((1000000/n)) --items $((100000/(2*n))) 2>/dev/null | python -c "$show" "rank-of-$n stratified ($c chunks x $((2*n)) steps)"
done 2>&1 | tee gpurun_out/enqueue_probe.txt
