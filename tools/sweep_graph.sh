#!/bin/bash
# hipGraph replay of a resident dataset's pass vs plain launches: contract workload, one rank's shard of an 8-GPU run
# (windowed, exchange forced), and the short-batch shapes.
show='import sys,json; d=json.loads(sys.stdin.readline()); print(sys.argv[1], "ms/pass %.2f" % d["ms_per_step"])'
for g in 0 1; do
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --use-graph $g 2>/dev/null | python -c "$show" "contract graph=$g"
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --force-exchange --windows 32 --ratings 12500000 --users 125000 --use-graph $g 2>/dev/null | python -c "$show" "rank-of-8 graph=$g"
  python tools/bench_variants.py --which pairwise,neighbor,svdpp --n 8000000 --use-graph $g 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('graph=$g', d['case'], 'ms/pass %.2f' % d['ms_per_pass'], 'launches', d['launches_per_pass'])"
done
