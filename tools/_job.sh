#!/bin/bash
for K in 1 2 0 1 2; do
  python bench.py --gpus 1 --steps 10 --warmup 3 --secondary '' --pmc off --no-cpu-baseline --no-orders --knob sort_batches=$K 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sort_batches=$K', d['ms_per_step'], d['roofline']['frac'], d['config']['conflict_free_batches_per_pass'])"
done
