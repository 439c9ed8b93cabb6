#!/bin/bash
R=/root/repo
cd $R
run() { python $R/tests/perf_ranker.py --sections 1201 --cpu-sections 0 --top-k 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read())['gpu_bulk']; print(round(d['ms_per_section']*1e3,2), end=' ')"; }
echo "positions: us per section, 10 processes each"
echo "library pins malloc's mmap threshold (default):"; for i in 1 2 3 4 5 6 7 8 9 10; do run; done; echo
echo "SVDF_KEEP_MALLOC_DYNAMIC=1:"; for i in 1 2 3 4 5 6 7 8 9 10; do SVDF_KEEP_MALLOC_DYNAMIC=1 run; done; echo
