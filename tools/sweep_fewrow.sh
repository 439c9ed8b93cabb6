#!/bin/bash
# block-size / batch-order sweep of the specialised few-row kernel on the pairwise workload (50 M pairs)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() {
  python bench.py --workload pairwise --pairs 50000000 --no-cpu-baseline --secondary "" "$@" 2>/dev/null | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', round(d['ms_per_step'],3), 'ms', round(d['roofline']['avg_launch_us'],3), 'us/launch', round(d['roofline']['frac'],4))"
}
for b in 64 128 256; do run --knob block_threads=$b; done
for b in 64 256; do run --knob block_threads=$b --knob sort_batches=2; done
run --knob xcd_remap=0
