#!/bin/bash
# round 4, call I: rank pairs (BASELINE configs[4], 200 M pairs, k = 128) at the DEMO learning rate (demo/pairwiseRank/pairwiseRank.conf: 0.005,
# what bench.py's pairwise workload uses): held-out pair accuracy / mean margin of the window-minibatch step at 125 / 32 / 12 / 6 / 3 windows per
# pass next to the exact pass of the same epochs (one rank plays all ranks: the step's result does not depend on their number)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r04i
mkdir -p $OUT
show='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(sys.argv[1], "ms/pass %.1f" % d["ms_per_step"], "accuracy", d.get("pair_accuracy_test_after_run"), "margin", d.get("mean_margin_test_after_run"), "windows", (d.get("exchange") or {}).get("windows"))'
timeout 900 python bench.py --workload pairwise --steps 3 --warmup 1 --no-cpu-baseline --pmc off --secondary "" 2> $OUT/exact.log | python -c "$show" "exact" | tee -a $OUT/pairs.txt
for w in 125 32 12 6 3; do
  timeout 900 python bench.py --workload pairwise --steps 2 --warmup 1 --no-cpu-baseline --pmc off --secondary "" --force-exchange --windows $w --delta-dtype fp16 2> $OUT/w$w.log | python -c "$show" "window step, $w windows" | tee -a $OUT/pairs.txt
done
