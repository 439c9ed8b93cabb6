#!/bin/bash
# round 3, GPU call R: the accuracy contract after 10 epochs at the FULL configs[2] size: 8 and 2 ranks sharing GPU 0 (gloo), stratified schedule
# and (8 ranks) the all-reduce window-minibatch step, each next to the exact sequential run of the same passes
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r03r
mkdir -p $OUT
export SVDF_BENCH_SHARE_GPU=1
show='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(sys.argv[1], "ms/pass %.1f" % d["ms_per_step"], "rmse", d.get("rmse_test_after_run"), "seq", d.get("rmse_sequential_reference"), "d", d.get("rmse_minus_sequential"), "passes", d.get("passes_before_rmse"))'
timeout 1700 python bench.py --gpus 8 --no-cpu-baseline --steps 9 --warmup 0 --exchange stratified 2> $OUT/s8.log | python -c "$show" "full size, 8 ranks, stratified" | tee -a $OUT/full10.txt
timeout 1700 python bench.py --gpus 2 --no-cpu-baseline --steps 9 --warmup 0 --exchange stratified 2> $OUT/s2.log | python -c "$show" "full size, 2 ranks, stratified" | tee -a $OUT/full10.txt
timeout 1700 python bench.py --gpus 8 --no-cpu-baseline --steps 9 --warmup 0 --exchange minibatch 2> $OUT/m8.log | python -c "$show" "full size, 8 ranks, window-minibatch all-reduce" | tee -a $OUT/full10.txt
