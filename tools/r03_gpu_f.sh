#!/bin/bash
# round 3, GPU call F: the FULL configs[2] size on 8 / 4 ranks that share GPU 0 (gloo hand-overs: slow, the point is the RMSE next to
# the exact sequential run of the same passes), stratified schedule
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r03f
mkdir -p $OUT
export SVDF_BENCH_SHARE_GPU=1
show='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(sys.argv[1], "ms/pass %.1f" % d["ms_per_step"], "rmse", d.get("rmse_test_after_run"), "seq", d.get("rmse_sequential_reference"), "d", d.get("rmse_minus_sequential"), "passes", d.get("passes_before_rmse"))'
for n in 8 4; do
  timeout 1500 python bench.py --gpus $n --no-cpu-baseline --steps 2 --warmup 0 --exchange stratified 2> $OUT/full_$n.log | python -c "$show" "full size, $n ranks on one GPU, stratified" | tee -a $OUT/full.txt
done
