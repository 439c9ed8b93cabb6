#!/bin/bash
# round 3, GPU call E: stratified schedule -- parity tests, the N = 2 / 3 flow through gloo with every rank on GPU 0, one rank's share at N = 2 / 4 / 8
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r03e
mkdir -p $OUT
timeout 600 python -X faulthandler -m pytest tests/test_gpu_window.py -x -q > $OUT/test_window.log 2>&1; echo "test_gpu_window rc=$?"; tail -3 $OUT/test_window.log
show='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); p=d.get("phase_ms") or {}; print(sys.argv[1], "ms/pass %.2f" % d["ms_per_step"], "launches/pass", d["roofline"]["launches"]//d["steps"], {k: round(v,3) for k,v in p.items() if k!="what"}, "rmse", d.get("rmse_test_after_run"), "seq", d.get("rmse_sequential_reference"), (d.get("exchange") or {}).get("path"))'
export SVDF_BENCH_SHARE_GPU=1
for ex in stratified minibatch; do
  timeout 600 python bench.py --gpus 2 --ratings 10000000 --no-cpu-baseline --steps 2 --exchange $ex 2> $OUT/share2_$ex.log | python -c "$show" "2 ranks on one GPU (gloo) $ex" | tee -a $OUT/share.txt
done
timeout 600 python bench.py --gpus 3 --ratings 10000000 --no-cpu-baseline --steps 2 --exchange stratified 2> $OUT/share3_stratified.log | python -c "$show" "3 ranks on one GPU (gloo) stratified" | tee -a $OUT/share.txt
unset SVDF_BENCH_SHARE_GPU
# one rank's compute share: stratum windows of an N-rank run = (ratings / N, users / N, items / N per block): emulate with items/N so that
# a window's instances fall on one block's worth of items -- 32 window steps per pass like the real schedule
for n in 8 4 2; do
  timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sequential-reference --force-exchange --exchange stratified --chunks 32 --ratings $((100000000/n)) --users $((1000000/n)) --items $((100000/n)) 2>$OUT/probe_$n.log | python -c "$show" "stratified rank-of-$n (32 window steps, items/N per block)" | tee -a $OUT/probe.txt
done
