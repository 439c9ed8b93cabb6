#!/bin/bash
# round 3, GPU call G: defaults of bench.py --gpus N on ranks that share GPU 0 (gloo), probe of one rank's share incl. the stratified schedule, full GPU suite
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r03g
mkdir -p $OUT
show='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); p=d.get("phase_ms") or {}; print(sys.argv[1], "ms/pass %.2f" % d["ms_per_step"], {k: round(v,3) for k,v in p.items() if k!="what"}, "rmse", d.get("rmse_test_after_run"), "seq", d.get("rmse_sequential_reference"), (d.get("exchange") or {}).get("step"), (d.get("exchange") or {}).get("handoffs_per_pass"))'
export SVDF_BENCH_SHARE_GPU=1
timeout 900 python bench.py --gpus 2 --ratings 20000000 --no-cpu-baseline --steps 2 2> $OUT/share2.log | python -c "$show" "2 ranks on one GPU (gloo), defaults" | tee -a $OUT/share.txt
timeout 900 python bench.py --gpus 3 --ratings 20000000 --no-cpu-baseline --steps 2 2> $OUT/share3.log | python -c "$show" "3 ranks on one GPU (gloo), defaults" | tee -a $OUT/share.txt
timeout 900 python bench.py --gpus 2 --workload pairwise --pairs 10000000 --no-cpu-baseline --steps 2 2> $OUT/share2p.log | python -c "$show" "2 ranks on one GPU (gloo), pairwise defaults" | tee -a $OUT/share.txt
unset SVDF_BENCH_SHARE_GPU
timeout 1500 bash tools/shard_scale_probe.sh > $OUT/shard_scale_probe.txt 2>&1; cat $OUT/shard_scale_probe.txt
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/gpu_suite.log 2>&1; echo "gpu suite rc=$?"; tail -3 $OUT/gpu_suite.log
