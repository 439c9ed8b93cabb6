#!/bin/bash
# round 3, GPU call A: window-minibatch kernels -- parity tests, one rank's compute share per pass at N = 2 / 4 / 8, full-size one-rank run
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r03a
timeout 900 python -m pytest tests/test_gpu_window.py -x -q > gpurun_out/r03a/test_window.log 2>&1; echo "test_gpu_window rc=$?" | tee -a gpurun_out/r03a/summary.txt
tail -5 gpurun_out/r03a/test_window.log
timeout 1500 bash tools/shard_scale_probe.sh > gpurun_out/r03a/shard_scale_probe.txt 2>&1
cat gpurun_out/r03a/shard_scale_probe.txt
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --force-exchange --secondary none > gpurun_out/r03a/bench_n1_minibatch.json 2> gpurun_out/r03a/bench_n1_minibatch.log
tail -3 gpurun_out/r03a/bench_n1_minibatch.log; python -c "
import json; d=json.load(open('gpurun_out/r03a/bench_n1_minibatch.json')); print({k: d.get(k) for k in ('value','ms_per_step','rmse_test_after_run','rmse_sequential_reference','rmse_minus_sequential','phase_ms','exchange')})"
