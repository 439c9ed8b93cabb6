#!/bin/bash
# round 4, call A: the N>1 bench ladder on the one-GPU box + the default N=1 line with in-run PMC traffic
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r04a
timeout 1700 python -m pytest tests/test_gpu_bench_multi.py -x -q > gpurun_out/r04a/test_bench_multi.log 2>&1
tail -5 gpurun_out/r04a/test_bench_multi.log
timeout 900 python bench.py > gpurun_out/r04a/bench_n1.json 2> gpurun_out/r04a/bench_n1.stderr.log
tail -c 1500 gpurun_out/r04a/bench_n1.stderr.log
cut -c1-1200 gpurun_out/r04a/bench_n1.json
