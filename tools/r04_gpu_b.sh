#!/bin/bash
# round 4, call B: user-unit window step (svdf_k_wunit.hip) vs the oracle simulation + the N>1 bench ladder test
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r04b
timeout 1500 python -m pytest tests/test_gpu_wunit.py -x -q > gpurun_out/r04b/test_wunit.log 2>&1
tail -25 gpurun_out/r04b/test_wunit.log
timeout 1500 python -m pytest tests/test_gpu_bench_multi.py -x -q > gpurun_out/r04b/test_bench_multi.log 2>&1
tail -8 gpurun_out/r04b/test_bench_multi.log
