#!/bin/bash
# round 4, call F: the handle's window step for user units (native multi tests), wunit tests, N>1 bench plumbing for SVD++ on a shared GPU
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r04f
timeout 1500 python -m pytest tests/test_gpu_native_multi.py tests/test_gpu_wunit.py -x -q > gpurun_out/r04f/tests.log 2>&1
tail -12 gpurun_out/r04f/tests.log
export SVDF_BENCH_SHARE_GPU=1
timeout 900 python bench.py --gpus 2 --workload svdpp --steps 2 --no-cpu-baseline > gpurun_out/r04f/svdpp_n2.json 2> gpurun_out/r04f/svdpp_n2.log
tail -4 gpurun_out/r04f/svdpp_n2.log; cut -c1-1500 gpurun_out/r04f/svdpp_n2.json
