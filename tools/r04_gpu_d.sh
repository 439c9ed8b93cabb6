#!/bin/bash
# round 4, call D: slot kernel of the user-unit window step: parity + throughput at the full configs[3] sizes
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r04d
timeout 1500 python -m pytest tests/test_gpu_wunit.py -x -q > gpurun_out/r04d/test_wunit.log 2>&1
tail -12 gpurun_out/r04d/test_wunit.log
timeout 900 python tools/wstep_probe.py svdpp 0 16,24,48 > gpurun_out/r04d/svdpp.jsonl 2> gpurun_out/r04d/svdpp.log
cat gpurun_out/r04d/svdpp.jsonl; tail -3 gpurun_out/r04d/svdpp.log
timeout 900 python tools/wstep_probe.py neighbourhood 0 32,128 > gpurun_out/r04d/neigh.jsonl 2> gpurun_out/r04d/neigh.log
cat gpurun_out/r04d/neigh.jsonl; tail -3 gpurun_out/r04d/neigh.log
