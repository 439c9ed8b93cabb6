#!/bin/bash
# round 4, call E: accuracy contract of the one-GPU window step over data seeds (full configs[3] sizes), 3 and 10 passes
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r04e
timeout 1500 python tools/wstep_probe.py svdpp 1,2 12,16,24 > gpurun_out/r04e/svdpp.jsonl 2> gpurun_out/r04e/svdpp.log
timeout 600 python tools/wstep_probe.py svdpp 0 12,16 9 >> gpurun_out/r04e/svdpp.jsonl 2>> gpurun_out/r04e/svdpp.log
cat gpurun_out/r04e/svdpp.jsonl
timeout 1500 python tools/wstep_probe.py neighbourhood 1,2 24,64,256 > gpurun_out/r04e/neigh.jsonl 2> gpurun_out/r04e/neigh.log
timeout 600 python tools/wstep_probe.py neighbourhood 0 24,256 9 >> gpurun_out/r04e/neigh.jsonl 2>> gpurun_out/r04e/neigh.log
cat gpurun_out/r04e/neigh.jsonl
