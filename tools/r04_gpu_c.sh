#!/bin/bash
# round 4, call C: first numbers of the one-GPU window step at the full configs[3] sizes
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r04c
timeout 900 python tools/wstep_probe.py svdpp 0 24,48,1000 > gpurun_out/r04c/svdpp.jsonl 2> gpurun_out/r04c/svdpp.log
cat gpurun_out/r04c/svdpp.jsonl; tail -3 gpurun_out/r04c/svdpp.log
timeout 900 python tools/wstep_probe.py neighbourhood 0 16,32,64,128 > gpurun_out/r04c/neigh.jsonl 2> gpurun_out/r04c/neigh.log
cat gpurun_out/r04c/neigh.jsonl; tail -3 gpurun_out/r04c/neigh.log
