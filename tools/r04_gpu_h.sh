#!/bin/bash
# round 4, call H: stratified schedule, tighter defaults: chunks per pass 8 / 16 (3 seeds x 2 / 4 / 8 ranks, 3 passes)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r04h
timeout 1500 python tools/contract_seeds.py 0,1,2 2,4,8 --skip-allreduce --chunks 8 --checks 3 > gpurun_out/r04h/c8.jsonl 2> gpurun_out/r04h/c8.log
grep stratified gpurun_out/r04h/c8.jsonl
timeout 1500 python tools/contract_seeds.py 1,2 2,4 --skip-allreduce --chunks 16 --checks 3 > gpurun_out/r04h/c16.jsonl 2> gpurun_out/r04h/c16.log
grep stratified gpurun_out/r04h/c16.jsonl
timeout 1500 python tools/contract_seeds.py 1 2,4 --skip-allreduce --chunks 8 --per-item 16 --checks 3 > gpurun_out/r04h/c8p16.jsonl 2> gpurun_out/r04h/c8p16.log
grep stratified gpurun_out/r04h/c8p16.jsonl
