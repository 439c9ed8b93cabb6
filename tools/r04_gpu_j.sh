#!/bin/bash
# round 4, call J: bf16 contribution rows (parity tests; contract over seeds; throughput), the final stratified defaults on every cell, pairs at the demo rate
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r04j
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_wunit.py tests/test_gpu_window.py -x -q > $OUT/tests.log 2>&1
tail -4 $OUT/tests.log
timeout 1500 python tools/contract_seeds.py 0,1,2 2,4 --skip-allreduce --chunks 8 --per-item 16 --checks 3,10 > $OUT/final_c8p16.jsonl 2> $OUT/final.log
grep stratified $OUT/final_c8p16.jsonl
timeout 1500 python tools/contract_seeds.py 0,1,2 2,8 --chunks 8 --per-item 16 --checks 3 --contrib bf16 > $OUT/bf16.jsonl 2> $OUT/bf16.log
grep -v sequential $OUT/bf16.jsonl
WSTEP_CONTRIB=bf16 timeout 900 python tools/wstep_probe.py svdpp 0,1 16 > $OUT/svdpp_bf16.jsonl 2> $OUT/svdpp_bf16.log
cat $OUT/svdpp_bf16.jsonl
WSTEP_CONTRIB=bf16 timeout 900 python tools/wstep_probe.py neighbourhood 0,1 24 > $OUT/neigh_bf16.jsonl 2> $OUT/neigh_bf16.log
cat $OUT/neigh_bf16.jsonl
bash tools/r04_gpu_i.sh
