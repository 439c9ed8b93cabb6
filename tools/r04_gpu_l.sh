#!/bin/bash
# round 4, call L: bench --gpus 2 on the shared GPU with the IPC secondaries; the contract cells with bf16 contributions at the ACTUAL defaults
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r04l
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_bench_multi.py -x -q > $OUT/bench_multi.log 2>&1
tail -12 $OUT/bench_multi.log
timeout 1500 python tools/contract_seeds.py 0,1,2 2,4 --chunks 8 --per-item 16 --checks 3 --contrib bf16 --skip-allreduce > $OUT/bf16_n24.jsonl 2> $OUT/a.log
timeout 1500 python tools/contract_seeds.py 0,1,2 8 --chunks 4 --per-item 32 --checks 3 --contrib bf16 > $OUT/bf16_n8.jsonl 2> $OUT/b.log
grep -v sequential $OUT/bf16_n24.jsonl $OUT/bf16_n8.jsonl
