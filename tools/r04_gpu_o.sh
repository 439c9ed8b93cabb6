#!/bin/bash
# round 4, call O: one rank's share per pass with the ROUND-4 defaults (bf16 contribution rows; stratified: 8 chunks / <= 16 per item below 8 ranks,
# 4 chunks / 32 from 8 ranks; 2 blocks per rank), measured like tools/shard_scale_probe.sh: the rank's shard on one GPU, exchange forced on with one rank
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r04o
mkdir -p $OUT
show='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); p=d.get("phase_ms") or {}; print(sys.argv[1], "ms/pass %.2f" % d["ms_per_step"], "launches/pass", d["roofline"]["launches"]//d["steps"], "windows", (d.get("exchange") or {}).get("windows"), "phase_ms", {k: round(v,3) for k,v in p.items() if k!="what"})'
common="--steps 3 --warmup 1 --no-cpu-baseline --no-sequential-reference --force-exchange --pmc off --secondary \"\" --no-window-step --contrib bf16"
for n in 2 4 8; do
  eval python bench.py $common --exchange minibatch --windows 32 --ratings $((100000000/n)) --users $((1000000/n)) 2>/dev/null | python -c "$show" "rank-of-$n all-reduce step (32 windows, bf16 contributions)" | tee -a $OUT/shares.txt
done
eval python bench.py $common --exchange stratified --chunks 32 --ratings 50000000 --users 500000 --items 25000 2>/dev/null | python -c "$show" "rank-of-2 stratified (8 chunks x 4 steps, <= 16 per item: 128 window steps)" | tee -a $OUT/shares.txt
eval python bench.py $common --exchange stratified --chunks 64 --ratings 25000000 --users 250000 --items 12500 2>/dev/null | python -c "$show" "rank-of-4 stratified (8 chunks x 8 steps, <= 16 per item)" | tee -a $OUT/shares.txt
eval python bench.py $common --exchange stratified --chunks 64 --stratified-per-item 32 --ratings 12500000 --users 125000 --items 6250 2>/dev/null | python -c "$show" "rank-of-8 stratified (4 chunks x 16 steps, <= 32 per item)" | tee -a $OUT/shares.txt
