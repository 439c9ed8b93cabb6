#!/bin/bash
# randomised differential run of round 4 (the round's build: user-unit window kernels, bf16 contributions, k_fewrow_gslots, ADVICE fixes)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/fuzz_r04_final.txt
: > $out
for s in 141 142 143; do timeout 900 python tests/fuzz_wunit.py --iters 400 --seed $s 2>&1 | tail -3 | sed "s/^/wunit seed $s: /" >> $out; done
for s in $(seq 5211 5222); do timeout 900 python tests/fuzz_parity.py --iters 1250 --seed $s 2>&1 | tail -1 | sed "s/^/seed $s: /" >> $out; done
for s in 5251; do timeout 900 python tests/fuzz_parity.py --iters 400 --seed $s --big 2>&1 | tail -1 | sed "s/^/seed $s --big: /" >> $out; done
for s in 131 132; do timeout 900 python tests/fuzz_multi.py --iters 800 --seed $s 2>&1 | tail -1 | sed "s/^/multi seed $s: /" >> $out; done
for s in 5271; do timeout 900 python tests/fuzz_ranker.py --iters 1000 --seed $s 2>&1 | tail -3 | sed "s/^/ranker seed $s: /" >> $out; done
grep -c MISMATCH $out; cat $out | cut -c1-220
