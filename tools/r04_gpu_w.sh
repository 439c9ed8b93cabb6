#!/bin/bash
# round 4, call W: randomised differential run of the one-wave-per-unit window kernel (tests/fuzz_wunit.py --wave) + the general mix
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/fuzz_r04w.txt
: > $out
for s in 51 52 53 54; do timeout 1200 python tests/fuzz_wunit.py --wave --iters 250 --seed $s 2>&1 | tail -3 | sed "s/^/wunit --wave seed $s: /" >> $out; done
for s in 61 62; do timeout 900 python tests/fuzz_wunit.py --iters 400 --seed $s 2>&1 | tail -3 | sed "s/^/wunit seed $s: /" >> $out; done
grep -c MISMATCH $out; cat $out | cut -c1-220
