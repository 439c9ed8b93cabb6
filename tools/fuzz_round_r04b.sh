#!/bin/bash
# randomised differential run on the build with the device-side builders (rand_init on the device, device-built window data sets): every
# configuration starts from a device-initialised model, every window of the multi / wunit runs is regrouped on the GPU
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/fuzz_r04b.txt
: > $out
for s in 4311 4312 4313 4314; do timeout 900 python tests/fuzz_parity.py --iters 1000 --seed $s 2>&1 | tail -1 | sed "s/^/seed $s: /" >> $out; done
for s in 4351; do timeout 900 python tests/fuzz_parity.py --iters 300 --seed $s --big 2>&1 | tail -1 | sed "s/^/seed $s --big: /" >> $out; done
for s in 51 52; do timeout 900 python tests/fuzz_wunit.py --iters 300 --seed $s 2>&1 | tail -1 | sed "s/^/wunit seed $s: /" >> $out; done
for s in 53; do timeout 900 python tests/fuzz_wunit.py --one-gpu --iters 300 --seed $s 2>&1 | tail -1 | sed "s/^/wunit --one-gpu seed $s: /" >> $out; done
for s in 41; do timeout 900 python tests/fuzz_multi.py --iters 600 --seed $s 2>&1 | tail -1 | sed "s/^/multi seed $s: /" >> $out; done
for s in 4371; do timeout 900 python tests/fuzz_ranker.py --iters 500 --seed $s 2>&1 | tail -1 | sed "s/^/ranker seed $s: /" >> $out; done
grep -c MISMATCH $out; cat $out | cut -c1-220
