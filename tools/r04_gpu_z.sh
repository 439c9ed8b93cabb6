#!/bin/bash
# round 4, call Z: randomised differential run of the one-GPU window sequences (single contributions in place) against the one-rank simulation
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/fuzz_r04z.txt
: > $out
for s in 71 72 73; do FUZZ_WUNIT_DEBUG=1 timeout 1200 python tests/fuzz_wunit.py --one-gpu --iters 300 --seed $s 2>&1 | grep -v amdgpu.ids | tail -4 | sed "s/^/wunit --one-gpu seed $s: /" >> $out; done
for s in 74 75; do FUZZ_WUNIT_DEBUG=1 timeout 1200 python tests/fuzz_wunit.py --one-gpu --wave --iters 200 --seed $s 2>&1 | grep -v amdgpu.ids | tail -4 | sed "s/^/wunit --one-gpu --wave seed $s: /" >> $out; done
grep -c MISMATCH $out; cat $out | cut -c1-300
