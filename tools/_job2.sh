#!/bin/bash
# round 6, final measurement job 2: hot-lane calibration on the final build (3 seeds), randomised differential runs
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r06_final; mkdir -p $OUT
python tools/hot_lane_calibration.py 100000000 0,1,2 1024,2048,3072 2>&1 | grep -v "^\[svdf\|amdgpu.ids" > $OUT/hot_lane_calibration.txt
cat $OUT/hot_lane_calibration.txt
S=8000; out=$OUT/fuzz.txt; : > $out
for s in $(seq $S $((S+3))); do timeout 900 python tests/fuzz_parity.py --iters 1250 --seed $s 2>&1 | tail -1 | sed "s/^/parity seed $s: /" >> $out; done
timeout 900 python tests/fuzz_parity.py --iters 300 --seed $((S+50)) --big 2>&1 | tail -1 | sed "s/^/parity --big: /" >> $out
timeout 900 python tests/fuzz_wunit.py --iters 400 --seed $((S+60)) 2>&1 | tail -2 | sed "s/^/wunit: /" >> $out
timeout 900 python tests/fuzz_wunit.py --iters 300 --seed $((S+61)) --one-gpu 2>&1 | tail -2 | sed "s/^/wunit one-gpu: /" >> $out
timeout 900 python tests/fuzz_multi.py --iters 500 --seed $((S+70)) 2>&1 | tail -1 | sed "s/^/multi: /" >> $out
timeout 900 python tests/fuzz_ranker.py --iters 600 --seed $((S+80)) 2>&1 | tail -2 | sed "s/^/ranker: /" >> $out
timeout 900 python tests/fuzz_builders.py --iters 200 --seed $((S+90)) 2>&1 | tail -1 | sed "s/^/builders: /" >> $out
echo "MISMATCH lines: $(grep -c MISMATCH $out)"; cut -c1-220 $out
