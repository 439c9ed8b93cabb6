#!/bin/bash
# round 3, GPU call B: window-minibatch kernels after the first tuning -- parity tests, probe, kernel stats of one rank of 8, knob A/B,
# full-size one-rank run with the exact sequential reference beside it
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r03b
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_window.py -x -q > $OUT/test_window.log 2>&1; echo "test_gpu_window rc=$?"
tail -3 $OUT/test_window.log
timeout 1500 bash tools/shard_scale_probe.sh > $OUT/shard_scale_probe.txt 2>&1
cat $OUT/shard_scale_probe.txt
show='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); p=d.get("phase_ms") or {}; print(sys.argv[1], "ms/pass %.2f" % d["ms_per_step"], {k: round(v,3) for k,v in p.items() if k!="what"})'
for kn in "window_groups=2" "window_slots=0"; do
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sequential-reference --force-exchange --exchange minibatch --windows 32 --ratings 12500000 --users 125000 --knob $kn 2>/dev/null | python -c "$show" "rank-of-8 $kn" | tee -a $OUT/knobs.txt
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sequential-reference --force-exchange --exchange minibatch --windows 32 --ratings 50000000 --users 500000 --knob $kn 2>/dev/null | python -c "$show" "rank-of-2 $kn" | tee -a $OUT/knobs.txt
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sequential-reference --force-exchange --exchange minibatch --windows 32 --ratings 12500000 --users 125000 > $OUT/kt_bench.json 2> $OUT/kt.stderr.log
find $OUT/kt -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_rank_of_8.csv \;
rm -rf $OUT/kt
head -8 $OUT/kernel_stats_rank_of_8.csv | cut -c1-200
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --force-exchange --secondary "" > $OUT/bench_n1_minibatch.json 2> $OUT/bench_n1_minibatch.log
tail -3 $OUT/bench_n1_minibatch.log; python -c "
import json; d=json.load(open('$OUT/bench_n1_minibatch.json')); print({k: d.get(k) for k in ('value','ms_per_step','rmse_test_after_run','rmse_sequential_reference','rmse_minus_sequential','phase_ms','exchange')})"
