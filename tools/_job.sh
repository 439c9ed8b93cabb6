#!/bin/bash
# round 6, final measurement job 1: GPU suite + smoke, the driver's bench command, rocprofv3 kernel stats of the same workloads
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r06_final; mkdir -p $OUT
python -m pytest tests -m gpu -x -q 2>&1 | grep -v amdgpu.ids | grep -E "passed|failed|error|Error|FAILED" | tail -8 > $OUT/suite.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -6 >> $OUT/suite.log
cat $OUT/suite.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.stdout 2> $OUT/bench.stderr; echo "bench rc=$? bytes=$(wc -c < $OUT/bench.stdout)"
cp bench_secondary.json $OUT/bench_secondary.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python bench.py --gpus 1 --steps 20 --warmup 5 --pmc off --no-cpu-baseline --svdpp-users 40000 > $OUT/kt_bench.json 2> $OUT/kt.stderr.log
echo "rocprof rc=$?"
find $OUT/kt -name "*kernel_trace.csv" -delete; find $OUT/kt -name "*.db" -delete
python tools/kt_summary.py $OUT/kt 2>/dev/null | head -30
grep -E "k_basicmf_runs_soa|k_window_apply|k_fewrow_slots<16|k_window_users_slots" $(find $OUT/kt -name "*kernel_stats.csv" | head -1) | cut -c1-200
