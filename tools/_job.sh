cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r05b
: > gpurun_out/r05b/bench_ab.txt
run() { tag="$1"; shift; timeout 600 python bench.py --no-cpu-baseline --pmc off --secondary '' --steps 10 --warmup 2 "$@" > /tmp/b.out 2> /tmp/b.err; tail -1 /tmp/b.out | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', round(d['value']/1e9,3), 'G inst/s', round(d['ms_per_step'],3), 'ms', d.get('rmse_test_after_run'))" >> gpurun_out/r05b/bench_ab.txt 2>&1 || tail -5 /tmp/b.err >> gpurun_out/r05b/bench_ab.txt; }
for m in 8 16 24 31; do for w in 1024 1536; do run "stream mode=$m waves=$w" --knob stream_exec=1 --knob stream_waves=$w --knob stream_debug_mode=$m; done; done
cat gpurun_out/r05b/bench_ab.txt
