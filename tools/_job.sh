cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r05e
timeout 1500 python -m pytest tests/test_gpu_bench_multi.py -x -q 2>&1 | grep -v amdgpu.ids | tail -12 > gpurun_out/r05e/pytest_multi.log
cat gpurun_out/r05e/pytest_multi.log
timeout 600 python bench.py --no-cpu-baseline --pmc off --secondary '' --steps 10 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['enqueue_ms_per_pass'], d['host_enqueue'])"
