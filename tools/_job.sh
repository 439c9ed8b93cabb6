cd "${GRAFT_REPO_ROOT:-/root/repo}"

for rx in 0 1; do timeout 600 python bench.py --no-cpu-baseline --pmc off --secondary '' --steps 5 --warmup 1 --factor 128 --ratings 50000000 --knob runs_exec=$rx 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('k=128 50M runs_exec=$rx', d['value'], d['ms_per_step'], d['roofline']['frac'], d['config'].get('conflict_free_batches_per_pass'))"; done
